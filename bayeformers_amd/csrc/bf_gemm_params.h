// bf_gemm_params.h — kernel-argument block shared by the GEMM translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GemmParams {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    void* y2;  // 256-wide kernel only: when set, act(y) goes to y and the pre-activation y (same dtype) to y2
    // five-slot ring kernel, 16-bit outputs: when set ([S][M][N] of the output type), y is multiplied by act'(gpre) on its way
    // out (act = the activation whose derivative is meant) — the input-gradient GEMM of a layer whose input was act(pre):
    // d pre = (dy W) o act'(pre) without a pass of its own over the activation-sized gradient (bf_gemm_nn_actgrad)
    const void* gpre;
    long long x_sstride;  // elements between samples of x (0 = one x shared by all samples)
    int S, M, N, K;
    int tiles_m, tiles_n;
    int act;      // BF_ACT_* applied to y in the epilogue
    int layers;   // L >= 1 layers that share x: w is [L][S][N][K], bias [L][S][N], y [L][S][M][N] (bf_gemm_nt_layers)
    // tile schedule of the 256-wide persistent kernel (bf_gemm256.hip): workgroup b runs sched[j * gridDim.x + b],
    // j = 0 .. sched_rounds - 1, until an entry with height 0
    const int4* sched;
    int sched_rounds;
    // NN form only: the contraction runs over `segs` segments of K each (x: [segs][S][M][K], w: [segs][S][K][N]) — the
    // input gradient of `segs` layers that read the same activations, dx = sum_l dy_l W_l, in one launch
    int segs;
    long long x_seg_stride, w_seg_stride;
    int flags;  // developer ablation bits, honoured by -DBF_DEV builds only (tools/): 1 = no DMA in the k-loop,
                // 8 = no epilogue, 16 = no global stores, 64 = every k-step's DMA re-reads k-step 0 (operands L2-hot);
                // ring kernel: 32 = the first two k-steps of a tile wait with vmcnt(24) (not held up by the previous tile's
                // stores), bits 8..11 = k: the workgroups of an XCD start k * 64 cycles apart, 4096 = one barrier per k-step from a
                // tile's second k-step on, 8192 = no s_setprio around the MFMA slots
};

// fast 256-wide LDS-DMA kernel (bf_gemm256.hip)
bool bf_gemm256_supported(int x_dtype, int w_dtype, int y_dtype, int S, int M, int N, int K, const void* d_x,
                          const void* d_w, int64_t x_sample_stride);
int bf_launch_gemm256(const GemmParams& p, int w_dtype, int y_dtype, hipStream_t stream);
// the same tile, schedule and ring on fp32 operands and outputs, v_mfma_f32_16x16x4_f32 (bf_gemm256_r5.hip)
bool bf_gemm256_f32_supported(int S, int M, int N, int K, const void* d_x, const void* d_w, const void* d_y,
                              const float* d_bias, int64_t x_sample_stride);
int bf_launch_gemm256_f32(const GemmParams& p, hipStream_t stream);
// TN form (contraction-major operands, fp32 out): out[b][n][k] = sum_m a[b][m][n] * b[b][m][k]
bool bf_gemm256_tn_supported(int dtype, int batch, int Mc, int Nl, int Kl, const void* d_a, const void* d_b,
                             const void* d_out);
int bf_launch_gemm256_tn(const void* d_a, const void* d_b, float* d_out, int dtype, int batch, int Mc, int Nl, int Kl,
                         hipStream_t stream);
// NN form (x K-contiguous, w contraction-major, 16-bit out): y[s][m][k] = sum_n x[s][m][n] * w[s][n][k]
bool bf_gemm256_nn_supported(int dtype, int S, int M, int Nl, int Kl, const void* d_x, const void* d_w, const void* d_y);
int bf_launch_gemm256_nn(const void* d_x, const void* d_w, void* d_y, int dtype, int S, int M, int Nl, int Kl,
                         hipStream_t stream, int segs = 1, const void* d_gpre = nullptr, int act = 0);
bool bf_gemm256_nn_actgrad_supported(int dtype, int S, int M, int Nl, int Kl, const void* d_x, const void* d_w, const void* d_y,
                                     const void* d_gpre);
#ifdef BF_DEV
// round-1 kernel (fixed 256x256 tiles, arithmetic tile order), kept in developer builds as the A/B baseline
int bf_launch_gemm256_r1(const GemmParams& p, int w_dtype, int y_dtype, hipStream_t stream);
#endif
