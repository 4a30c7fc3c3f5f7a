// bf_gemm_params.h — kernel-argument block shared by the GEMM translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GemmParams {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    long long x_sstride;  // elements between samples of x (0 = one x shared by all samples)
    int S, M, N, K;
    int tiles_m, tiles_n;
    int act;      // BF_ACT_* applied to y in the epilogue
    int layers;   // L >= 1 layers that share x: w is [L][S][N][K], bias [L][S][N], y [L][S][M][N] (bf_gemm_nt_layers)
    int flags;  // developer ablation bits (BF_GEMM_ABLATE): 1 = no DMA in the k-loop, 8 = no stores, 16 = no row mask,
                // 64 = every k-step's DMA re-reads k-step 0 (operands always L2-hot)
};

// fast 256x256x64 LDS-DMA kernel (bf_gemm256.hip)
bool bf_gemm256_supported(int x_dtype, int w_dtype, int y_dtype, int S, int M, int N, int K, const void* d_x,
                          const void* d_w, int64_t x_sample_stride);
int bf_launch_gemm256(const GemmParams& p, int w_dtype, int y_dtype, hipStream_t stream);
