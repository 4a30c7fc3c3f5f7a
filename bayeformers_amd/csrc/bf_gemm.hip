// bf_gemm.hip — batched NT GEMM on the gfx950 matrix cores: y[s] = x[s] * W_s^T + b_s.
//
// Replaces F.linear(input, weight, bias) at /root/reference/bayeformers/nn/layers/linear.py:104, for all S
// Monte-Carlo samples in one launch (blockIdx.y = sample).  W_s [N][K] is what bf_sample.hip wrote; both operands
// are K-contiguous, which is exactly the layout of the 16x16x32 MFMA fragments (8 consecutive k per lane).
//
// The MFMA is issued with the operands swapped — D = W_frag (rows n) x X_frag (cols m) — so that a lane's four
// accumulator registers are four CONSECUTIVE output features of one row of y: the epilogue stores 8 B (bf16) or
// 16 B (fp32) per lane instead of four scattered scalars.
//
// Kernels in this file
//   gemm_nt_kernel<T,XT,YT,ALIGNED> : bf16/fp16 MFMA (v_mfma_f32_16x16x32_*), 128x128x32 tile, 4 waves (2x2),
//                                     register-prefetched, double-buffered LDS; any M,N,K (K%8 != 0 -> scalar loads)
//   gemm_nt_f32_kernel              : exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), 64x64x16 tile — the parity path
#include <stdlib.h>

#include "bf_common.h"
#include "bf_gemm_params.h"

namespace {

// Kernel choice: 2 = the scheduled 256-wide persistent kernel (bf_gemm256.hip).  Developer builds (-DBF_DEV, tools/)
// can override it with BF_GEMM_VARIANT: 0 = force the generic kernel, 1 = the round-1 fixed-tile kernel (A/B baseline).
int gemm_variant() {
#ifdef BF_DEV
    const char* v = getenv("BF_GEMM_VARIANT");
    if (v) return atoi(v);
#endif
    return 2;
}

template <typename T>
struct Mfma16;
template <>
struct Mfma16<__bf16> {
    using frag = bf16x8_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mfma16<_Float16> {
    using frag = f16x8_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// 8 consecutive elements of a row starting at column k, as T; zero beyond K.
template <typename T, typename ST, bool ALIGNED>
__device__ __forceinline__ typename Mfma16<T>::frag load_chunk(const ST* row, int k, int K) {
    using frag = typename Mfma16<T>::frag;
    frag r;
    if constexpr (ALIGNED) {
        if (k < K) {
            if constexpr (sizeof(ST) == 2) {
                r = *reinterpret_cast<const frag*>(row + k);
            } else {
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(row + k);
                const f32x4_t b = *reinterpret_cast<const f32x4_t*>(row + k + 4);
                const f32x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                r = __builtin_convertvector(v, frag);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = (T)0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (k + i < K) ? (T)(float)row[k + i] : (T)0.0f;
    }
    return r;
}

template <typename YT>
__device__ __forceinline__ void store4(YT* y, long long m, int n, int M, int N, f32x4_t v, bool vec_ok) {
    if (m >= M || n >= N) return;
    YT* o = y + m * (long long)N + n;
    if (vec_ok && n + 3 < N) {
        if constexpr (sizeof(YT) == 4) {
            *reinterpret_cast<f32x4_t*>(o) = v;
        } else if constexpr (__is_same(YT, __bf16)) {
            *reinterpret_cast<bf16x4_t*>(o) = __builtin_convertvector(v, bf16x4_t);
        } else {
            *reinterpret_cast<f16x4_t*>(o) = __builtin_convertvector(v, f16x4_t);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < N) o[j] = (YT)v[j];
    }
}

constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 8;

template <typename T, typename XT, typename YT, bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
    using frag = typename Mfma16<T>::frag;
    __shared__ __attribute__((aligned(16))) T sX[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) T sW[2][BN][LDK];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int s = blockIdx.y;
    const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = p.M, N = p.N, K = p.K;

    const XT* x = reinterpret_cast<const XT*>(p.x) + (long long)s * p.x_sstride;
    const T* w = reinterpret_cast<const T*>(p.w) + (long long)s * N * K;

    // loader mapping: two 8-element chunks per operand per thread
    const int lr = tid >> 2, lk = (tid & 3) * 8;
    const XT* xrow0 = x + (long long)min(m0 + lr, M - 1) * K;
    const XT* xrow1 = x + (long long)min(m0 + lr + 64, M - 1) * K;
    const T* wrow0 = w + (long long)min(n0 + lr, N - 1) * K;
    const T* wrow1 = w + (long long)min(n0 + lr + 64, N - 1) * K;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (K + BK - 1) / BK;
    frag gx0, gx1, gw0, gw1;
    gx0 = load_chunk<T, XT, ALIGNED>(xrow0, lk, K);
    gx1 = load_chunk<T, XT, ALIGNED>(xrow1, lk, K);
    gw0 = load_chunk<T, T, ALIGNED>(wrow0, lk, K);
    gw1 = load_chunk<T, T, ALIGNED>(wrow1, lk, K);
    *reinterpret_cast<frag*>(&sX[0][lr][lk]) = gx0;
    *reinterpret_cast<frag*>(&sX[0][lr + 64][lk]) = gx1;
    *reinterpret_cast<frag*>(&sW[0][lr][lk]) = gw0;
    *reinterpret_cast<frag*>(&sW[0][lr + 64][lk]) = gw1;
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            const int k = (kt + 1) * BK + lk;
            gx0 = load_chunk<T, XT, ALIGNED>(xrow0, k, K);
            gx1 = load_chunk<T, XT, ALIGNED>(xrow1, k, K);
            gw0 = load_chunk<T, T, ALIGNED>(wrow0, k, K);
            gw1 = load_chunk<T, T, ALIGNED>(wrow1, k, K);
        }
        frag wf[4], xf[4];
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[i] = *reinterpret_cast<const frag*>(&sW[buf][wn * 64 + i * 16 + fr][fk]);
            xf[i] = *reinterpret_cast<const frag*>(&sX[buf][wm * 64 + i * 16 + fr][fk]);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = Mfma16<T>::run(wf[nb], xf[mb], acc[nb][mb]);
        if (more) {
            *reinterpret_cast<frag*>(&sX[buf ^ 1][lr][lk]) = gx0;
            *reinterpret_cast<frag*>(&sX[buf ^ 1][lr + 64][lk]) = gx1;
            *reinterpret_cast<frag*>(&sW[buf ^ 1][lr][lk]) = gw0;
            *reinterpret_cast<frag*>(&sW[buf ^ 1][lr + 64][lk]) = gw1;
        }
        __syncthreads();
    }

    // epilogue: D rows = n (4 consecutive per lane), cols = m
    YT* y = reinterpret_cast<YT*>(p.y) + (long long)s * M * N;
    const float* bias = p.bias ? p.bias + (long long)s * N : nullptr;
    const bool vec_ok = (N % 4) == 0 && ((uintptr_t)p.y % 16) == 0;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int n = n0 + wn * 64 + nb * 16 + (lane >> 4) * 4;
        f32x4_t b = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) b[j] = bias[n + j];
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const long long m = m0 + wm * 64 + mb * 16 + (lane & 15);
            store4<YT>(y, m, n, M, N, bf_apply_act(acc[nb][mb] + b, p.act), vec_ok);
        }
    }
}

// exact fp32: v_mfma_f32_16x16x4_f32 == a k-ordered fmaf chain (one rounding per product).
constexpr int FM = 64, FN = 64, FK = 16, FLD = FK + 1;

__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const GemmParams p) {
    __shared__ float sX[FM][FLD];
    __shared__ float sW[FN][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int s = blockIdx.y;
    const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
    const int m0 = tm * FM, n0 = tn * FN;
    const int M = p.M, N = p.N, K = p.K;
    const float* x = reinterpret_cast<const float*>(p.x) + (long long)s * p.x_sstride;
    const float* w = reinterpret_cast<const float*>(p.w) + (long long)s * N * K;

    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const float* xrow = x + (long long)min(m0 + lr, M - 1) * K;
    const float* wrow = w + (long long)min(n0 + lr, N - 1) * K;

    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += FK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + lk + i;
            sX[lr][lk + i] = k < K ? xrow[k] : 0.f;
            sW[lr][lk + i] = k < K ? wrow[k] : 0.f;
        }
        __syncthreads();
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < FK / 4; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = sW[wn * 32 + i * 16 + fr][kk * 4 + fk];
                b[i] = sX[wm * 32 + i * 16 + fr][kk * 4 + fk];
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb], b[mb], acc[nb][mb], 0, 0, 0);
        }
        __syncthreads();
    }

    float* y = reinterpret_cast<float*>(p.y) + (long long)s * M * N;
    const float* bias = p.bias ? p.bias + (long long)s * N : nullptr;
    const bool vec_ok = (N % 4) == 0 && ((uintptr_t)p.y % 16) == 0;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + wn * 32 + nb * 16 + (lane >> 4) * 4;
        f32x4_t b = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) b[j] = bias[n + j];
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const long long m = m0 + wm * 32 + mb * 16 + (lane & 15);
            store4<float>(y, m, n, M, N, bf_apply_act(acc[nb][mb] + b, p.act), vec_ok);
        }
    }
}

template <typename T, typename XT, typename YT>
int launch_16(const GemmParams& p, bool aligned, hipStream_t stream) {
    dim3 grid((uint32_t)(p.tiles_m * p.tiles_n), (uint32_t)p.S);
    if (aligned)
        hipLaunchKernelGGL((gemm_nt_kernel<T, XT, YT, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<T, XT, YT, false>), grid, dim3(256), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int launch_16_xy(const GemmParams& p, int x_dtype, int y_dtype, int t_dtype, bool aligned, hipStream_t stream) {
    const bool xf = x_dtype == BF_DT_F32, yf = y_dtype == BF_DT_F32;
    if (!xf && x_dtype != t_dtype) BF_FAIL("bf_gemm_nt: x dtype %d incompatible with w dtype %d", x_dtype, t_dtype);
    if (!yf && y_dtype != t_dtype) BF_FAIL("bf_gemm_nt: y dtype %d incompatible with w dtype %d", y_dtype, t_dtype);
    if (xf && yf) return launch_16<T, float, float>(p, aligned, stream);
    if (xf) return launch_16<T, float, T>(p, aligned, stream);
    if (yf) return launch_16<T, T, float>(p, aligned, stream);
    return launch_16<T, T, T>(p, aligned, stream);
}

}  // namespace

int bf_launch_gemm_nt(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                      const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, hipStream_t stream,
                      int act, int layers, void* d_pre) {
    if (layers < 1) BF_FAIL("bf_gemm_nt: layers must be >= 1 (got %d)", layers);
    if (d_pre) {
        // pre-activation wanted next to act(y) (the forward of a training step): one launch with two outputs when the
        // 256-wide kernel takes the shape, else the GEMM into d_pre followed by the elementwise activation
        const bool fast = layers == 1 && gemm_variant() != 0 && w_dtype != BF_DT_F32 && (long long)M * N >= 128 * 128 &&
                          ((uintptr_t)d_bias & 15) == 0 && ((uintptr_t)d_pre & 15) == 0 &&
                          bf_gemm256_supported(x_dtype, w_dtype, y_dtype, S, M, N, K, d_x, d_w, x_sample_stride);
        if (!fast || act == BF_ACT_NONE) {
            int rc = bf_launch_gemm_nt(d_x, x_dtype, x_sample_stride, d_w, w_dtype, d_bias, d_pre, y_dtype, S, M, N, K,
                                       stream, BF_ACT_NONE, layers, nullptr);
            if (rc) return rc;
            if (act == BF_ACT_NONE)
                BF_HIP_CHECK(hipMemcpyAsync(d_y, d_pre, (size_t)layers * S * M * N * bf_dtype_size(y_dtype),
                                            hipMemcpyDeviceToDevice, stream));
            else
                rc = bf_launch_gelu(d_pre, d_y, y_dtype, (uint64_t)layers * S * M * N, stream);
            return rc;
        }
    }
    if (layers > 1) {
        // L layers sharing x: one launch of the 256x256 kernel when it applies, else one launch per layer
        const bool fast = gemm_variant() != 0 && (long long)M * N >= 128 * 128 &&
                          ((uintptr_t)d_bias & 15) == 0 && (long long)layers * S <= 65535 &&
                          (w_dtype == BF_DT_F32
                               ? x_dtype == BF_DT_F32 && y_dtype == BF_DT_F32 && getenv("BF_F32_GENERIC") == nullptr &&
                                     bf_gemm256_f32_supported(layers * S, M, N, K, d_x, d_w, d_y, d_bias, x_sample_stride)
                               : bf_gemm256_supported(x_dtype, w_dtype, y_dtype, layers * S, M, N, K, d_x, d_w, x_sample_stride));
        if (!fast) {
            const size_t ws = bf_dtype_size(w_dtype), ys = bf_dtype_size(y_dtype);
            for (int l = 0; l < layers; ++l) {
                const int rc = bf_launch_gemm_nt(
                    d_x, x_dtype, x_sample_stride, (const char*)d_w + (size_t)l * S * N * K * ws, w_dtype,
                    d_bias ? d_bias + (size_t)l * S * N : nullptr, (char*)d_y + (size_t)l * S * M * N * ys, y_dtype, S,
                    M, N, K, stream, act, 1, nullptr);
                if (rc) return rc;
            }
            return 0;
        }
    }
    if (!d_x || !d_w || !d_y) BF_FAIL("bf_gemm_nt: NULL operand");
    if (S < 1 || M < 1 || N < 1 || K < 1) BF_FAIL("bf_gemm_nt: bad shape S=%d M=%d N=%d K=%d", S, M, N, K);
    if (S > 65535) BF_FAIL("bf_gemm_nt: S=%d exceeds gridDim.y", S);
    GemmParams p{};
    p.x = d_x;
    p.w = d_w;
    p.bias = d_bias;
    p.y = d_y;
    p.y2 = d_pre;
    p.x_sstride = x_sample_stride;
    p.S = S;
    p.M = M;
    p.N = N;
    p.K = K;
    p.act = act;
    p.layers = layers;
    if (w_dtype == BF_DT_F32) {
        if (x_dtype != BF_DT_F32 || y_dtype != BF_DT_F32) BF_FAIL("bf_gemm_nt: fp32 weights need fp32 x and y");
        // large aligned problems: the 256-wide ring kernel on v_mfma_f32_16x16x4_f32 (BF_F32_GENERIC: developer A/B)
        static const bool generic_only = getenv("BF_F32_GENERIC") != nullptr;
        if (!generic_only && (long long)M * N >= 128 * 128 && (long long)layers * S <= 65535 &&
            bf_gemm256_f32_supported(layers * S, M, N, K, d_x, d_w, d_y, d_bias, x_sample_stride) &&
            (!d_pre || ((uintptr_t)d_pre & 15) == 0))
            return bf_launch_gemm256_f32(p, stream);
        p.tiles_m = (M + FM - 1) / FM;
        p.tiles_n = (N + FN - 1) / FN;
        hipLaunchKernelGGL(gemm_nt_f32_kernel, dim3((uint32_t)(p.tiles_m * p.tiles_n), (uint32_t)S), dim3(256), 0,
                           stream, p);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // large aligned problems: the 256x256x64 LDS-DMA kernel; everything else: the generic 128x128x32 kernel
    const int variant = gemm_variant();
    if (variant != 0 && (long long)M * N >= 128 * 128 && ((uintptr_t)d_bias & 15) == 0 &&
        bf_gemm256_supported(x_dtype, w_dtype, y_dtype, layers * S, M, N, K, d_x, d_w, x_sample_stride)) {
#ifdef BF_DEV
        if (variant == 1) return bf_launch_gemm256_r1(p, w_dtype, y_dtype, stream);
#endif
        return bf_launch_gemm256(p, w_dtype, y_dtype, stream);
    }
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    const size_t xs = bf_dtype_size(x_dtype);
    const bool aligned = (K % 8) == 0 && ((uintptr_t)d_x % 16) == 0 && ((uintptr_t)d_w % 16) == 0 &&
                         ((size_t)x_sample_stride * xs) % 16 == 0;
    if (w_dtype == BF_DT_BF16) return launch_16_xy<__bf16>(p, x_dtype, y_dtype, BF_DT_BF16, aligned, stream);
    if (w_dtype == BF_DT_F16) return launch_16_xy<_Float16>(p, x_dtype, y_dtype, BF_DT_F16, aligned, stream);
    BF_FAIL("bf_gemm_nt: bad w dtype %d", w_dtype);
}
