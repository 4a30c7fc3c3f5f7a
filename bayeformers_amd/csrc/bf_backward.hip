// bf_backward.hip — backward of the sampled-weight linear layer.
//
// Reference behaviour being reproduced: autograd through `F.linear(input, mu + eps * softplus(rho), ...)`
// (/root/reference/bayeformers/nn/layers/linear.py:97,104, parameters/gaussian.py:100-101) with eps a constant of
// the graph (drawn under no_grad) and the two log-prob scalars DETACHED (`.data =`, linear.py:99-102), i.e. the KL
// terms give no gradient in the reference; only the likelihood path does:
//     dX[s]   = dY[s] W_s                      dW_s = dY[s]^T X[s]
//     dmu_w   = sum_s dW_s                     drho_w = (sum_s dW_s * eps_s) * softplus'(rho_w)
//     dmu_b   = sum_s sum_m dY[s][m][:]        drho_b = (sum_s db_s * eps_b,s) * softplus'(rho_b)
// eps is REGENERATED from the Philox counter (same seed / sample index / stream as the forward): nothing but x is
// kept from the forward.
//
// Kernels here are the glue around the two MFMA GEMMs (bf_gemm_nt): a batched 16-bit transpose (the NT kernel wants
// both operands contiguous along the reduction axis), the column sum for the bias, and the eps-weighted reduction
// over samples.  All HBM-bound streaming kernels.
#include <algorithm>

#include "bf_common.h"
#include "bf_philox.h"

namespace {

// out[b][c][r] = in[b][r][c] for 2- or 4-byte elements; 64x64 tile through LDS (padded), 256 threads.
template <typename E>
__global__ __launch_bounds__(256) void transpose_kernel(const E* __restrict__ in, E* __restrict__ out, int rows,
                                                        int cols) {
    __shared__ E tile[64][64 + 4 / sizeof(E)];
    const long long b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row-groups
    in += b * (long long)rows * cols;
    out += b * (long long)rows * cols;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < rows && c < cols) ? in[(long long)r * cols + c] : (E)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(long long)c * rows + r] = tile[tx][ty * 16 + i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ out, int M, int N) {
    // grid = (ceil(N/64), S); block = 64 columns x 4 row-lanes
    __shared__ float sh[4][64];
    const int s = blockIdx.y, n = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;
    const T* p = dy + (long long)s * M * N;
    float acc = 0.f;
    if (n < N)
        for (int m = lane_r; m < M; m += 4) acc += (float)p[(long long)m * N + n];
    sh[lane_r][threadIdx.x & 63] = acc;
    __syncthreads();
    if (lane_r == 0 && n < N) out[(long long)s * N + n] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// 16-bit transpose, 64x64 tile, 16-byte global accesses on both sides.  Rows are packed in pairs into dwords on the
// way into LDS, so that the transposed image tileT[c][r/2] is read back as 16-byte vectors of 8 consecutive r.
// The 4-dword chunks of a tileT row are XOR-swizzled with (c >> 3): conflict-free 4-byte writes and 16-byte reads.
// Requires rows % 64 == 0, cols % 64 == 0 and 16-byte aligned bases.
//
// SUM_DT != 0 (BF_DT_BF16 / BF_DT_F16): the block also leaves the column sums of its 64 rows, in fp32, at
// colpart[(b * gridDim.y + blockIdx.y) * cols + c] — the bias gradient's partial sums for free while dy is transposed.
template <int SUM_DT>
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                          int rows, int cols, float* __restrict__ colpart) {
    __shared__ __attribute__((aligned(16))) uint32_t tileT[64][32];
    const long long b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    in += b * (long long)rows * cols;
    out += b * (long long)rows * cols;
    const int t = threadIdx.x;
    {
        const int rp = t >> 3, cv = t & 7;  // row pair 0..31, 8-column vector 0..7
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        const u32x4 a = *reinterpret_cast<const u32x4*>(in + (long long)(r0 + 2 * rp) * cols + c0 + cv * 8);
        const u32x4 bq = *reinterpret_cast<const u32x4*>(in + (long long)(r0 + 2 * rp + 1) * cols + c0 + cv * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t ea = (j & 1) ? (a[j >> 1] >> 16) : (a[j >> 1] & 0xffffu);
            const uint32_t eb = (j & 1) ? (bq[j >> 1] >> 16) : (bq[j >> 1] & 0xffffu);
            const int c = cv * 8 + j;
            tileT[c][(((rp >> 2) ^ cv) << 2) | (rp & 3)] = ea | (eb << 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = pass * 32 + (t >> 3), rv = t & 7;  // output row c, 8 consecutive r = 4 dwords
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&tileT[c][(rv ^ ((c >> 3) & 7)) << 2]);
        *reinterpret_cast<u32x4*>(out + (long long)(c0 + c) * rows + r0 + rv * 8) = v;
        if constexpr (SUM_DT != 0) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SUM_DT == BF_DT_BF16) {
                    acc += __builtin_bit_cast(float, v[j] << 16) + __builtin_bit_cast(float, v[j] & 0xffff0000u);
                } else {
                    const auto h = __builtin_bit_cast(__attribute__((ext_vector_type(2))) _Float16, v[j]);
                    acc += (float)h[0] + (float)h[1];
                }
            }
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 4);
            if (rv == 0) colpart[((long long)b * gridDim.y + blockIdx.y) * cols + c0 + c] = acc;
        }
    }
}

// column sums of dy[s] ([M][N], 16-bit), 16-byte loads: block = 64 column-vectors (512 columns) x 4 row lanes over a
// chunk of kColsumRows rows; partial[s][chunk][n] is reduced by colsum_finish_kernel (deterministic order).
constexpr int kColsumRows = 256;
template <typename T, typename V8>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ dy, float* __restrict__ partial,
                                                             int M, int N, int chunks) {
    __shared__ float sh[4][64][8];
    const int s = blockIdx.z, ch = blockIdx.y, cv = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const T* p = dy + (long long)s * M * N;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int m1 = min(M, (ch + 1) * kColsumRows);
    if (cv * 8 < N) {
        // (a launch is only a few waves per CU: 8 row loads are in flight per thread before any is consumed)
        constexpr int U = 8;
        int m = ch * kColsumRows + rl;
        for (; m + 4 * (U - 1) < m1; m += 4 * U) {
            V8 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(p + (long long)(m + 4 * u) * N + cv * 8));
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += (float)v[u][i];
        }
        for (; m < m1; m += 4) {
            const V8 v = *reinterpret_cast<const V8*>(p + (long long)m * N + cv * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sh[rl][threadIdx.x & 63][i] = acc[i];
    __syncthreads();
    if (rl == 0 && cv * 8 < N) {
        float* o = partial + ((long long)s * chunks + ch) * N + cv * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o[i] = (sh[0][threadIdx.x][i] + sh[1][threadIdx.x][i]) + (sh[2][threadIdx.x][i] + sh[3][threadIdx.x][i]);
    }
}
// The backward of a GELU fused into the forward GEMM's epilogue, merged with the bias gradient of that layer:
// dpre = dy * gelu'(pre) (fp32 arithmetic, rounded to T as the unfused GeluBackward would) and the column sums of the
// ROUNDED dpre (what the weight-gradient GEMM consumes).  Same blocking as colsum_partial_kernel.
template <typename T, typename V8>
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const T* __restrict__ dy, const T* __restrict__ pre,
                                                              T* __restrict__ dpre, float* __restrict__ partial, int M,
                                                              int N, int chunks) {
    __shared__ float sh[4][64][8];
    const int s = blockIdx.z, ch = blockIdx.y, cv = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long long base = (long long)s * M * N;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int m1 = min(M, (ch + 1) * kColsumRows);
    if (cv * 8 < N) {
        for (int m = ch * kColsumRows + rl; m < m1; m += 4) {
            const long long o = base + (long long)m * N + cv * 8;
            const V8 g = __builtin_nontemporal_load(reinterpret_cast<const V8*>(dy + o));
            const V8 x = __builtin_nontemporal_load(reinterpret_cast<const V8*>(pre + o));
            V8 r;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                r[i] = (T)((float)g[i] * bf_gelu_grad((float)x[i]));
                acc[i] += (float)r[i];
            }
            *reinterpret_cast<V8*>(dpre + o) = r;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sh[rl][threadIdx.x & 63][i] = acc[i];
    __syncthreads();
    if (rl == 0 && cv * 8 < N) {
        float* o = partial + ((long long)s * chunks + ch) * N + cv * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o[i] = (sh[0][threadIdx.x][i] + sh[1][threadIdx.x][i]) + (sh[2][threadIdx.x][i] + sh[3][threadIdx.x][i]);
    }
}

template <typename T, typename V8>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ in, T* __restrict__ out, unsigned long long n8) {
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n8; i += (unsigned long long)gridDim.x * 256ull) {
        const V8 x = reinterpret_cast<const V8*>(in)[i];
        V8 r;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2_t v = bf_gelu2(f32x2_t{(float)x[j], (float)x[j + 1]});
            r[j] = (T)v[0];
            r[j + 1] = (T)v[1];
        }
        reinterpret_cast<V8*>(out)[i] = r;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_tail_kernel(const T* __restrict__ in, T* __restrict__ out, unsigned long long lo,
                                                        unsigned long long n) {
    const unsigned long long i = lo + blockIdx.x * 256ull + threadIdx.x;
    if (i < n) out[i] = (T)bf_gelu((float)in[i]);
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            int N, int chunks) {
    // block = 64 columns x 4 chunk lanes (the chunk axis is 64-128 long: one serial thread per column was
    // latency-bound at 18 us per call); fixed summation order -> deterministic
    __shared__ float sh[4][64];
    const int s = blockIdx.y, n = blockIdx.x * 64 + (threadIdx.x & 63), cl = threadIdx.x >> 6;
    float acc = 0.f;
    if (n < N)
        for (int c = cl; c < chunks; c += 4) acc += partial[((long long)s * chunks + c) * N + n];
    sh[cl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (cl == 0 && n < N)
        out[(long long)s * N + n] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// dmu[e] = sum_s dw[s][e];  drho[e] = (sum_s dw[s][e] * eps(s, e)) * softplus'(rho[e]).  thread = 4 scalars = one Philox
// block per sample.  dw is [S][splits][n]: the split-K partial products of one sample are summed first.
// A launch has only ~9 waves per CU (n / 4 threads), so the kernel lives on loads in flight: the 16-byte loads of
// kPgBatch samples (x splits) are issued together before any arithmetic (VEC: n % 4 == 0 and 16-byte aligned dw; one
// load per thread and sample-split otherwise runs at 1.7 TB/s), and the sample-independent half of the Philox rounds
// is computed once per thread (bf_philox_prepare).
constexpr int kPgBatch = 5;  // 10 measured: no change (32.44-32.49 vs 32.45-32.50 ms per training step, three interleaved runs)
// `g` = the thread's group of 4 scalars within the tensor
template <bool VEC>
__device__ __forceinline__ void param_grad_body(const float* __restrict__ dw, const float* __restrict__ rho,
                                                unsigned long long n, int S, int splits, uint32_t k0, uint32_t k1,
                                                uint32_t sample_base, uint32_t stream, float* __restrict__ dmu,
                                                float* __restrict__ drho, unsigned long long g) {
    const unsigned long long e0 = g * 4;
    if (e0 >= n) return;
    const int nv = n - e0 >= 4 ? 4 : (int)(n - e0);
    const bf_philox_inv inv = bf_philox_prepare((uint32_t)g, (uint32_t)(g >> 32), stream, k0, k1);
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, se[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC) {
        const unsigned long long sstride = (unsigned long long)splits * n;
        for (int s0 = 0; s0 < S; s0 += kPgBatch) {
            f32x4_t d[kPgBatch];
#pragma unroll
            for (int b = 0; b < kPgBatch; ++b) {
                d[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (s0 + b < S) d[b] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(dw + (s0 + b) * sstride + e0));
            }
            for (int j = 1; j < splits; ++j) {
                f32x4_t t[kPgBatch];
#pragma unroll
                for (int b = 0; b < kPgBatch; ++b) {
                    t[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (s0 + b < S)
                        t[b] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(dw + (s0 + b) * sstride + j * n + e0));
                }
#pragma unroll
                for (int b = 0; b < kPgBatch; ++b) d[b] += t[b];
            }
#pragma unroll
            for (int b = 0; b < kPgBatch; ++b) {
                if (s0 + b < S) {
                    float z[4];
                    bf_normal4_split_dev(inv, sample_base + (uint32_t)(s0 + b), stream, k0, k1, z);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        sm[i] += d[b][i];
                        se[i] = fmaf(d[b][i], z[i], se[i]);
                    }
                }
            }
        }
    } else {
        for (int s = 0; s < S; ++s) {
            float z[4];
            bf_normal4_split_dev(inv, sample_base + (uint32_t)s, stream, k0, k1, z);
            const float* p = dw + (unsigned long long)s * splits * n + e0;
            float d[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < splits; ++j, p += n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] += i < nv ? p[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sm[i] += d[i];
                se[i] = fmaf(d[i], z[i], se[i]);
            }
        }
    }
    for (int i = 0; i < nv; ++i) {
        if (dmu) dmu[e0 + i] = sm[i];
        const float r = rho[e0 + i];
        // torch softplus backward (beta = 1, threshold = 20): grad * (r > 20 ? 1 : e^r / (e^r + 1))
        const float zr = expf(r);
        drho[e0 + i] = se[i] * (r > 20.0f ? 1.0f : zr / (zr + 1.0f));
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void param_grad_kernel(const float* __restrict__ dw, const float* __restrict__ rho,
                                                         unsigned long long n, int S, int splits, uint32_t k0, uint32_t k1,
                                                         uint32_t sample_base, const uint32_t* __restrict__ counter,
                                                         uint32_t stream, float* __restrict__ dmu,
                                                         float* __restrict__ drho) {
    sample_base += counter ? *counter : 0u;
    param_grad_body<VEC>(dw, rho, n, S, splits, k0, k1, sample_base, stream, dmu, drho,
                         (unsigned long long)blockIdx.x * 256 + threadIdx.x);
}

// The same for MANY tensors in one launch (bf_param_grad_table; the sampling plan's table pattern, bf_sample.hip): block b
// belongs to entry entry_of_block[b]; its threads are groups (b - block_begin) * 256 + tid of that entry's tensor.  A
// BERT-base training step has 73 weight tensors whose per-sample gradients (4.6 GB of fp32 with the split-K partials) were
// read by 73 launches of ~9 waves per CU each; one launch of 83 k blocks keeps the loads of every CU in flight.
struct PgEntry {
    const float* dw;
    const float* rho;
    float* dmu;
    float* drho;
    unsigned long long n;
    uint32_t stream, block_begin;
    int32_t splits, vec;
};
__global__ __launch_bounds__(256) void param_grad_table_kernel(const PgEntry* __restrict__ table,
                                                               const uint32_t* __restrict__ entry_of_block, int S, uint32_t k0,
                                                               uint32_t k1, uint32_t sample_base,
                                                               const uint32_t* __restrict__ counter) {
    const PgEntry& e = table[entry_of_block[blockIdx.x]];
    sample_base += counter ? *counter : 0u;
    const unsigned long long g = (unsigned long long)(blockIdx.x - e.block_begin) * 256 + threadIdx.x;
    if (__builtin_amdgcn_readfirstlane(e.vec))
        param_grad_body<true>(e.dw, e.rho, e.n, S, __builtin_amdgcn_readfirstlane(e.splits), k0, k1, sample_base, e.stream, e.dmu, e.drho, g);
    else
        param_grad_body<false>(e.dw, e.rho, e.n, S, __builtin_amdgcn_readfirstlane(e.splits), k0, k1, sample_base, e.stream, e.dmu, e.drho, g);
}

// Opt-in Bayes-by-Backprop gradient of the KL terms (the reference detaches them, layers/linear.py:99-102):
//   L = sum_s gp[s] * log_prior_s + gq[s] * log_q_s,   W_s = mu + softplus(rho) * eps_s
//   dL/dmu[e]  = sum_s gp[s] * score(W_s[e])
//   dL/drho[e] = softplus'(rho[e]) * sum_s ( gp[s] * score(W_s[e]) * eps_s[e]  -  gq[s] / sigma[e] )
// score(w) = d log p(w) / dw:  Gaussian prior -(w - mu_p) / sigma_p^2;  mixture -w * (r1/s1^2 + r2/s2^2) with the
// responsibilities r_i of the two components.  eps regenerated from the Philox counter; thread = 4 scalars.
struct KlParams {
    const float* mu;
    const float* rho;
    const float* mu_p;
    const float* rho_p;
    const double* g;  // [S][2] = {dL/dlog_prior_s, dL/dlog_q_s}
    float* dmu;
    float* drho;
    unsigned long long n;
    float a1, b1, a2, b2;  // mixture: t_i = a_i w^2 + b_i (natural log), a_i = -1/(2 s_i^2)
    int prior;
    int S;
    uint32_t k0, k1, sample_base, stream;
    const uint32_t* counter;
    bf_prior_check_t chk;  // mixture constants against the device scalars they were read from (bf_prior_t)
    uint32_t* stale;       // bf_stale_counter
};

__global__ __launch_bounds__(256) void kl_grad_kernel(const KlParams p) {
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long e0 = g * 4;
    if (e0 >= p.n) return;
    const int nv = p.n - e0 >= 4 ? 4 : (int)(p.n - e0);
    const uint32_t base = p.sample_base + (p.counter ? *p.counter : 0u);
    float mu[4], sg[4], dsp[4], pmu[4], pinv2[4], am[4] = {0.f, 0.f, 0.f, 0.f}, ar[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) {
        const bool ok = i < nv;
        mu[i] = ok ? p.mu[e0 + i] : 0.f;
        const float r = ok ? p.rho[e0 + i] : 0.f;
        const float zr = expf(r);
        sg[i] = r > 20.0f ? r : log1pf(zr);
        dsp[i] = r > 20.0f ? 1.0f : zr / (zr + 1.0f);
        pmu[i] = 0.f;
        pinv2[i] = 0.f;
        if (p.prior == BF_PRIOR_GAUSSIAN && ok) {
            const float rp = p.rho_p[e0 + i];
            const float sp = rp > 20.0f ? rp : log1pf(expf(rp));
            pmu[i] = p.mu_p[e0 + i];
            pinv2[i] = 1.0f / (sp * sp);
        }
    }
    for (int s = 0; s < p.S; ++s) {
        const float gp = (float)p.g[2 * s], gq = (float)p.g[2 * s + 1];
        float z[4];
        bf_normal4_dev((uint32_t)g, (uint32_t)(g >> 32), base + (uint32_t)s, p.stream, p.k0, p.k1, z);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float w = fmaf(sg[i], z[i], mu[i]);
            float score = 0.f;
            if (p.prior == BF_PRIOR_GAUSSIAN) {
                score = -(w - pmu[i]) * pinv2[i];
            } else if (p.prior == BF_PRIOR_MIXTURE) {
                const float w2 = w * w;
                const float t1 = fmaf(p.a1, w2, p.b1), t2 = fmaf(p.a2, w2, p.b2);
                const float m = fmaxf(t1, t2);
                const float e1 = expf(t1 - m), e2 = expf(t2 - m);
                const float r1 = e1 / (e1 + e2);
                // 1/s_i^2 = -2 a_i
                score = 2.0f * w * (r1 * p.a1 + (1.0f - r1) * p.a2);
            }
            am[i] = fmaf(gp, score, am[i]);
            ar[i] += gp * score * z[i] - gq / sg[i];
        }
    }
    const bool stale = bf_prior_check_failed(p.chk);  // stale constants: the gradient is poisoned, the host is told
    if (stale && g == 0) bf_stale_bump(p.stale);
    for (int i = 0; i < nv; ++i) {
        if (p.dmu) p.dmu[e0 + i] = stale ? __builtin_nanf("") : am[i];
        p.drho[e0 + i] = stale ? __builtin_nanf("") : ar[i] * dsp[i];
    }
}

// ---------------------------------------------------------------------------------------------- bnn.Embedding
// out[t][:] = mu[id_t][:] + softplus(rho[id_t][:]) * eps(sample of token t, element id_t*D + d): only the gathered rows
// are sampled; the same id inside one sample gets the same vector (one table draw per sample, as for bnn.Linear).
template <typename OT>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ ids,
                                                            const float* __restrict__ mu, const float* __restrict__ rho,
                                                            OT* __restrict__ out, long long n_tokens,
                                                            long long tokens_per_sample, long long V, int D, uint32_t k0,
                                                            uint32_t k1, uint32_t sample_base,
                                                            const uint32_t* __restrict__ counter, uint32_t stream) {
    const int d4 = D >> 2;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_tokens * d4) return;
    const long long tk = q / d4;
    const int d = (int)(q - tk * d4) * 4;
    long long id = ids[tk];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const uint32_t sample = sample_base + (counter ? *counter : 0u) + (uint32_t)(tk / tokens_per_sample);
    const unsigned long long e = (unsigned long long)id * D + d;
    const f32x4_t m = *reinterpret_cast<const f32x4_t*>(mu + e);
    const f32x4_t r = *reinterpret_cast<const f32x4_t*>(rho + e);
    float z[4];
    bf_normal4_dev((uint32_t)(e >> 2), (uint32_t)(e >> 34), sample, stream, k0, k1, z);
    f32x4_t w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float zr = expf(r[i]);
        const float sg = r[i] > 20.0f ? r[i] : log1pf(zr);
        w[i] = fmaf(sg, z[i], m[i]);
    }
    OT* o = out + tk * D + d;
    if constexpr (sizeof(OT) == 4) *reinterpret_cast<f32x4_t*>(o) = w;
    else if constexpr (__is_same(OT, __bf16)) *reinterpret_cast<bf16x4_t*>(o) = __builtin_convertvector(w, bf16x4_t);
    else *reinterpret_cast<f16x4_t*>(o) = __builtin_convertvector(w, f16x4_t);
}

// dmu[id] += g, drho[id] += g * eps * softplus'(rho): scatter-add with fp32 atomics (rows repeat across tokens)
template <typename GT>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long long* __restrict__ ids, const GT* __restrict__ g,
                                                            const float* __restrict__ rho, float* __restrict__ dmu,
                                                            float* __restrict__ drho, long long n_tokens,
                                                            long long tokens_per_sample, long long V, int D, uint32_t k0,
                                                            uint32_t k1, uint32_t sample_base,
                                                            const uint32_t* __restrict__ counter, uint32_t stream) {
    const int d4 = D >> 2;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_tokens * d4) return;
    const long long tk = q / d4;
    const int d = (int)(q - tk * d4) * 4;
    long long id = ids[tk];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const uint32_t sample = sample_base + (counter ? *counter : 0u) + (uint32_t)(tk / tokens_per_sample);
    const unsigned long long e = (unsigned long long)id * D + d;
    float z[4];
    bf_normal4_dev((uint32_t)(e >> 2), (uint32_t)(e >> 34), sample, stream, k0, k1, z);
    const GT* gp = g + tk * D + d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float gi = (float)gp[i];
        const float r = rho[e + i];
        const float zr = expf(r);
        if (dmu) atomicAdd(dmu + e + i, gi);
        atomicAdd(drho + e + i, gi * z[i] * (r > 20.0f ? 1.0f : zr / (zr + 1.0f)));
    }
}

}  // namespace

int bf_launch_embedding_fwd(const long long* d_ids, const float* d_mu, const float* d_rho, void* d_out, int out_dtype,
                            long long n_tokens, long long tokens_per_sample, long long V, int D, uint64_t seed,
                            uint32_t sample_base, uint32_t stream_id, hipStream_t stream) {
    if (!d_ids || !d_mu || !d_rho || !d_out) BF_FAIL("bf_embedding_fwd: NULL argument");
    if (n_tokens < 1 || tokens_per_sample < 1 || V < 1 || D < 4 || D % 4) BF_FAIL("bf_embedding_fwd: bad shape (D must be a multiple of 4)");
    if (((uintptr_t)d_mu | (uintptr_t)d_rho | (uintptr_t)d_out) & 15) BF_FAIL("bf_embedding_fwd: operands must be 16-byte aligned");
    const long long work = n_tokens * (D / 4);
    const dim3 grid((uint32_t)((work + 255) / 256));
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (out_dtype == BF_DT_BF16)
        hipLaunchKernelGGL(embedding_fwd_kernel<__bf16>, grid, dim3(256), 0, stream, d_ids, d_mu, d_rho, (__bf16*)d_out,
                           n_tokens, tokens_per_sample, V, D, k0, k1, sample_base, bf_sample_counter(), stream_id);
    else if (out_dtype == BF_DT_F16)
        hipLaunchKernelGGL(embedding_fwd_kernel<_Float16>, grid, dim3(256), 0, stream, d_ids, d_mu, d_rho, (_Float16*)d_out,
                           n_tokens, tokens_per_sample, V, D, k0, k1, sample_base, bf_sample_counter(), stream_id);
    else
        hipLaunchKernelGGL(embedding_fwd_kernel<float>, grid, dim3(256), 0, stream, d_ids, d_mu, d_rho, (float*)d_out,
                           n_tokens, tokens_per_sample, V, D, k0, k1, sample_base, bf_sample_counter(), stream_id);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_embedding_bwd(const long long* d_ids, const void* d_grad, int grad_dtype, const float* d_rho, float* d_dmu,
                            float* d_drho, long long n_tokens, long long tokens_per_sample, long long V, int D,
                            uint64_t seed, uint32_t sample_base, uint32_t stream_id, hipStream_t stream) {
    if (!d_ids || !d_grad || !d_rho || !d_drho) BF_FAIL("bf_embedding_bwd: NULL argument");
    if (n_tokens < 1 || tokens_per_sample < 1 || V < 1 || D < 4 || D % 4) BF_FAIL("bf_embedding_bwd: bad shape");
    const long long work = n_tokens * (D / 4);
    const dim3 grid((uint32_t)((work + 255) / 256));
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (grad_dtype == BF_DT_BF16)
        hipLaunchKernelGGL(embedding_bwd_kernel<__bf16>, grid, dim3(256), 0, stream, d_ids, (const __bf16*)d_grad, d_rho,
                           d_dmu, d_drho, n_tokens, tokens_per_sample, V, D, k0, k1, sample_base, bf_sample_counter(), stream_id);
    else if (grad_dtype == BF_DT_F16)
        hipLaunchKernelGGL(embedding_bwd_kernel<_Float16>, grid, dim3(256), 0, stream, d_ids, (const _Float16*)d_grad,
                           d_rho, d_dmu, d_drho, n_tokens, tokens_per_sample, V, D, k0, k1, sample_base, bf_sample_counter(), stream_id);
    else
        hipLaunchKernelGGL(embedding_bwd_kernel<float>, grid, dim3(256), 0, stream, d_ids, (const float*)d_grad, d_rho,
                           d_dmu, d_drho, n_tokens, tokens_per_sample, V, D, k0, k1, sample_base, bf_sample_counter(), stream_id);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_kl_grad(const bf_tensor_t* t, int S, uint64_t seed, uint32_t sample_base, const double* d_g,
                      float* d_dmu, float* d_drho, hipStream_t stream) {
    if (!t || !t->d_mu || !t->d_rho || !d_g || !d_drho) BF_FAIL("bf_kl_grad: NULL argument");
    if (t->n == 0 || S < 1) BF_FAIL("bf_kl_grad: empty");
    KlParams p{};
    p.mu = t->d_mu; p.rho = t->d_rho; p.mu_p = t->prior.d_mu; p.rho_p = t->prior.d_rho;
    p.g = d_g; p.dmu = d_dmu; p.drho = d_drho; p.n = t->n; p.prior = t->prior.kind; p.S = S;
    p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32); p.sample_base = sample_base; p.stream = t->stream_id;
    p.counter = bf_sample_counter();
    p.chk = bf_prior_check_of(t->prior);
    p.stale = bf_stale_counter_dev();
    if (p.prior == BF_PRIOR_MIXTURE) {
        const double pi = t->prior.pi, s1 = t->prior.sigma1, s2 = t->prior.sigma2;
        if (!(s1 > 0.0) || !(s2 > 0.0) || !(pi >= 0.0) || !(pi <= 1.0)) BF_FAIL("bf_kl_grad: bad mixture prior");
        p.a1 = (float)(-0.5 / (s1 * s1));
        p.a2 = (float)(-0.5 / (s2 * s2));
        p.b1 = (float)(log(pi) - log(s1));
        p.b2 = (float)(log1p(-pi) - log(s2));
    } else if (p.prior == BF_PRIOR_GAUSSIAN && (!p.mu_p || !p.rho_p)) {
        BF_FAIL("bf_kl_grad: gaussian prior needs d_mu/d_rho");
    }
    const uint64_t groups = (t->n + 3) / 4;
    hipLaunchKernelGGL(kl_grad_kernel, dim3((uint32_t)((groups + 255) / 256)), dim3(256), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_transpose(const void* d_in, void* d_out, int elem_size, int batch, int rows, int cols, hipStream_t stream) {
    if (!d_in || !d_out) BF_FAIL("bf_transpose: NULL argument");
    if (batch < 1 || rows < 1 || cols < 1 || batch > 65535) BF_FAIL("bf_transpose: bad shape %d x %d x %d", batch, rows, cols);
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
    if (elem_size == 2 && rows % 64 == 0 && cols % 64 == 0 && (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0)
        hipLaunchKernelGGL(transpose16_kernel<0>, grid, dim3(256), 0, stream, (const uint16_t*)d_in, (uint16_t*)d_out, rows,
                           cols, (float*)nullptr);
    else if (elem_size == 2)
        hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t*)d_in, (uint16_t*)d_out, rows, cols);
    else if (elem_size == 4)
        hipLaunchKernelGGL(transpose_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t*)d_in, (uint32_t*)d_out, rows, cols);
    else
        BF_FAIL("bf_transpose: element size %d", elem_size);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t bf_colsum_workspace_bytes(int S, int M, int N) {
    // the larger of the two partial layouts: 64-row chunks (fused into the transpose) / kColsumRows-row chunks
    return (size_t)S * ((M + 63) / 64) * N * sizeof(float);
}

bool bf_transpose_colsum_supported(int dtype, int batch, int rows, int cols, const void* d_in, const void* d_out) {
    return dtype != BF_DT_F32 && rows % 64 == 0 && cols % 64 == 0 && batch >= 1 && batch <= 65535 &&
           (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;
}

// transpose of a 16-bit [batch][rows][cols] tensor that also yields out_sums[g][c] = sum of the rows of each group of
// `batch_per_group` consecutive batch entries (d_partial: batch * rows / 64 * cols floats of scratch)
int bf_launch_transpose_colsum(const void* d_in, void* d_out, int dtype, int batch, int rows, int cols,
                               int batch_per_group, float* d_partial, float* d_out_sums, hipStream_t stream) {
    if (!d_in || !d_out || !d_partial || !d_out_sums) BF_FAIL("bf_transpose_colsum: NULL argument");
    if (!bf_transpose_colsum_supported(dtype, batch, rows, cols, d_in, d_out) || batch % batch_per_group)
        BF_FAIL("bf_transpose_colsum: unsupported shape %d x %d x %d", batch, rows, cols);
    dim3 grid(cols / 64, rows / 64, batch);
    if (dtype == BF_DT_BF16)
        hipLaunchKernelGGL(transpose16_kernel<BF_DT_BF16>, grid, dim3(256), 0, stream, (const uint16_t*)d_in, (uint16_t*)d_out,
                           rows, cols, d_partial);
    else
        hipLaunchKernelGGL(transpose16_kernel<BF_DT_F16>, grid, dim3(256), 0, stream, (const uint16_t*)d_in, (uint16_t*)d_out,
                           rows, cols, d_partial);
    const int groups = batch / batch_per_group, chunks = batch_per_group * (rows / 64);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 63) / 64, groups), dim3(256), 0, stream, d_partial, d_out_sums,
                       cols, chunks);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_gelu(const void* d_in, void* d_out, int dtype, uint64_t n, hipStream_t stream) {
    if (!d_in || !d_out) BF_FAIL("bf_gelu: NULL argument");
    const bool vec = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0 && dtype != BF_DT_F32;
    const unsigned long long n8 = vec ? n / 8 : 0;
    if (n8) {
        const unsigned grid = (unsigned)std::min<unsigned long long>((n8 + 255) / 256, 256ull * 16);
        if (dtype == BF_DT_BF16)
            hipLaunchKernelGGL((gelu_kernel<__bf16, bf16x8_t>), dim3(grid), dim3(256), 0, stream, (const __bf16*)d_in,
                               (__bf16*)d_out, n8);
        else
            hipLaunchKernelGGL((gelu_kernel<_Float16, f16x8_t>), dim3(grid), dim3(256), 0, stream, (const _Float16*)d_in,
                               (_Float16*)d_out, n8);
    }
    const unsigned long long lo = n8 * 8;
    if (lo < n) {
        const unsigned grid = (unsigned)((n - lo + 255) / 256);
        if (dtype == BF_DT_BF16)
            hipLaunchKernelGGL(gelu_tail_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, (const __bf16*)d_in, (__bf16*)d_out, lo, n);
        else if (dtype == BF_DT_F16)
            hipLaunchKernelGGL(gelu_tail_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, (const _Float16*)d_in,
                               (_Float16*)d_out, lo, n);
        else
            hipLaunchKernelGGL(gelu_tail_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)d_in, (float*)d_out, lo, n);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

bool bf_gelu_bwd_colsum_supported(int dtype, int S, int M, int N, const void* d_dy, const void* d_pre, const void* d_out) {
    return dtype != BF_DT_F32 && N % 8 == 0 && S >= 1 && S <= 65535 && M >= 1 &&
           (((uintptr_t)d_dy | (uintptr_t)d_pre | (uintptr_t)d_out) & 15) == 0;
}

int bf_launch_gelu_bwd_colsum(const void* d_dy, const void* d_pre, void* d_dpre, int dtype, int S, int M, int N,
                              float* d_partial, float* d_colsum, hipStream_t stream) {
    if (!d_dy || !d_pre || !d_dpre || !d_partial || !d_colsum) BF_FAIL("bf_gelu_bwd_colsum: NULL argument");
    if (!bf_gelu_bwd_colsum_supported(dtype, S, M, N, d_dy, d_pre, d_dpre))
        BF_FAIL("bf_gelu_bwd_colsum: needs 16-bit tensors, N %% 8 == 0 and 16-byte aligned pointers");
    const int chunks = (M + kColsumRows - 1) / kColsumRows;
    dim3 pgrid((N / 8 + 63) / 64, chunks, S);
    if (dtype == BF_DT_BF16)
        hipLaunchKernelGGL((gelu_bwd_colsum_kernel<__bf16, bf16x8_t>), pgrid, dim3(256), 0, stream, (const __bf16*)d_dy,
                           (const __bf16*)d_pre, (__bf16*)d_dpre, d_partial, M, N, chunks);
    else
        hipLaunchKernelGGL((gelu_bwd_colsum_kernel<_Float16, f16x8_t>), pgrid, dim3(256), 0, stream, (const _Float16*)d_dy,
                           (const _Float16*)d_pre, (_Float16*)d_dpre, d_partial, M, N, chunks);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64, S), dim3(256), 0, stream, d_partial, d_colsum, N, chunks);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_colsum(const void* d_dy, int dtype, float* d_out, int S, int M, int N, float* d_partial,
                     hipStream_t stream) {
    if (!d_dy || !d_out) BF_FAIL("bf_colsum: NULL argument");
    if (dtype != BF_DT_F32 && N % 8 == 0 && d_partial && ((uintptr_t)d_dy & 15) == 0 && S <= 65535) {
        const int chunks = (M + kColsumRows - 1) / kColsumRows;
        dim3 pgrid((N / 8 + 63) / 64, chunks, S);
        if (dtype == BF_DT_BF16)
            hipLaunchKernelGGL((colsum_partial_kernel<__bf16, bf16x8_t>), pgrid, dim3(256), 0, stream, (const __bf16*)d_dy,
                               d_partial, M, N, chunks);
        else
            hipLaunchKernelGGL((colsum_partial_kernel<_Float16, f16x8_t>), pgrid, dim3(256), 0, stream,
                               (const _Float16*)d_dy, d_partial, M, N, chunks);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64, S), dim3(256), 0, stream, d_partial, d_out, N, chunks);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    dim3 grid((N + 63) / 64, S);
    if (dtype == BF_DT_BF16)
        hipLaunchKernelGGL(colsum_kernel<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)d_dy, d_out, M, N);
    else if (dtype == BF_DT_F16)
        hipLaunchKernelGGL(colsum_kernel<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)d_dy, d_out, M, N);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)d_dy, d_out, M, N);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

static uint32_t pg_blocks(uint64_t n) { return (uint32_t)(((n + 3) / 4 + 255) / 256); }

size_t bf_pgrad_table_bytes(const bf_pgrad_t* t, int n, uint32_t* total_blocks) {
    uint64_t blocks = 0;
    for (int i = 0; i < n; ++i) blocks += pg_blocks(t[i].n);
    if (total_blocks) *total_blocks = (uint32_t)blocks;
    return bf_align_up((size_t)n * sizeof(PgEntry), 256) + (size_t)blocks * sizeof(uint32_t);
}

int bf_pgrad_table_build(const bf_pgrad_t* t, int n, void* h_blob, size_t blob_bytes) {
    uint32_t total = 0;
    if (blob_bytes < bf_pgrad_table_bytes(t, n, &total)) BF_FAIL("bf_param_grad_table_build: blob too small");
    PgEntry* ent = reinterpret_cast<PgEntry*>(h_blob);
    uint32_t* map = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(h_blob) + bf_align_up((size_t)n * sizeof(PgEntry), 256));
    uint32_t blk = 0;
    for (int i = 0; i < n; ++i) {
        if (!t[i].d_dw || !t[i].d_rho || !t[i].d_drho || t[i].n == 0 || t[i].splits < 1)
            BF_FAIL("bf_param_grad_table_build: entry %d: NULL pointer, empty tensor or splits < 1", i);
        PgEntry& e = ent[i];
        e.dw = t[i].d_dw; e.rho = t[i].d_rho; e.dmu = t[i].d_dmu; e.drho = t[i].d_drho;
        e.n = t[i].n; e.stream = t[i].stream_id; e.block_begin = blk; e.splits = t[i].splits;
        e.vec = (t[i].n % 4 == 0 && ((uintptr_t)t[i].d_dw & 15) == 0) ? 1 : 0;
        const uint32_t nb = pg_blocks(t[i].n);
        for (uint32_t b = 0; b < nb; ++b) map[blk + b] = (uint32_t)i;
        blk += nb;
    }
    return 0;
}

int bf_launch_pgrad_table(const void* d_blob, int n, uint32_t total_blocks, int S, uint64_t seed, uint32_t sample_base,
                          hipStream_t stream) {
    if (!d_blob || n < 1 || total_blocks < 1 || S < 1) BF_FAIL("bf_param_grad_table: empty launch");
    const PgEntry* ent = reinterpret_cast<const PgEntry*>(d_blob);
    const uint32_t* map = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(d_blob) + bf_align_up((size_t)n * sizeof(PgEntry), 256));
    hipLaunchKernelGGL(param_grad_table_kernel, dim3(total_blocks), dim3(256), 0, stream, ent, map, S, (uint32_t)seed,
                       (uint32_t)(seed >> 32), sample_base, bf_sample_counter());
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_param_grad(const float* d_dw, const float* d_rho, uint64_t n, int S, int splits, uint64_t seed,
                         uint32_t sample_base, uint32_t stream_id, float* d_dmu, float* d_drho, hipStream_t stream) {
    if (!d_dw || !d_rho || !d_drho) BF_FAIL("bf_param_grad: NULL argument");
    if (n == 0 || S < 1 || splits < 1) BF_FAIL("bf_param_grad: empty");
    const uint64_t groups = (n + 3) / 4;
    const dim3 grid((uint32_t)((groups + 255) / 256));
    if (n % 4 == 0 && ((uintptr_t)d_dw & 15) == 0)
        hipLaunchKernelGGL(param_grad_kernel<true>, grid, dim3(256), 0, stream, d_dw, d_rho, (unsigned long long)n, S, splits,
                           (uint32_t)seed, (uint32_t)(seed >> 32), sample_base, bf_sample_counter(), stream_id, d_dmu, d_drho);
    else
        hipLaunchKernelGGL(param_grad_kernel<false>, grid, dim3(256), 0, stream, d_dw, d_rho, (unsigned long long)n, S, splits,
                           (uint32_t)seed, (uint32_t)(seed >> 32), sample_base, bf_sample_counter(), stream_id, d_dmu, d_drho);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
