// bf_sample.hip — fused reparameterise + log-prob kernel for gfx950 (wave64).
//
// Replaces, per bnn.Linear.forward and per Monte-Carlo sample, the ~35 ATen launches of
//   Gaussian.sample      /root/reference/bayeformers/nn/parameters/gaussian.py:90-101  (eps draw, mu + eps*sigma)
//   Gaussian.sigma       .../gaussian.py:81-88                                          (softplus, recomputed 3x there)
//   Gaussian.log_prob    .../gaussian.py:103-116                                        (posterior, and MOPED prior)
//   ScaledGaussianMixture.log_prob  .../gaussian.py:160-171
//   the four accumulations in Linear.forward  /root/reference/bayeformers/nn/layers/linear.py:99-102
// with ONE launch over all S samples: mu/rho (and the Gaussian prior's mu/rho) are read once from HBM
// (8 or 16 B per scalar), softplus/log are evaluated once per scalar, epsilon is generated in registers from the
// Philox counter (bf_philox.h), W_s is written once as bf16/fp16/fp32 (2-4 B per scalar-sample) and the two
// log-probs are reduced wave -> block -> fixed-order partials (deterministic, fp64 at block level and above).
//
// Roofline: HBM-bound at S<=2, VALU/transcendental-bound beyond (one Philox4x32-10 call per 4 normals, one
// Box-Muller per 2, ~1 exp2 + 1 log2 per scalar-sample for the mixture prior).
//
// Numerics (documented deviations from the reference's fp32 expression order):
//   log q uses eps^2/2 instead of (W-mu)^2/(2 sigma^2)  — identical analytically, avoids the cancellation;
//   the mixture uses max + log1p(exp(-|d|)) instead of log(pi*exp(lp1) + (1-pi)*exp(lp2)), which in the
//   reference underflows to -inf for |w| >= 14.3 with sigma1 = 1; here it stays finite.
#include "bf_common.h"
#include "bf_philox.h"

namespace {

constexpr int kThreads = 256;        // 4 waves
constexpr int kElemsPerThread = 8;   // two Philox groups -> one 16-byte bf16 store per sample
constexpr int kMaxSChunk = 32;
constexpr int kMaxSeg = 2;
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

struct SegDesc {
    const float* mu;
    const float* rho;
    const float* mu_p;
    const float* rho_p;
    void* out;
    unsigned long long n;
    float a1, b1, a2, b2;  // mixture: t_i = a_i * w^2 + b_i  (natural log)
    int prior_kind;
    int out_dtype;
    uint32_t stream;
    uint32_t block_begin;
    int vec_in;   // mu/rho(/prior) 16B-aligned -> float4 loads on full chunks
    int vec_out;  // out rows 16B-aligned (n % 8 == 0 and base aligned)
};

struct SampleParams {
    SegDesc seg[kMaxSeg];
    int nseg;
    int S;
    int s_chunk;
    uint32_t k0, k1;
    uint32_t sample_base;
    uint32_t nblk;
    double* partials;  // [nblk][S][2]
};

__device__ __forceinline__ float softplus_f(float r) {
    // torch.nn.functional.softplus(beta=1, threshold=20)
    return r > 20.0f ? r : log1pf(expf(r));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int OUT_DT>
__device__ __forceinline__ void store8(void* out, unsigned long long idx, const float w[8], bool vec, int nvalid) {
    if constexpr (OUT_DT == BF_DT_BF16) {
        __bf16* o = reinterpret_cast<__bf16*>(out) + idx;
        if (vec) {
            f32x8_t v = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
            *reinterpret_cast<bf16x8_t*>(o) = __builtin_convertvector(v, bf16x8_t);
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = (__bf16)w[i];
        }
    } else if constexpr (OUT_DT == BF_DT_F16) {
        _Float16* o = reinterpret_cast<_Float16*>(out) + idx;
        if (vec) {
            f32x8_t v = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
            *reinterpret_cast<f16x8_t*>(o) = __builtin_convertvector(v, f16x8_t);
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = (_Float16)w[i];
        }
    } else {
        float* o = reinterpret_cast<float*>(out) + idx;
        if (vec) {
            *reinterpret_cast<f32x4_t*>(o) = f32x4_t{w[0], w[1], w[2], w[3]};
            *reinterpret_cast<f32x4_t*>(o + 4) = f32x4_t{w[4], w[5], w[6], w[7]};
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = w[i];
        }
    }
}

__device__ __forceinline__ void load8(const float* p, unsigned long long e0, int nvalid, bool vec, float fill,
                                      float v[8]) {
    if (vec && nvalid == 8) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p + e0);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + e0 + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = i < nvalid ? p[e0 + i] : fill;
    }
}

// grid = (nblk, ceil(S / s_chunk)); block = 256 threads; thread = 8 consecutive scalars x s_chunk samples.
template <int PRIOR, int OUT_DT, bool HAS_OUT>
__device__ __forceinline__ void sample_logprob_body(const SampleParams& p, const SegDesc& sg, float (*red)[kMaxSChunk][2]) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const unsigned long long e0 =
        ((unsigned long long)(blockIdx.x - sg.block_begin) * kThreads + tid) * kElemsPerThread;
    const int nvalid = e0 >= sg.n ? 0 : (sg.n - e0 >= 8 ? 8 : (int)(sg.n - e0));
    const int s_begin = blockIdx.y * p.s_chunk;
    const int s_count = min(p.s_chunk, p.S - s_begin);

    float mu[8], sigma[8];
    float pmu[8], pinv[8];  // gaussian prior: mean and 1/(2 sigma_p^2)
    float constq = 0.f, constp = 0.f;
    if (nvalid > 0) {
        float rho[8];
        load8(sg.mu, e0, nvalid, sg.vec_in, 0.f, mu);
        load8(sg.rho, e0, nvalid, sg.vec_in, 0.f, rho);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sigma[i] = softplus_f(rho[i]);
            if (i < nvalid) constq += -kLogSqrt2Pi - logf(sigma[i]);
        }
        if constexpr (PRIOR == BF_PRIOR_GAUSSIAN) {
            float prho[8];
            load8(sg.mu_p, e0, nvalid, sg.vec_in, 0.f, pmu);
            load8(sg.rho_p, e0, nvalid, sg.vec_in, 0.f, prho);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float sp = softplus_f(prho[i]);
                pinv[i] = 0.5f / (sp * sp);
                if (i < nvalid) constp += -kLogSqrt2Pi - logf(sp);
            }
        }
    }
    const uint32_t g_lo = (uint32_t)(e0 >> 2), g_hi = (uint32_t)(e0 >> 34);
    // (e0>>2)+1 cannot carry into the high word: e0 is a multiple of 8, so the low group index is even.

    for (int si = 0; si < s_count; ++si) {
        float lq = 0.f, lp = 0.f;
        if (nvalid > 0) {
            const uint32_t sample = p.sample_base + (uint32_t)(s_begin + si);
            float z[8], w[8];
            bf_normal4_dev(g_lo, g_hi, sample, sg.stream, p.k0, p.k1, z);
            bf_normal4_dev(g_lo + 1u, g_hi, sample, sg.stream, p.k0, p.k1, z + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w[i] = fmaf(sigma[i], z[i], mu[i]);
                float tq = -0.5f * z[i] * z[i];
                float tp = 0.f;
                if constexpr (PRIOR == BF_PRIOR_MIXTURE) {
                    const float w2 = w[i] * w[i];
                    const float t1 = fmaf(sg.a1, w2, sg.b1);
                    const float t2 = fmaf(sg.a2, w2, sg.b2);
                    const float m = fmaxf(t1, t2);
                    const float d = fabsf(t1 - t2);
                    // log(e^t1 + e^t2) = m + ln2 * log2(1 + 2^(-d*log2e))
                    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * d);
                    tp = fmaf(0.69314718055994531f, __builtin_amdgcn_logf(1.0f + e), m);
                } else if constexpr (PRIOR == BF_PRIOR_GAUSSIAN) {
                    const float dlt = w[i] - pmu[i];
                    tp = -(dlt * dlt) * pinv[i];
                }
                if (i < nvalid) {
                    lq += tq;
                    lp += tp;
                }
            }
            if constexpr (HAS_OUT) {
                store8<OUT_DT>(sg.out, (unsigned long long)(s_begin + si) * sg.n + e0, w, sg.vec_out && nvalid == 8,
                               nvalid);
            }
        }
        lq = wave_sum(lq);
        lp = wave_sum(lp);
        if (lane == 0) {
            red[wid][si][0] = lp;
            red[wid][si][1] = lq;
        }
    }
    // sample-independent parts: sum_e(-c - log sigma_e) for q, sum_e(-c - log sigma_p,e) for a Gaussian prior
    constq = wave_sum(constq);
    constp = wave_sum(constp);
    __shared__ float cst[4][2];
    if (lane == 0) {
        cst[wid][0] = constp;
        cst[wid][1] = constq;
    }
    __syncthreads();
    if (tid < 2 * s_count) {
        const int si = tid >> 1, j = tid & 1;
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += (double)red[w][si][j];
        double c = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) c += (double)cst[w][j];
        p.partials[((size_t)blockIdx.x * p.S + (s_begin + si)) * 2 + j] = acc + c;
    }
}

template <int PRIOR, int OUT_DT, bool HAS_OUT>
__device__ __forceinline__ void dispatch_body(const SampleParams& p, const SegDesc& sg, float (*red)[kMaxSChunk][2]) {
    sample_logprob_body<PRIOR, OUT_DT, HAS_OUT>(p, sg, red);
}

__global__ __launch_bounds__(kThreads) void bf_sample_logprob_kernel(const SampleParams p) {
    __shared__ float red[4][kMaxSChunk][2];
    const int si = (p.nseg > 1 && blockIdx.x >= p.seg[1].block_begin) ? 1 : 0;
    const SegDesc& sg = p.seg[si];
    const bool has_out = sg.out != nullptr;
#define BF_CASE(PR, DT)                                                  \
    if (sg.prior_kind == PR && (!has_out || sg.out_dtype == DT)) {       \
        if (has_out)                                                     \
            dispatch_body<PR, DT, true>(p, sg, red);                     \
        else                                                             \
            dispatch_body<PR, DT, false>(p, sg, red);                    \
        return;                                                          \
    }
    // without an output the dtype is irrelevant: route everything through the BF16 instantiation
    if (!has_out) {
        if (sg.prior_kind == BF_PRIOR_MIXTURE) dispatch_body<BF_PRIOR_MIXTURE, BF_DT_BF16, false>(p, sg, red);
        else if (sg.prior_kind == BF_PRIOR_GAUSSIAN) dispatch_body<BF_PRIOR_GAUSSIAN, BF_DT_BF16, false>(p, sg, red);
        else dispatch_body<BF_PRIOR_NONE, BF_DT_BF16, false>(p, sg, red);
        return;
    }
    BF_CASE(BF_PRIOR_MIXTURE, BF_DT_BF16)
    BF_CASE(BF_PRIOR_MIXTURE, BF_DT_F16)
    BF_CASE(BF_PRIOR_MIXTURE, BF_DT_F32)
    BF_CASE(BF_PRIOR_GAUSSIAN, BF_DT_BF16)
    BF_CASE(BF_PRIOR_GAUSSIAN, BF_DT_F16)
    BF_CASE(BF_PRIOR_GAUSSIAN, BF_DT_F32)
    BF_CASE(BF_PRIOR_NONE, BF_DT_BF16)
    BF_CASE(BF_PRIOR_NONE, BF_DT_F16)
    BF_CASE(BF_PRIOR_NONE, BF_DT_F32)
#undef BF_CASE
}

// out[s][j] = sum_b partials[b][s][j] in a fixed order (deterministic).  grid = 2*S blocks of 256 threads.
__global__ __launch_bounds__(256) void bf_reduce_partials_kernel(const double* __restrict__ partials, uint32_t nblk,
                                                                 int S, double* __restrict__ out) {
    __shared__ double sh[256];
    const int sj = blockIdx.x;  // s*2 + j
    double acc = 0.0;
    for (uint32_t b = threadIdx.x; b < nblk; b += 256) acc += partials[(size_t)b * S * 2 + sj];
    sh[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[sj] = sh[0];
}

__global__ __launch_bounds__(256) void bf_philox_normal_kernel(float* __restrict__ out, unsigned long long n, int S,
                                                               uint32_t k0, uint32_t k1, uint32_t sample_base,
                                                               uint32_t stream) {
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g * 4 >= n) return;
    const int s = blockIdx.y;
    float z[4];
    bf_normal4_dev((uint32_t)g, (uint32_t)(g >> 32), sample_base + (uint32_t)s, stream, k0, k1, z);
    float* o = out + (unsigned long long)s * n + g * 4;
    for (int i = 0; i < 4; ++i)
        if (g * 4 + i < n) o[i] = z[i];
    (void)S;
}

inline uint32_t blocks_for(uint64_t n) {
    const uint64_t per_block = (uint64_t)kThreads * kElemsPerThread;
    return (uint32_t)((n + per_block - 1) / per_block);
}

}  // namespace

size_t bf_sample_partials_bytes(const bf_tensor_t* tensors, int n_tensors, int S) {
    uint64_t nblk = 0;
    for (int t = 0; t < n_tensors; ++t) nblk += blocks_for(tensors[t].n);
    return bf_align_up((size_t)nblk * (size_t)S * 2 * sizeof(double), 256);
}

int bf_launch_philox_normal(float* d_out, uint64_t n, int S, uint64_t seed, uint32_t sample_base, uint32_t stream_id,
                            hipStream_t stream) {
    if (n == 0 || S <= 0) return 0;
    const uint64_t groups = (n + 3) / 4;
    dim3 grid((uint32_t)((groups + 255) / 256), (uint32_t)S);
    hipLaunchKernelGGL(bf_philox_normal_kernel, grid, dim3(256), 0, stream, d_out, (unsigned long long)n, S,
                       (uint32_t)seed, (uint32_t)(seed >> 32), sample_base, stream_id);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_sample_logprob(const bf_tensor_t* tensors, int n_tensors, int S, uint64_t seed, uint32_t sample_base,
                             double* d_logprob_out, void* d_workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n_tensors < 1 || n_tensors > kMaxSeg) BF_FAIL("bf_sample_logprob: n_tensors must be 1 or 2 (got %d)", n_tensors);
    if (S < 1) BF_FAIL("bf_sample_logprob: S must be >= 1 (got %d)", S);
    if (!d_logprob_out) BF_FAIL("bf_sample_logprob: d_logprob_out is NULL");
    const size_t need = bf_sample_partials_bytes(tensors, n_tensors, S);
    if (!d_workspace || workspace_bytes < need)
        BF_FAIL("bf_sample_logprob: workspace too small (%zu < %zu bytes)", workspace_bytes, need);

    SampleParams p{};
    p.nseg = n_tensors;
    p.S = S;
    p.k0 = (uint32_t)seed;
    p.k1 = (uint32_t)(seed >> 32);
    p.sample_base = sample_base;
    p.partials = reinterpret_cast<double*>(d_workspace);
    uint32_t blk = 0;
    for (int t = 0; t < n_tensors; ++t) {
        const bf_tensor_t& T = tensors[t];
        if (!T.d_mu || !T.d_rho) BF_FAIL("bf_sample_logprob: tensor %d has NULL mu/rho", t);
        if (T.n == 0) BF_FAIL("bf_sample_logprob: tensor %d is empty", t);
        SegDesc& sg = p.seg[t];
        sg.mu = T.d_mu;
        sg.rho = T.d_rho;
        sg.n = T.n;
        sg.out = T.d_sample_out;
        sg.out_dtype = T.out_dtype;
        sg.stream = T.stream_id;
        sg.prior_kind = T.prior.kind;
        sg.mu_p = T.prior.d_mu;
        sg.rho_p = T.prior.d_rho;
        uintptr_t align_bits = (uintptr_t)T.d_mu | (uintptr_t)T.d_rho;
        switch (T.prior.kind) {
            case BF_PRIOR_MIXTURE: {
                const double pi = T.prior.pi, s1 = T.prior.sigma1, s2 = T.prior.sigma2;
                if (!(s1 > 0.0) || !(s2 > 0.0) || !(pi >= 0.0) || !(pi <= 1.0))
                    BF_FAIL("bf_sample_logprob: bad mixture prior (pi=%g sigma1=%g sigma2=%g)", pi, s1, s2);
                sg.a1 = (float)(-0.5 / (s1 * s1));
                sg.a2 = (float)(-0.5 / (s2 * s2));
                sg.b1 = (float)(log(pi) - log(s1) - 0.91893853320467274178);
                sg.b2 = (float)(log1p(-pi) - log(s2) - 0.91893853320467274178);
                break;
            }
            case BF_PRIOR_GAUSSIAN:
                if (!T.prior.d_mu || !T.prior.d_rho) BF_FAIL("bf_sample_logprob: gaussian prior needs d_mu/d_rho");
                align_bits |= (uintptr_t)T.prior.d_mu | (uintptr_t)T.prior.d_rho;
                break;
            case BF_PRIOR_NONE:
                break;
            default:
                BF_FAIL("bf_sample_logprob: unknown prior kind %d", T.prior.kind);
        }
        if (T.d_sample_out && (T.out_dtype < BF_DT_F32 || T.out_dtype > BF_DT_F16))
            BF_FAIL("bf_sample_logprob: bad out_dtype %d", T.out_dtype);
        sg.vec_in = (align_bits & 15) == 0;
        sg.vec_out = T.d_sample_out && ((uintptr_t)T.d_sample_out & 15) == 0 && (T.n % 8) == 0;
        sg.block_begin = blk;
        blk += blocks_for(T.n);
    }
    p.nblk = blk;
    // enough (block, sample-chunk) pairs to fill 256 CUs x 8 blocks, without re-evaluating softplus more than needed
    int ny = (int)((2048 + blk - 1) / blk);
    if (ny > S) ny = S;
    if (ny < 1) ny = 1;
    int chunk = (S + ny - 1) / ny;
    if (chunk > kMaxSChunk) chunk = kMaxSChunk;
    ny = (S + chunk - 1) / chunk;
    p.s_chunk = chunk;

    hipLaunchKernelGGL(bf_sample_logprob_kernel, dim3(blk, (uint32_t)ny), dim3(kThreads), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(bf_reduce_partials_kernel, dim3((uint32_t)(2 * S)), dim3(256), 0, stream,
                       (const double*)p.partials, blk, S, d_logprob_out);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
