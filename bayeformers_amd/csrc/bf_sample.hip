// bf_sample.hip — fused reparameterise + log-prob kernel for gfx950 (wave64).
//
// Replaces, per bnn.Linear.forward and per Monte-Carlo sample, the ~35 ATen launches of
//   Gaussian.sample      /root/reference/bayeformers/nn/parameters/gaussian.py:90-101  (eps draw, mu + eps*sigma)
//   Gaussian.sigma       .../gaussian.py:81-88                                          (softplus, recomputed 3x there)
//   Gaussian.log_prob    .../gaussian.py:103-116                                        (posterior, and MOPED prior)
//   ScaledGaussianMixture.log_prob  .../gaussian.py:160-171
//   the four accumulations in Linear.forward  /root/reference/bayeformers/nn/layers/linear.py:99-102
// with ONE launch over all S samples: mu/rho (and the Gaussian prior's mu/rho) are read once from HBM with 16-byte
// loads (8 or 16 B per scalar), softplus/log are evaluated once per scalar and reused for every sample, epsilon is
// generated in registers from the Philox counter (bf_philox.h, one Philox4x32-7 block = the 4 scalars a thread
// owns), W_s is written once as bf16/fp16/fp32 and the two log-probs are reduced lane -> wave (DPP) -> block (LDS)
// -> fixed-order fp64 partials (deterministic).
//
// Roofline: HBM-bound at S <= 2; beyond that VALU/transcendental-bound (per scalar-sample: 1/4 Philox block,
// 1/2 Box-Muller, one fma, ~10 ops + 2 transcendentals for the mixture prior).
//
// Numerics (documented deviations from the reference's fp32 expression order):
//   log q uses eps^2/2 instead of (W-mu)^2/(2 sigma^2)  — identical analytically, avoids the cancellation;
//   the mixture uses max + log1p(exp(-|d|)) instead of log(pi*exp(lp1) + (1-pi)*exp(lp2)), which in the
//   reference underflows to -inf for |w| >= 14.3 with sigma1 = 1; here it stays finite;
//   softplus/log run on the hardware exp2/log2 units with an fma-compensated argument (relative error ~2e-7).
#include <string.h>

#include "bf_common.h"
#include "bf_device.h"
#include "bf_philox.h"

namespace {

constexpr int kThreads = 256;  // 4 waves
constexpr int kGPT = 2;  // groups of 4 scalars (= Philox blocks per sample) a thread owns (1 / 2 / 3 measured: profiles/r2c_*)
constexpr int kEPT = 4 * kGPT;         // scalars per thread
constexpr int kMaxSChunk = 32;
constexpr int kMaxSeg = 2;
constexpr int OUT_NONE = -1;
// A Gaussian prior whose mean IS the posterior's mean and whose sigma is one constant — MOPED with a frozen mean,
// /root/reference/bayeformers/nn/layers/linear.py:147-150 (prior.mu shares the pretrained tensor, prior.rho = 1): the
// caller asserts it through bf_prior_t (pi = 1, sigma1 = sigma_p; include/bayeformers_amd.h).  The kernel then reads
// 8 instead of 16 bytes per scalar and W - mu_p is sigma * eps: log p = -log sqrt(2 pi) - log sigma_p - (sigma eps)^2 / (2 sigma_p^2).
constexpr int PRIOR_GAUSS_ALIAS = 3;

__host__ inline int effective_prior(const bf_prior_t& pr) {
    return (pr.kind == BF_PRIOR_GAUSSIAN && pr.pi == 1.0f && pr.sigma1 > 0.0f) ? PRIOR_GAUSS_ALIAS : pr.kind;
}

struct SegDesc {
    const float* mu;
    const float* rho;
    const float* mu_p;
    const float* rho_p;
    void* out;
    unsigned long long n;
    uint32_t stream;
    uint32_t block_begin;
    int vec_in;   // mu/rho(/prior) 16B-aligned -> float4 loads on full chunks
    int vec_out;  // out rows aligned for one vector store per thread (n % 4 == 0 and base aligned)
    float rho_alias;  // asserted alias: the constant value of rho_p (spot-checked)
};

struct SampleParams {
    SegDesc seg[kMaxSeg];
    float a1, b1, a2, b2;  // mixture: t_i = a_i * w^2 + b_i  (natural log)
    int nseg;
    int S;
    int ny;  // sample chunks (gridDim.y)
    uint32_t k0, k1;
    uint32_t sample_base;
    const uint32_t* counter;  // optional device counter added to sample_base
    uint32_t nblk;
    double* partials;  // [nblk][S][2]
    bf_prior_check_t chk;  // mixture: the device scalars the constants were read from
    uint32_t* stale;       // bf_stale_counter
};


template <int OUT_DT>
__device__ __forceinline__ void store4(void* out, unsigned long long idx, const float (&w)[4], bool vec, int nvalid) {
    if constexpr (OUT_DT == BF_DT_BF16) {
        __bf16* o = reinterpret_cast<__bf16*>(out) + idx;
        if (vec) {
            *reinterpret_cast<bf16x4_t*>(o) = __builtin_convertvector((f32x4_t{w[0], w[1], w[2], w[3]}), bf16x4_t);
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = (__bf16)w[i];
        }
    } else if constexpr (OUT_DT == BF_DT_F16) {
        _Float16* o = reinterpret_cast<_Float16*>(out) + idx;
        if (vec) {
            *reinterpret_cast<f16x4_t*>(o) = __builtin_convertvector((f32x4_t{w[0], w[1], w[2], w[3]}), f16x4_t);
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = (_Float16)w[i];
        }
    } else {
        float* o = reinterpret_cast<float*>(out) + idx;
        if (vec) {
            *reinterpret_cast<f32x4_t*>(o) = f32x4_t{w[0], w[1], w[2], w[3]};
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = w[i];
        }
    }
}

__device__ __forceinline__ void load4(const float* p, unsigned long long e0, int nvalid, bool vec, float (&v)[4]) {
    if (vec && nvalid == 4) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p + e0);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = i < nvalid ? p[e0 + i] : 0.f;
    }
}

constexpr int OUT_RUNTIME = -2;

// what one block needs to know about the tensor it works on
struct BodyArgs {
    const float* mu;
    const float* rho;
    const float* mu_p;
    const float* rho_p;
    void* out;
    unsigned long long n;
    float a1, b1, a2, b2;
    uint32_t stream;
    uint32_t rel_block;  // block index within the tensor
    int vec_in, vec_out;
    int out_dt;          // used when OUT_DT == OUT_RUNTIME
    int S, ny;
    uint32_t k0, k1, sample_base;
    double* partial_row;  // this block's [S][2] row of the partials
    bf_prior_check_t chk;  // mixture constants re-check (p[0] == NULL: none)
    float rho_alias;       // asserted alias: the constant value of rho_p
    uint32_t* stale;
};

// block = 256 threads; a thread owns kGPT groups of 4 consecutive scalars (group j of thread t = group j*256 + t of the
// block: 16-byte loads stay contiguous across the wave) x the samples of chunk blockIdx.y.  Several groups per thread
// amortise the two wave reductions a sample needs (about a quarter of the per-sample instructions at one group).
// raw mu / rho of a block's full groups, loaded ahead of the block that runs before it (bf_sample_table_pair_kernel)
struct PreLoaded {
    f32x4_t mu[kGPT], rho[kGPT];
};

template <int PRIOR, int OUT_DT, bool PRE = false>
__device__ __forceinline__ void sample_body(const BodyArgs& a, float (*red)[4][kMaxSChunk][2], float (*cst)[2],
                                            const PreLoaded* pre = nullptr) {
    constexpr int G = kGPT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // balanced sample chunks: chunk y covers [y*S/ny, (y+1)*S/ny)
    const int s_begin = (int)(((long long)blockIdx.y * a.S) / a.ny);
    const int s_end = (int)(((long long)(blockIdx.y + 1) * a.S) / a.ny);

    unsigned long long e0[G];
    int nvalid[G];
    float mu[G][4], sigma[G][4];
    float pmu[G][4], pinv[G][4];  // gaussian prior: mean and 1/(2 sigma_p^2)
    float constq = 0.f, constp = 0.f;
    bool all_full = true;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        e0[j] = (((unsigned long long)a.rel_block * G + j) * kThreads + tid) * 4;
        nvalid[j] = e0[j] >= a.n ? 0 : (a.n - e0[j] >= 4 ? 4 : (int)(a.n - e0[j]));
        all_full = all_full && nvalid[j] == 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) mu[j][i] = sigma[j][i] = pmu[j][i] = pinv[j][i] = 0.f;
        if (nvalid[j] > 0) {
            float rho[4];
            if constexpr (PRE) {  // (the caller checked: every group of this block is full and 16-byte aligned)
#pragma unroll
                for (int i = 0; i < 4; ++i) mu[j][i] = pre->mu[j][i], rho[i] = pre->rho[j][i];
            } else {
                load4(a.mu, e0[j], nvalid[j], a.vec_in, mu[j]);
                load4(a.rho, e0[j], nvalid[j], a.vec_in, rho);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sigma[j][i] = softplus_fast(rho[i]);
                if (i < nvalid[j]) constq += -kLogSqrt2Pi - log_fast(sigma[j][i]);
            }
            if constexpr (PRIOR == PRIOR_GAUSS_ALIAS) constp += (float)nvalid[j] * a.b1;  // b1 = -log sqrt(2 pi) - log sigma_p
            if constexpr (PRIOR == BF_PRIOR_GAUSSIAN) {
                float prho[4];
                load4(a.mu_p, e0[j], nvalid[j], a.vec_in, pmu[j]);
                load4(a.rho_p, e0[j], nvalid[j], a.vec_in, prho);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sp = softplus_fast(prho[i]);
                    pinv[j][i] = 0.5f * __builtin_amdgcn_rcpf(sp * sp);
                    if (i < nvalid[j]) constp += -kLogSqrt2Pi - log_fast(sp);
                }
            }
        }
    }

    // The prior's baked constants against the device state they are a copy of (bf_prior_t; an in-place edit through .data
    // leaves no trace on the host): a failed check poisons this block's log-prior partial and bumps the stale counter.
    if constexpr (PRIOR == BF_PRIOR_MIXTURE) {
        if (tid == 0 && bf_prior_check_failed(a.chk)) {
            constp = __builtin_nanf("");
            bf_stale_bump(a.stale);
        }
    }
    if constexpr (PRIOR == PRIOR_GAUSS_ALIAS) {
        if (lane == 0 && nvalid[0] > 0) {  // one element per wave: prior.mu == mu and prior.rho == the asserted constant
            const float m = a.mu_p[e0[0]], r = a.rho_p[e0[0]];
            if (__float_as_uint(m) != __float_as_uint(mu[0][0]) || __float_as_uint(r) != __float_as_uint(a.rho_alias)) {
                constp = __builtin_nanf("");
                bf_stale_bump(a.stale);
            }
        }
    }

    // Fast path — every lane of the wave owns only full groups and the output takes one vector store per group: no
    // validity selects, no per-sample branches, and the per-scalar arithmetic on PAIRS (v_pk_fma_f32 / v_pk_mul_f32).
    // Only the last block of a tensor whose size is not a multiple of the block goes through the general loop below.
    const bool fast = __builtin_amdgcn_readfirstlane((int)__all(all_full && (!a.out || a.vec_out))) != 0;
    if (fast) {
        int out_dt = OUT_DT;
        if constexpr (OUT_DT == OUT_RUNTIME) out_dt = __builtin_amdgcn_readfirstlane(a.out_dt);
        char* outp = reinterpret_cast<char*>(a.out);
        bf_philox_inv inv[G];  // the sample-independent part of each group's Philox block, once
#pragma unroll
        for (int j = 0; j < G; ++j)
            inv[j] = bf_philox_prepare((uint32_t)(e0[j] >> 2), (uint32_t)(e0[j] >> 34), a.stream, a.k0, a.k1);
        for (int s = s_begin; s < s_end; ++s) {
            f32x2_t q2 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < G; ++j) {
                float z[4];
                bf_normal4_split_dev(inv[j], a.sample_base + (uint32_t)s, a.stream, a.k0, a.k1, z);
                const f32x2_t z01 = {z[0], z[1]}, z23 = {z[2], z[3]};
                const f32x2_t w01 = __builtin_elementwise_fma(f32x2_t{sigma[j][0], sigma[j][1]}, z01, f32x2_t{mu[j][0], mu[j][1]});
                const f32x2_t w23 = __builtin_elementwise_fma(f32x2_t{sigma[j][2], sigma[j][3]}, z23, f32x2_t{mu[j][2], mu[j][3]});
                q2 = __builtin_elementwise_fma(z01, z01, q2);
                q2 = __builtin_elementwise_fma(z23, z23, q2);
                if constexpr (PRIOR == BF_PRIOR_MIXTURE) {
                    const f32x2_t v01 = w01 * w01, v23 = w23 * w23;
                    const f32x2_t a1 = {a.a1, a.a1}, b1 = {a.b1, a.b1}, a2 = {a.a2, a.a2}, b2 = {a.b2, a.b2};
                    const f32x2_t t1a = __builtin_elementwise_fma(a1, v01, b1), t2a = __builtin_elementwise_fma(a2, v01, b2);
                    const f32x2_t t1b = __builtin_elementwise_fma(a1, v23, b1), t2b = __builtin_elementwise_fma(a2, v23, b2);
                    const f32x2_t ma = __builtin_elementwise_max(t1a, t2a), mb = __builtin_elementwise_max(t1b, t2b);
                    const f32x2_t da = __builtin_elementwise_abs(t1a - t2a) * (f32x2_t)(-1.4426950408889634f);
                    const f32x2_t db = __builtin_elementwise_abs(t1b - t2b) * (f32x2_t)(-1.4426950408889634f);
                    f32x2_t la, lb;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        // log(e^t1 + e^t2) = max + ln2 * log2(1 + 2^(-|t1 - t2| log2 e))
                        la[i] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(da[i]));
                        lb[i] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(db[i]));
                    }
                    p2 += __builtin_elementwise_fma((f32x2_t)(kLn2), la, ma);
                    p2 += __builtin_elementwise_fma((f32x2_t)(kLn2), lb, mb);
                } else if constexpr (PRIOR == BF_PRIOR_GAUSSIAN) {
                    const f32x2_t d01 = w01 - f32x2_t{pmu[j][0], pmu[j][1]}, d23 = w23 - f32x2_t{pmu[j][2], pmu[j][3]};
                    p2 = __builtin_elementwise_fma(d01 * d01, f32x2_t{-pinv[j][0], -pinv[j][1]}, p2);
                    p2 = __builtin_elementwise_fma(d23 * d23, f32x2_t{-pinv[j][2], -pinv[j][3]}, p2);
                } else if constexpr (PRIOR == PRIOR_GAUSS_ALIAS) {
                    // W - mu_p = sigma eps; the common factor -1 / (2 sigma_p^2) is applied once per sample below
                    const f32x2_t d01 = f32x2_t{sigma[j][0], sigma[j][1]} * z01, d23 = f32x2_t{sigma[j][2], sigma[j][3]} * z23;
                    p2 = __builtin_elementwise_fma(d01, d01, p2);
                    p2 = __builtin_elementwise_fma(d23, d23, p2);
                }
                if (outp) {
                    const unsigned long long idx = (unsigned long long)s * a.n + e0[j];
                    const f32x4_t w4 = {w01[0], w01[1], w23[0], w23[1]};
                    if (out_dt == BF_DT_BF16) *reinterpret_cast<bf16x4_t*>(outp + idx * 2) = __builtin_convertvector(w4, bf16x4_t);
                    else if (out_dt == BF_DT_F16) *reinterpret_cast<f16x4_t*>(outp + idx * 2) = __builtin_convertvector(w4, f16x4_t);
                    else *reinterpret_cast<f32x4_t*>(outp + idx * 4) = w4;
                }
            }
            // per sample only the four in-row DPP steps: the row sums of lanes 15/31/47/63 go to LDS and the block's
            // final fp64 pass adds 16 terms per scalar instead of 4 (the two cross-row steps, the readlane and their
            // moves were ~10 of the ~150 instructions a wave spends per sample)
            const float lq = row_sum(-0.5f * (q2[0] + q2[1]));
            const float lp = row_sum(PRIOR == PRIOR_GAUSS_ALIAS ? -a.a1 * (p2[0] + p2[1]) : p2[0] + p2[1]);  // a1 = 1 / (2 sigma_p^2)
            if ((lane & 15) == 15) *reinterpret_cast<f32x2_t*>(&red[wid][lane >> 4][s - s_begin][0]) = f32x2_t{lp, lq};
        }
    } else
    for (int s = s_begin; s < s_end; ++s) {
        float lq = 0.f, lp = 0.f;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (nvalid[j] <= 0) continue;
            float z[4], w[4];
            bf_normal4_dev((uint32_t)(e0[j] >> 2), (uint32_t)(e0[j] >> 34), a.sample_base + (uint32_t)s, a.stream, a.k0,
                           a.k1, z);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] = fmaf(sigma[j][i], z[i], mu[j][i]);
                const float tq = -0.5f * z[i] * z[i];
                float tp = 0.f;
                if constexpr (PRIOR == BF_PRIOR_MIXTURE) {
                    const float w2 = w[i] * w[i];
                    const float t1 = fmaf(a.a1, w2, a.b1);
                    const float t2 = fmaf(a.a2, w2, a.b2);
                    const float m = fmaxf(t1, t2);
                    const float d = fabsf(t1 - t2);
                    // log(e^t1 + e^t2) = m + ln2 * log2(1 + 2^(-d*log2e))
                    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * d);
                    tp = fmaf(kLn2, __builtin_amdgcn_logf(1.0f + e), m);
                } else if constexpr (PRIOR == BF_PRIOR_GAUSSIAN) {
                    const float dlt = w[i] - pmu[j][i];
                    tp = -(dlt * dlt) * pinv[j][i];
                } else if constexpr (PRIOR == PRIOR_GAUSS_ALIAS) {
                    const float dlt = sigma[j][i] * z[i];
                    tp = -(dlt * dlt) * a.a1;
                }
                if (i < nvalid[j]) {
                    lq += tq;
                    lp += tp;
                }
            }
            if (a.out) {
                const unsigned long long idx = (unsigned long long)s * a.n + e0[j];
                const bool vec = a.vec_out && nvalid[j] == 4;
                if constexpr (OUT_DT == OUT_RUNTIME) {
                    if (a.out_dt == BF_DT_BF16) store4<BF_DT_BF16>(a.out, idx, w, vec, nvalid[j]);
                    else if (a.out_dt == BF_DT_F16) store4<BF_DT_F16>(a.out, idx, w, vec, nvalid[j]);
                    else store4<BF_DT_F32>(a.out, idx, w, vec, nvalid[j]);
                } else if constexpr (OUT_DT != OUT_NONE) {
                    store4<OUT_DT>(a.out, idx, w, vec, nvalid[j]);
                }
            }
        }
        lq = wave_sum(lq);
        lp = wave_sum(lp);
        if (lane < 4) *reinterpret_cast<f32x2_t*>(&red[wid][lane][s - s_begin][0]) = lane == 0 ? f32x2_t{lp, lq} : f32x2_t{0.f, 0.f};
    }
    // sample-independent parts: sum_e(-c - log sigma_e) for q, sum_e(-c - log sigma_p,e) for a Gaussian prior
    constq = wave_sum(constq);
    constp = wave_sum(constp);
    if (lane == 0) {
        cst[wid][0] = constp;
        cst[wid][1] = constq;
    }
    __syncthreads();
    if (tid < 2 * (s_end - s_begin)) {
        const int sl = tid >> 1, j = tid & 1;
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += (double)red[w][r][sl][j];
            acc += (double)cst[w][j];
        }
        a.partial_row[(s_begin + sl) * 2 + j] = acc;
    }
}

// Per-layer launch: grid = (nblk, ny).  Segment 0 (the weight) is written as OUT_DT, segment 1 (the bias) as fp32.
template <int PRIOR, int OUT_DT>
__global__ __launch_bounds__(kThreads) void bf_sample_logprob_kernel(const SampleParams p) {
    __shared__ float red[4][4][kMaxSChunk][2];
    __shared__ float cst[4][2];
    const int si = (p.nseg > 1 && blockIdx.x >= p.seg[1].block_begin) ? 1 : 0;
    const SegDesc& sg = p.seg[si];
    BodyArgs a;
    a.mu = sg.mu; a.rho = sg.rho; a.mu_p = sg.mu_p; a.rho_p = sg.rho_p; a.out = sg.out; a.n = sg.n;
    a.a1 = p.a1; a.b1 = p.b1; a.a2 = p.a2; a.b2 = p.b2;
    a.stream = sg.stream; a.rel_block = blockIdx.x - sg.block_begin;
    a.vec_in = sg.vec_in; a.vec_out = sg.vec_out; a.out_dt = BF_DT_F32;
    a.S = p.S; a.ny = p.ny; a.k0 = p.k0; a.k1 = p.k1;
    a.sample_base = p.sample_base + (p.counter ? *p.counter : 0u);
    a.partial_row = p.partials + (size_t)blockIdx.x * p.S * 2;
    a.chk = p.chk; a.rho_alias = sg.rho_alias; a.stale = p.stale;
    if (si == 0) sample_body<PRIOR, OUT_DT>(a, red, cst);
    else sample_body<PRIOR, BF_DT_F32>(a, red, cst);
}

// Cross-layer launch: one grid over the blocks of MANY tensors described by a device-resident table.
struct TableEntry {
    const float* mu;
    const float* rho;
    const float* mu_p;
    const float* rho_p;
    void* out;
    unsigned long long n;
    float a1, b1, a2, b2;
    uint32_t stream;
    uint32_t block_begin;
    int prior_kind;
    int out_dt;
    int vec_in, vec_out;
    bf_prior_check_t chk;  // mixture constants re-check
    float rho_alias;       // asserted alias: the constant value of rho_p
    int pad_;
};

// ONLY >= 0: every tensor of the launch has that (effective) prior kind — the caller says so (bf_sample_logprob_table's
// prior_kinds) — and the kernel is compiled for it alone: a launch of MOPED-aliased tensors (BERT with delta and
// freeze=True: all of them) then needs 62-65 VGPRs = 8 waves per SIMD instead of the 80 / 6 of the kernel that must be able
// to take every kind.
constexpr int kOnlyWaves = 8;  // minimum waves per SIMD asked of the single-kind instantiations: 64 VGPRs (3 dwords of scratch)
// measured in the BERT-base step, one box, three interleaved runs each (profiles/r4e_sampling_kernel_occupancy_ab.txt):
// kernel for every prior kind (80 VGPRs, 6 waves) 0.529-0.542 ms, single-kind at 68 VGPRs / 7 waves 0.520-0.527, at 64 / 8 waves 0.514-0.522
template <int ONLY>
__global__ __launch_bounds__(kThreads, ONLY >= 0 ? kOnlyWaves : 1) void bf_sample_table_kernel(const TableEntry* __restrict__ table,
                                                                   const uint32_t* __restrict__ entry_of_block,
                                                                   uint32_t block0, int S, int ny, uint32_t k0,
                                                                   uint32_t k1, uint32_t sample_base,
                                                                   const uint32_t* __restrict__ counter,
                                                                   double* __restrict__ partials, uint32_t* stale) {
    __shared__ float red[4][4][kMaxSChunk][2];
    __shared__ float cst[4][2];
    const uint32_t gb = block0 + blockIdx.x;
    const TableEntry& e = table[entry_of_block[gb]];
    BodyArgs a;
    a.mu = e.mu; a.rho = e.rho; a.mu_p = e.mu_p; a.rho_p = e.rho_p; a.out = e.out; a.n = e.n;
    a.a1 = e.a1; a.b1 = e.b1; a.a2 = e.a2; a.b2 = e.b2;
    a.stream = e.stream; a.rel_block = gb - e.block_begin;
    a.vec_in = e.vec_in; a.vec_out = e.vec_out; a.out_dt = e.out_dt;
    a.S = S; a.ny = ny; a.k0 = k0; a.k1 = k1;
    a.sample_base = sample_base + (counter ? *counter : 0u);
    a.partial_row = partials + (size_t)gb * S * 2;
    a.chk = e.chk; a.rho_alias = e.rho_alias; a.stale = stale;
    if constexpr (ONLY >= 0) {
        sample_body<ONLY, OUT_RUNTIME>(a, red, cst);
        return;
    }
    const int pk = __builtin_amdgcn_readfirstlane(e.prior_kind);
    if (pk == PRIOR_GAUSS_ALIAS) sample_body<PRIOR_GAUSS_ALIAS, OUT_RUNTIME>(a, red, cst);
    else if (pk == BF_PRIOR_GAUSSIAN) sample_body<BF_PRIOR_GAUSSIAN, OUT_RUNTIME>(a, red, cst);
    else if (pk == BF_PRIOR_MIXTURE) sample_body<BF_PRIOR_MIXTURE, OUT_RUNTIME>(a, red, cst);
    else sample_body<BF_PRIOR_NONE, OUT_RUNTIME>(a, red, cst);
}


// out[g][s][j] = sum of partial rows [rows[g], rows[g+1]) in a fixed order.  grid = (2*S, G).
__global__ __launch_bounds__(256) void bf_reduce_groups_kernel(const double* __restrict__ partials,
                                                               const uint32_t* __restrict__ rows, int S,
                                                               double* __restrict__ out) {
    __shared__ double sh[256];
    const int sj = blockIdx.x, g = blockIdx.y;
    const uint32_t r0 = rows[g], r1 = rows[g + 1];
    double acc = 0.0;
    for (uint32_t b = r0 + threadIdx.x; b < r1; b += 256) acc += partials[(size_t)b * S * 2 + sj];
    sh[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)g * S * 2 + sj] = sh[0];
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
    const uint64_t per_block = (uint64_t)kThreads * kEPT;
    return (uint32_t)((n + per_block - 1) / per_block);
}

template <int PRIOR>
void launch_prior(int out_dt, dim3 grid, hipStream_t stream, const SampleParams& p) {
    switch (out_dt) {
        case BF_DT_BF16: hipLaunchKernelGGL((bf_sample_logprob_kernel<PRIOR, BF_DT_BF16>), grid, dim3(kThreads), 0, stream, p); break;
        case BF_DT_F16: hipLaunchKernelGGL((bf_sample_logprob_kernel<PRIOR, BF_DT_F16>), grid, dim3(kThreads), 0, stream, p); break;
        case BF_DT_F32: hipLaunchKernelGGL((bf_sample_logprob_kernel<PRIOR, BF_DT_F32>), grid, dim3(kThreads), 0, stream, p); break;
        default: hipLaunchKernelGGL((bf_sample_logprob_kernel<PRIOR, OUT_NONE>), grid, dim3(kThreads), 0, stream, p); break;
    }
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

static int pick_ny(uint32_t blk, int S);

// One launch covers tensors[first .. first+count) (count <= 2, same prior kind; the second is written as fp32).
static int launch_group(const bf_tensor_t* tensors, int first, int count, uint32_t blk_offset, int S, uint64_t seed,
                        uint32_t sample_base, double* partials, uint32_t nblk_total_for_layout, hipStream_t stream) {
    SampleParams p{};
    p.nseg = count;
    p.S = S;
    p.k0 = (uint32_t)seed;
    p.k1 = (uint32_t)(seed >> 32);
    p.sample_base = sample_base;
    p.counter = bf_sample_counter();
    // this group's blocks write partial rows [blk_offset, blk_offset + blk)
    p.partials = partials + (size_t)blk_offset * (size_t)S * 2;
    uint32_t blk = 0;
    const int prior_kind = effective_prior(tensors[first].prior);
    for (int t = 0; t < count; ++t) {
        const bf_tensor_t& T = tensors[first + t];
        SegDesc& sg = p.seg[t];
        sg.mu = T.d_mu;
        sg.rho = T.d_rho;
        sg.n = T.n;
        sg.out = T.d_sample_out;
        sg.stream = T.stream_id;
        sg.mu_p = T.prior.d_mu;
        sg.rho_p = T.prior.d_rho;
        uintptr_t align_bits = (uintptr_t)T.d_mu | (uintptr_t)T.d_rho;
        if (prior_kind == BF_PRIOR_GAUSSIAN) align_bits |= (uintptr_t)T.prior.d_mu | (uintptr_t)T.prior.d_rho;
        sg.vec_in = (align_bits & 15) == 0;
        const size_t osz = t == 0 ? bf_dtype_size(T.out_dtype) : 4;
        sg.vec_out = T.d_sample_out && ((uintptr_t)T.d_sample_out % (4 * osz)) == 0 && (T.n % 4) == 0;
        sg.block_begin = blk;
        sg.rho_alias = T.prior.sigma2;
        blk += blocks_for(T.n);
    }
    p.stale = bf_stale_counter_dev();
    if (prior_kind == BF_PRIOR_MIXTURE) p.chk = bf_prior_check_of(tensors[first].prior);
    if (prior_kind == BF_PRIOR_MIXTURE) {
        const double pi = tensors[first].prior.pi, s1 = tensors[first].prior.sigma1, s2 = tensors[first].prior.sigma2;
        p.a1 = (float)(-0.5 / (s1 * s1));
        p.a2 = (float)(-0.5 / (s2 * s2));
        p.b1 = (float)(log(pi) - log(s1) - 0.91893853320467274178);
        p.b2 = (float)(log1p(-pi) - log(s2) - 0.91893853320467274178);
    }
    if (prior_kind == PRIOR_GAUSS_ALIAS) {
        const double sp = tensors[first].prior.sigma1;
        p.a1 = (float)(0.5 / (sp * sp));
        p.b1 = (float)(-0.91893853320467274178 - log(sp));
    }
    p.nblk = blk;
    const int ny = pick_ny(blk, S);
    p.ny = ny;
    const dim3 grid(blk, (uint32_t)ny);
    const int out_dt = tensors[first].d_sample_out ? tensors[first].out_dtype : OUT_NONE;
    switch (prior_kind) {
        case BF_PRIOR_MIXTURE: launch_prior<BF_PRIOR_MIXTURE>(out_dt, grid, stream, p); break;
        case BF_PRIOR_GAUSSIAN: launch_prior<BF_PRIOR_GAUSSIAN>(out_dt, grid, stream, p); break;
        case PRIOR_GAUSS_ALIAS: launch_prior<PRIOR_GAUSS_ALIAS>(out_dt, grid, stream, p); break;
        default: launch_prior<BF_PRIOR_NONE>(out_dt, grid, stream, p); break;
    }
    BF_HIP_CHECK(hipGetLastError());
    (void)nblk_total_for_layout;
    return 0;
}

static int pick_ny(uint32_t blk, int S) {
    // Sample chunks per element block.  The launch runs in rounds of kResident co-resident blocks (256 CUs x 8 blocks
    // of 4 waves, <= 64 VGPRs); a block costs ~0.35 sample-equivalents for loads + softplus + log and 1 per sample.
    // Pick the ny that minimises rounds x per-block cost (avoids a nearly empty last round).
    constexpr double kResident = 2048.0, kSetup = 0.35;
    int ny = 1;
    double best = 1e300;
    for (int c = 1; c <= S; ++c) {
        const int chunk = (S + c - 1) / c;
        if (chunk > kMaxSChunk) continue;
        const double cost = ceil((double)blk * c / kResident) * (kSetup + chunk);
        if (cost < best - 1e-9) {
            best = cost;
            ny = c;
        }
    }
    return ny;
}

static int validate_tensor(const bf_tensor_t& T, int t) {
    if (!T.d_mu || !T.d_rho) BF_FAIL("bf_sample_logprob: tensor %d has NULL mu/rho", t);
    if (T.n == 0) BF_FAIL("bf_sample_logprob: tensor %d is empty", t);
    switch (T.prior.kind) {
        case BF_PRIOR_MIXTURE: {
            const double pi = T.prior.pi, s1 = T.prior.sigma1, s2 = T.prior.sigma2;
            if (!(s1 > 0.0) || !(s2 > 0.0) || !(pi >= 0.0) || !(pi <= 1.0))
                BF_FAIL("bf_sample_logprob: bad mixture prior (pi=%g sigma1=%g sigma2=%g)", pi, s1, s2);
            break;
        }
        case BF_PRIOR_GAUSSIAN:
            if (!T.prior.d_mu || !T.prior.d_rho) BF_FAIL("bf_sample_logprob: gaussian prior needs d_mu/d_rho");
            break;
        case BF_PRIOR_NONE:
            break;
        default:
            BF_FAIL("bf_sample_logprob: unknown prior kind %d", T.prior.kind);
    }
    if (T.d_sample_out && (T.out_dtype < BF_DT_F32 || T.out_dtype > BF_DT_F16))
        BF_FAIL("bf_sample_logprob: bad out_dtype %d", T.out_dtype);
    return 0;
}

size_t bf_table_blob_bytes(const bf_tensor_t* tensors, int n_tensors, uint32_t* total_blocks) {
    uint64_t blk = 0;
    for (int t = 0; t < n_tensors; ++t) blk += blocks_for(tensors[t].n);
    if (total_blocks) *total_blocks = (uint32_t)blk;
    return bf_align_up((size_t)n_tensors * sizeof(TableEntry), 256) + bf_align_up((size_t)blk * sizeof(uint32_t), 256);
}

int bf_table_build(const bf_tensor_t* tensors, int n_tensors, void* h_blob, size_t blob_bytes, uint32_t* h_block_begin,
                   int32_t* h_kinds) {
    uint32_t total = 0;
    const size_t need = bf_table_blob_bytes(tensors, n_tensors, &total);
    if (!h_blob || blob_bytes < need) BF_FAIL("bf_sample_table_build: blob too small (%zu < %zu bytes)", blob_bytes, need);
    TableEntry* ent = reinterpret_cast<TableEntry*>(h_blob);
    uint32_t* map = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(h_blob) +
                                                bf_align_up((size_t)n_tensors * sizeof(TableEntry), 256));
    uint32_t blk = 0;
    for (int t = 0; t < n_tensors; ++t) {
        const bf_tensor_t& T = tensors[t];
        if (int rc = validate_tensor(T, t)) return rc;
        TableEntry& e = ent[t];
        memset(&e, 0, sizeof(e));
        e.mu = T.d_mu; e.rho = T.d_rho; e.mu_p = T.prior.d_mu; e.rho_p = T.prior.d_rho;
        e.out = T.d_sample_out; e.n = T.n; e.stream = T.stream_id; e.block_begin = blk;
        e.prior_kind = effective_prior(T.prior); e.out_dt = T.out_dtype;
        if (h_kinds) h_kinds[t] = e.prior_kind;
        uintptr_t align_bits = (uintptr_t)T.d_mu | (uintptr_t)T.d_rho;
        if (T.prior.kind == BF_PRIOR_GAUSSIAN) align_bits |= (uintptr_t)T.prior.d_mu | (uintptr_t)T.prior.d_rho;
        e.vec_in = (align_bits & 15) == 0;
        const size_t osz = bf_dtype_size(T.out_dtype);
        e.vec_out = T.d_sample_out && ((uintptr_t)T.d_sample_out % (4 * osz)) == 0 && (T.n % 4) == 0;
        if (T.prior.kind == BF_PRIOR_MIXTURE) {
            const double pi = T.prior.pi, s1 = T.prior.sigma1, s2 = T.prior.sigma2;
            e.a1 = (float)(-0.5 / (s1 * s1));
            e.a2 = (float)(-0.5 / (s2 * s2));
            e.b1 = (float)(log(pi) - log(s1) - 0.91893853320467274178);
            e.b2 = (float)(log1p(-pi) - log(s2) - 0.91893853320467274178);
        }
        if (e.prior_kind == PRIOR_GAUSS_ALIAS) {
            const double sp = T.prior.sigma1;
            e.a1 = (float)(0.5 / (sp * sp));
            e.b1 = (float)(-0.91893853320467274178 - log(sp));
            e.rho_alias = T.prior.sigma2;
        }
        e.chk = bf_prior_check_of(T.prior);
        if (h_block_begin) h_block_begin[t] = blk;
        const uint32_t nb = blocks_for(T.n);
        for (uint32_t b = 0; b < nb; ++b) map[blk + b] = (uint32_t)t;
        blk += nb;
    }
    if (h_block_begin) h_block_begin[n_tensors] = blk;
    return 0;
}

int bf_launch_sample_table(const void* d_blob, int n_tensors, uint32_t block_begin, uint32_t block_end, int S,
                           uint64_t seed, uint32_t sample_base, double* d_partials, hipStream_t stream, int prior_kinds) {
    if (!d_blob || !d_partials) BF_FAIL("bf_sample_logprob_table: NULL blob or partials");
    if (S < 1 || block_end <= block_begin) BF_FAIL("bf_sample_logprob_table: empty launch (S=%d)", S);
    const TableEntry* ent = reinterpret_cast<const TableEntry*>(d_blob);
    const uint32_t* map = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(d_blob) +
                                                            bf_align_up((size_t)n_tensors * sizeof(TableEntry), 256));
    const uint32_t blk = block_end - block_begin;
    const int ny = pick_ny(blk, S);
#define BF_TABLE_LAUNCH(ONLY)                                                                                          \
    hipLaunchKernelGGL(bf_sample_table_kernel<ONLY>, dim3(blk, (uint32_t)ny), dim3(kThreads), 0, stream, ent, map,      \
                       block_begin, S, ny, (uint32_t)seed, (uint32_t)(seed >> 32), sample_base, bf_sample_counter(), d_partials, \
                       bf_stale_counter_dev())
    if (prior_kinds == (1 << PRIOR_GAUSS_ALIAS)) BF_TABLE_LAUNCH(PRIOR_GAUSS_ALIAS);
    else if (prior_kinds == (1 << BF_PRIOR_MIXTURE)) BF_TABLE_LAUNCH(BF_PRIOR_MIXTURE);
    else BF_TABLE_LAUNCH(-1);
#undef BF_TABLE_LAUNCH
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_reduce_partials(const double* d_partials, uint32_t nrows, int S, double* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(bf_reduce_partials_kernel, dim3((uint32_t)(2 * S)), dim3(256), 0, stream, d_partials, nrows, S,
                       d_out);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_reduce_groups(const double* d_partials, const uint32_t* d_rows, int G, int S, double* d_out,
                            hipStream_t stream) {
    if (!d_partials || !d_rows || !d_out) BF_FAIL("bf_reduce_logprob: NULL argument");
    if (G < 1 || S < 1) BF_FAIL("bf_reduce_logprob: bad G=%d S=%d", G, S);
    hipLaunchKernelGGL(bf_reduce_groups_kernel, dim3((uint32_t)(2 * S), (uint32_t)G), dim3(256), 0, stream,
                       d_partials, d_rows, S, d_out);
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
    uint32_t nblk = 0;
    for (int t = 0; t < n_tensors; ++t) {
        const bf_tensor_t& T = tensors[t];
        if (int rc = validate_tensor(T, t)) return rc;
        nblk += blocks_for(T.n);
    }
    double* partials = reinterpret_cast<double*>(d_workspace);
    // one launch when the two tensors can share a kernel instantiation: same prior kind (and identical mixture
    // constants), second tensor written as fp32 (or not at all)
    bool together = n_tensors == 2 && effective_prior(tensors[0].prior) == effective_prior(tensors[1].prior) &&
                    (!tensors[1].d_sample_out || tensors[1].out_dtype == BF_DT_F32);
    if (together && effective_prior(tensors[0].prior) == PRIOR_GAUSS_ALIAS)
        together = tensors[0].prior.sigma1 == tensors[1].prior.sigma1 && tensors[0].prior.sigma2 == tensors[1].prior.sigma2;
    if (together && tensors[0].prior.kind == BF_PRIOR_MIXTURE)
        together = tensors[0].prior.pi == tensors[1].prior.pi && tensors[0].prior.sigma1 == tensors[1].prior.sigma1 &&
                   tensors[0].prior.sigma2 == tensors[1].prior.sigma2 && tensors[0].prior.d_pi == tensors[1].prior.d_pi &&
                   tensors[0].prior.d_sigma1 == tensors[1].prior.d_sigma1 && tensors[0].prior.d_sigma2 == tensors[1].prior.d_sigma2;
    int rc;
    if (n_tensors == 1 || together) {
        rc = launch_group(tensors, 0, n_tensors, 0, S, seed, sample_base, partials, nblk, stream);
    } else {
        bf_tensor_t second = tensors[1];
        rc = launch_group(tensors, 0, 1, 0, S, seed, sample_base, partials, nblk, stream);
        if (rc) return rc;
        if (second.d_sample_out && second.out_dtype != BF_DT_F32) {
            // a lone tensor is "segment 0": written in its own dtype
            rc = launch_group(&second, 0, 1, blocks_for(tensors[0].n), S, seed, sample_base, partials, nblk, stream);
        } else {
            // written as fp32 through the segment-0 path of the F32 instantiation
            second.out_dtype = BF_DT_F32;
            rc = launch_group(&second, 0, 1, blocks_for(tensors[0].n), S, seed, sample_base, partials, nblk, stream);
        }
    }
    if (rc) return rc;
    hipLaunchKernelGGL(bf_reduce_partials_kernel, dim3((uint32_t)(2 * S)), dim3(256), 0, stream,
                       (const double*)partials, nblk, S, d_logprob_out);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
