// bf_fused_small.hip — single-kernel fused reparameterise + MFMA GEMM + log-probs for small M (M <= 64 rows per
// sample): all of Linear.forward (/root/reference/bayeformers/nn/layers/linear.py:83-104) in one launch.
//
// In this regime (BASELINE config 2 read literally: x = [32, 768]; BERT's pooler and classifier) every sampled weight
// is used by at most 4 MFMA column blocks, so the weights never need to exist in memory at all:
//   * grid = (N/16 feature blocks, S samples); block = 4 waves, each owning every 4th 32-deep slice of K;
//   * a lane loads mu/rho of its 8 consecutive k (two 16-byte loads each — a whole 128-byte row segment per
//     4 lanes), evaluates softplus, draws its 8 epsilons from two Philox blocks IN REGISTERS, forms
//     W = mu + sigma*eps, accumulates the log-prior / log-q terms, packs the 8 values to bf16/fp16 and feeds them
//     straight to v_mfma_f32_16x16x32 as the A operand (D rows = output features); the B operand is the
//     K-contiguous activation fragment read from L2;
//   * every weight-sample is generated exactly once in the whole launch, so the same pass yields exact log-probs;
//   * the four K-slices are summed through LDS, the sampled bias (one more Philox block per lane) is added, and each
//     lane stores 4 consecutive output features.
// HBM traffic = mu, rho (+ prior) once per SAMPLE from L2/Infinity Cache (the 4.7 MB of a 768x768 layer are read
// from HBM once and re-read S-1 times from cache) + x + y: latency-bound, the point is the single launch.
#include <stdlib.h>

#include "bf_common.h"
#include "bf_device.h"
#include "bf_philox.h"

namespace {


struct FusedParams {
    const void* x;
    long long x_sstride;
    void* y;
    const float* mu_w;
    const float* rho_w;
    const float* mu_pw;
    const float* rho_pw;
    const float* mu_b;
    const float* rho_b;
    const float* mu_pb;
    const float* rho_pb;
    double* partials;  // [gridDim.x][S][2]
    const uint32_t* counter;  // optional device counter added to sample_base
    float a1, b1, a2, b2;      // weight mixture constants
    float ba1, bb1, ba2, bb2;  // bias mixture constants
    int prior_w, prior_b;
    int M, N, K, S;
    uint32_t k0, k1, sample_base, stream_w, stream_b;
    bf_prior_check_t chk_w, chk_b;  // mixture constants against the device scalars they were read from (bf_prior_t)
    uint32_t* stale;                // bf_stale_counter
};

template <typename T>
struct Mf;
template <>
struct Mf<__bf16> {
    using frag = bf16x8_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mf<_Float16> {
    using frag = f16x8_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};


// log prior of one sampled value (natural log), prior kind is wave-uniform
__device__ __forceinline__ float prior_term(int kind, float w, float a1, float b1, float a2, float b2, float pmu,
                                            float pinv, float plogc) {
    if (kind == BF_PRIOR_MIXTURE) {
        const float w2 = w * w;
        const float t1 = fmaf(a1, w2, b1), t2 = fmaf(a2, w2, b2);
        const float m = fmaxf(t1, t2), d = fabsf(t1 - t2);
        return fmaf(kLn2, __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * d)), m);
    }
    if (kind == BF_PRIOR_GAUSSIAN) {
        const float dl = w - pmu;
        return plogc - dl * dl * pinv;
    }
    return 0.f;
}

template <typename XT, typename T>
__device__ __forceinline__ typename Mf<T>::frag load_xfrag(const XT* p) {
    using frag = typename Mf<T>::frag;
    if constexpr (sizeof(XT) == 2) {
        return *reinterpret_cast<const frag*>(p);
    } else {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4);
        const f32x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_convertvector(v, frag);
    }
}

template <typename T, typename XT, int MB, int NW>
__global__ __launch_bounds__(NW * 64) void fused_small_kernel(const FusedParams p) {
    using frag = typename Mf<T>::frag;
    __shared__ float red[NW][MB][64][4];
    __shared__ float lpsum[NW][2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.y, n0 = blockIdx.x * 16;
    const int M = p.M, N = p.N, K = p.K;
    const uint32_t sample = p.sample_base + (p.counter ? *p.counter : 0u) + (uint32_t)s;

    const int nrow = n0 + (lane & 15);        // the weight row this lane generates
    const bool nvalid = nrow < N;
    const int kq = (lane >> 4) * 8;           // its 8 consecutive k inside a 32-deep slice
    const XT* x = reinterpret_cast<const XT*>(p.x) + (long long)s * p.x_sstride;

    f32x4_t acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float lq = 0.f, lp = 0.f;
    const int pk = p.prior_w;

    // operands of one 32-deep K slice, software-pipelined one slice ahead (this kernel is latency-, not
    // bandwidth-bound: ~2 waves per SIMD, so the next slice's loads must be in flight while this one is computed)
    struct Slice {
        f32x4_t m0, m1, r0, r1, a0, a1, c0, c1;
        frag xf[MB];
    };
    auto load_slice = [&](int kb, Slice& sl) {
        const int k = kb * 32 + kq;
        if (nvalid) {
            const float* pm = p.mu_w + (long long)nrow * K + k;
            const float* pr = p.rho_w + (long long)nrow * K + k;
            sl.m0 = *reinterpret_cast<const f32x4_t*>(pm);
            sl.m1 = *reinterpret_cast<const f32x4_t*>(pm + 4);
            sl.r0 = *reinterpret_cast<const f32x4_t*>(pr);
            sl.r1 = *reinterpret_cast<const f32x4_t*>(pr + 4);
            if (pk == BF_PRIOR_GAUSSIAN) {
                const float* qm = p.mu_pw + (long long)nrow * K + k;
                const float* qr = p.rho_pw + (long long)nrow * K + k;
                sl.a0 = *reinterpret_cast<const f32x4_t*>(qm);
                sl.a1 = *reinterpret_cast<const f32x4_t*>(qm + 4);
                sl.c0 = *reinterpret_cast<const f32x4_t*>(qr);
                sl.c1 = *reinterpret_cast<const f32x4_t*>(qr + 4);
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = min(mb * 16 + (lane & 15), M - 1);
            sl.xf[mb] = load_xfrag<XT, T>(x + (long long)m * K + k);
        }
    };
    auto compute_slice = [&](int kb, const Slice& sl) {
        const int k = kb * 32 + kq;
        // epsilon: element index e = nrow*K + k .. +7 -> Philox groups e>>2 and (e>>2)+1
        const unsigned long long e = (unsigned long long)nrow * K + k;
        float z[8];
        bf_normal4_dev((uint32_t)(e >> 2), (uint32_t)(e >> 34), sample, p.stream_w, p.k0, p.k1, z);
        bf_normal4_dev((uint32_t)((e >> 2) + 1), (uint32_t)(((e >> 2) + 1) >> 32), sample, p.stream_w, p.k0, p.k1, z + 4);
        f32x8_t wv;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float w = 0.f;
            if (nvalid) {
                const float mu = i < 4 ? sl.m0[i & 3] : sl.m1[i & 3];
                const float rho = i < 4 ? sl.r0[i & 3] : sl.r1[i & 3];
                const float sg = softplus_fast(rho);
                w = fmaf(sg, z[i], mu);
                lq += -kLogSqrt2Pi - log_fast(sg) - 0.5f * z[i] * z[i];
                float pmu = 0.f, pinv = 0.f, plogc = 0.f;
                if (pk == BF_PRIOR_GAUSSIAN) {
                    pmu = i < 4 ? sl.a0[i & 3] : sl.a1[i & 3];
                    const float sp = softplus_fast(i < 4 ? sl.c0[i & 3] : sl.c1[i & 3]);
                    pinv = 0.5f * __builtin_amdgcn_rcpf(sp * sp);
                    plogc = -kLogSqrt2Pi - log_fast(sp);
                }
                lp += prior_term(pk, w, p.a1, p.b1, p.a2, p.b2, pmu, pinv, plogc);
            }
            wv[i] = w;
        }
        const frag wf = __builtin_convertvector(wv, frag);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = Mf<T>::run(wf, sl.xf[mb], acc[mb]);
    };

    const int nkb = K / 32;
    if (wid < nkb) {
        Slice cur, nxt;
        load_slice(wid, cur);
        for (int kb = wid; kb < nkb; kb += NW) {
            const bool more = kb + NW < nkb;
            if (more) load_slice(kb + NW, nxt);
            compute_slice(kb, cur);
            if (more) cur = nxt;
        }
    }

    // ---- sum the four K-slices; wave w finishes m-block w
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<f32x4_t*>(&red[wid][mb][lane][0]) = acc[mb];
    lq = wave_sum(lq);
    lp = wave_sum(lp);
    if (lane == 0) {
        lpsum[wid][0] = lp;
        lpsum[wid][1] = lq;
    }
    __syncthreads();

    // sampled bias of this lane's 4 output features n0 + (lane>>4)*4 + j: exactly one Philox block
    const int nb4 = n0 + (lane >> 4) * 4;
    f32x4_t bias = {0.f, 0.f, 0.f, 0.f};
    float blq = 0.f, blp = 0.f;
    if (p.mu_b) {
        float z[4];
        bf_normal4_dev((uint32_t)(nb4 >> 2), 0u, sample, p.stream_b, p.k0, p.k1, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (nb4 + j < N) {
                const float sg = softplus_fast(p.rho_b[nb4 + j]);
                const float b = fmaf(sg, z[j], p.mu_b[nb4 + j]);
                bias[j] = b;
                blq += -kLogSqrt2Pi - log_fast(sg) - 0.5f * z[j] * z[j];
                float pmu = 0.f, pinv = 0.f, plogc = 0.f;
                if (p.prior_b == BF_PRIOR_GAUSSIAN) {
                    const float sp = softplus_fast(p.rho_pb[nb4 + j]);
                    pmu = p.mu_pb[nb4 + j];
                    pinv = 0.5f * __builtin_amdgcn_rcpf(sp * sp);
                    plogc = -kLogSqrt2Pi - log_fast(sp);
                }
                blp += prior_term(p.prior_b, b, p.ba1, p.bb1, p.ba2, p.bb2, pmu, pinv, plogc);
            }
        }
    }
    for (int mb = wid; mb < MB; mb += NW) {
        f32x4_t v = bias;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += *reinterpret_cast<const f32x4_t*>(&red[w][mb][lane][0]);
        const int m = mb * 16 + (lane & 15);
        if (m < M) {
            using YT = XT;
            YT* o = reinterpret_cast<YT*>(p.y) + ((long long)s * M + m) * N + nb4;
            if (nb4 + 3 < N && (N % 4) == 0) {
                if constexpr (sizeof(YT) == 4) *reinterpret_cast<f32x4_t*>(o) = v;
                else if constexpr (__is_same(YT, __bf16)) *reinterpret_cast<bf16x4_t*>(o) = __builtin_convertvector(v, bf16x4_t);
                else *reinterpret_cast<f16x4_t*>(o) = __builtin_convertvector(v, f16x4_t);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (nb4 + j < N) o[j] = (YT)v[j];
            }
        }
    }
    // ---- log-probs of this (feature block, sample): weights from all four waves + the bias (counted once)
    if (wid == 0) {
        // each bias feature is held by 16 lanes (lane & 15): count it on lanes 0, 16, 32, 48 only
        const float bq = wave_sum((lane & 15) == 0 ? blq : 0.f);
        const float bp = wave_sum((lane & 15) == 0 ? blp : 0.f);
        if (lane == 0) {
            double tp = (double)bp, tq = (double)bq;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                tp += (double)lpsum[w][0];
                tq += (double)lpsum[w][1];
            }
            if (bf_prior_check_failed(p.chk_w) || bf_prior_check_failed(p.chk_b)) {  // stale constants: poison, report
                tp = (double)__builtin_nanf("");
                bf_stale_bump(p.stale);
            }
            double* row = p.partials + ((size_t)blockIdx.x * p.S + s) * 2;
            row[0] = tp;
            row[1] = tq;
        }
    }
}

template <typename T, typename XT, int NW>
int launch_nw(const FusedParams& p, int MB, dim3 grid, hipStream_t stream) {
    switch (MB) {
        case 1: hipLaunchKernelGGL((fused_small_kernel<T, XT, 1, NW>), grid, dim3(NW * 64), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((fused_small_kernel<T, XT, 2, NW>), grid, dim3(NW * 64), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((fused_small_kernel<T, XT, 3, NW>), grid, dim3(NW * 64), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((fused_small_kernel<T, XT, 4, NW>), grid, dim3(NW * 64), 0, stream, p); break;
        case 5: case 6: hipLaunchKernelGGL((fused_small_kernel<T, XT, 6, NW>), grid, dim3(NW * 64), 0, stream, p); break;
        default: hipLaunchKernelGGL((fused_small_kernel<T, XT, 8, NW>), grid, dim3(NW * 64), 0, stream, p); break;
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, typename XT>
int launch_mb(const FusedParams& p, int MB, dim3 grid, hipStream_t stream) {
    // K is split over the waves of a block: 8 waves when that still leaves >= 2 slices per wave and the grid alone
    // would not give each SIMD ~4 waves
    const int nkb = p.K / 32;
    const int nw = (nkb >= 16 && (long long)grid.x * grid.y < 2048) ? 8 : 4;
    if (nw == 8) return launch_nw<T, XT, 8>(p, MB, grid, stream);
    return launch_nw<T, XT, 4>(p, MB, grid, stream);
}

void mixture_consts(const bf_prior_t& pr, float& a1, float& b1, float& a2, float& b2) {
    const double pi = pr.pi, s1 = pr.sigma1, s2 = pr.sigma2;
    a1 = (float)(-0.5 / (s1 * s1));
    a2 = (float)(-0.5 / (s2 * s2));
    b1 = (float)(log(pi) - log(s1) - 0.91893853320467274178);
    b2 = (float)(log1p(-pi) - log(s2) - 0.91893853320467274178);
}

}  // namespace

// Rows per sample up to which the single fused kernel is dispatched for an N x K layer.  The kernel itself takes up to 128
// (8 MFMA column blocks per sampled weight fragment); the measured crossover against sampling launch + tiled GEMM
// (profiles/r4b_mid_m_crossover.txt, r4c_mlp_fused_threshold_ab.txt): up to 64 rows it wins or ties at every layer size
// (one launch instead of three), from 65 to 128 rows only for layers of at most 512 x 512 weights (BASELINE configs[0]'s
// hidden layers) — wider layers re-read x once per 16 output features and fall behind the tiled GEMM.
// bf_set_fused_small_max_rows() caps both (tools/crossover_bench.py, BF_FUSED_SMALL_MAX_ROWS).
static int g_fused_small_max_rows = 128;
extern "C" int bf_fused_small_max_rows(void) { return g_fused_small_max_rows; }
extern "C" int bf_set_fused_small_max_rows(int rows) {
    if (rows < 0 || rows > 128) BF_FAIL("bf_set_fused_small_max_rows: 0 .. 128 (got %d)", rows);
    g_fused_small_max_rows = rows;
    return 0;
}
extern "C" int bf_fused_small_rows_for(int N, int K) {
    const int by_size = (long long)N * K <= 512ll * 512ll ? 128 : 64;
    return by_size < g_fused_small_max_rows ? by_size : g_fused_small_max_rows;
}

// Is the single-kernel path applicable?  (M <= bf_fused_small_max_rows(), K % 32 == 0, 16-byte aligned operands, 16-bit MFMA operands.)
bool bf_fused_small_supported(int x_dtype, int y_dtype, int compute_dtype, int64_t x_sample_stride, const void* d_x,
                              const bf_tensor_t* weight, const bf_tensor_t* bias, int S, int M, int N, int K) {
    if (compute_dtype != BF_DT_BF16 && compute_dtype != BF_DT_F16) return false;
    if (x_dtype != compute_dtype && x_dtype != BF_DT_F32) return false;
    if (y_dtype != x_dtype) return false;
    if (M < 1 || M > bf_fused_small_rows_for(N, K) || K % 32 != 0 || S > 65535) return false;
    const size_t xs = bf_dtype_size(x_dtype);
    uintptr_t bits = (uintptr_t)d_x | (uintptr_t)weight->d_mu | (uintptr_t)weight->d_rho | (uintptr_t)((size_t)x_sample_stride * xs);
    if (weight->prior.kind == BF_PRIOR_GAUSSIAN) bits |= (uintptr_t)weight->prior.d_mu | (uintptr_t)weight->prior.d_rho;
    if (bits & 15) return false;
    if (((size_t)K * xs) % 16 != 0) return false;
    (void)bias;
    return true;
}

size_t bf_fused_small_partial_rows(int N) { return (size_t)(N + 15) / 16; }

int bf_launch_fused_small(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                          const bf_tensor_t* bias, void* d_y, int compute_dtype, int S, int M, int N, int K, uint64_t seed,
                          uint32_t sample_base, double* d_partials, hipStream_t stream) {
    FusedParams p{};
    p.x = d_x;
    p.x_sstride = x_sample_stride;
    p.y = d_y;
    p.mu_w = weight->d_mu;
    p.rho_w = weight->d_rho;
    p.mu_pw = weight->prior.d_mu;
    p.rho_pw = weight->prior.d_rho;
    p.prior_w = weight->prior.kind;
    if (p.prior_w == BF_PRIOR_MIXTURE) mixture_consts(weight->prior, p.a1, p.b1, p.a2, p.b2);
    p.chk_w = bf_prior_check_of(weight->prior);
    p.chk_b = bf_prior_check_t{};
    p.stale = bf_stale_counter_dev();
    p.prior_b = BF_PRIOR_NONE;
    if (bias) {
        p.mu_b = bias->d_mu;
        p.rho_b = bias->d_rho;
        p.mu_pb = bias->prior.d_mu;
        p.rho_pb = bias->prior.d_rho;
        p.prior_b = bias->prior.kind;
        if (p.prior_b == BF_PRIOR_MIXTURE) mixture_consts(bias->prior, p.ba1, p.bb1, p.ba2, p.bb2);
        p.chk_b = bf_prior_check_of(bias->prior);
        p.stream_b = bias->stream_id;
    }
    p.partials = d_partials;
    p.M = M; p.N = N; p.K = K; p.S = S;
    p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
    p.sample_base = sample_base;
    p.counter = bf_sample_counter();
    p.stream_w = weight->stream_id;
    const dim3 grid((uint32_t)((N + 15) / 16), (uint32_t)S);
    const int MB = (M + 15) / 16;
    if (compute_dtype == BF_DT_BF16) {
        if (x_dtype == BF_DT_F32) return launch_mb<__bf16, float>(p, MB, grid, stream);
        return launch_mb<__bf16, __bf16>(p, MB, grid, stream);
    }
    if (x_dtype == BF_DT_F32) return launch_mb<_Float16, float>(p, MB, grid, stream);
    return launch_mb<_Float16, _Float16>(p, MB, grid, stream);
}
