// bf_norm.hip — residual add + LayerNorm in one pass, the op that consumes a Bayesian dense layer's output in the
// transformer blocks the reference converts (HF BertSelfOutput / BertOutput: LayerNorm(dropout(dense(h)) + input),
// called around /root/reference/bayeformers/nn/layers/linear.py:83-104's forward).  Not a reference function: it
// replaces two framework kernels (elementwise add, then layer_norm) that re-read the GEMM output from HBM.
//
// HBM-bound streaming kernel: one wave64 per row, the row lives in registers (fp32) between the load and the
// store, statistics by the two-pass formula on those registers (mean, then sum of squared deviations), wave
// reductions on the DPP network.  Algorithmic bytes per row: N * (2 reads + 1 write) * sizeof(T).
#include "bf_common.h"
#include "bf_device.h"
#include "bf_philox.h"

namespace {


// 8 consecutive elements <-> 8 floats
__device__ __forceinline__ void load8(const __bf16* p, float (&v)[8]) {
    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load8(const _Float16* p, float (&v)[8]) {
    const f16x8_t t = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i], v[4 + i] = b[i];
}
// The normalised rows are written through (sc0 sc1).  Measured on the whole BERT-base step, three interleaved rounds on one
// box (profiles/r3k_layernorm_store_policy.txt): nontemporal 8.81-8.87 ms, plain 8.78-8.79, sc1 8.71-8.77, sc0 sc1 8.70-8.75:
// write-through rows are what the GEMM that reads them next (cold, from another XCD's point of view) finds fastest.
__device__ __forceinline__ void st16(f32x4_t* p, f32x4_t v) {
    // (inline asm: there is no builtin for a flat-addressed store with these cache bits.  The trailing s_nop keeps the
    // compiler's next instruction from overwriting the data registers before the store has read them — it does not pad
    // hazards of instructions inside an asm statement)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store8(__bf16* p, const float (&v)[8]) {
    const bf16x8_t t = __builtin_convertvector((f32x8_t{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}), bf16x8_t);
    st16(reinterpret_cast<f32x4_t*>(p), __builtin_bit_cast(f32x4_t, t));
}
__device__ __forceinline__ void store8(_Float16* p, const float (&v)[8]) {
    const f16x8_t t = __builtin_convertvector((f32x8_t{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}), f16x8_t);
    st16(reinterpret_cast<f32x4_t*>(p), __builtin_bit_cast(f32x4_t, t));
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    st16(reinterpret_cast<f32x4_t*>(p), f32x4_t{v[0], v[1], v[2], v[3]});
    st16(reinterpret_cast<f32x4_t*>(p + 4), f32x4_t{v[4], v[5], v[6], v[7]});
}

constexpr int kRowsPerBlock = 4;  // one wave per row

// hidden dropout of the dense output (HF BertSelfOutput / BertOutput: LayerNorm(dropout(dense(h)) + input), training mode):
// the 8-element vector `vec` (= 8 consecutive features) of row `row` is dropout group row * (N / 8) + vec of bf_philox.h
__device__ __forceinline__ uint32_t drop8_bits(const bf_dropout_t& d, long long row, int nvec, int vec) {
    const unsigned long long g = (unsigned long long)row * (unsigned)nvec + (unsigned)vec + (((unsigned long long)d.g0_hi << 32) | d.g0_lo);
    return bf_dropout_keep8((uint32_t)g, (uint32_t)(g >> 32), bf_dropout_call(d), d.site, d.k0, d.k1, d.thresh);
}
__device__ __forceinline__ void drop8_apply(const bf_dropout_t& d, uint32_t keep, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ((keep >> i) & 1u) ? v[i] * d.inv_keep : 0.f;
}
__device__ __forceinline__ void drop8(const bf_dropout_t& d, long long row, int nvec, int vec, float (&v)[8]) {
    drop8_apply(d, drop8_bits(d, row, nvec, vec), v);
}

// VPL = 8-element vectors per lane: a row has N/8 <= 64*VPL of them
template <typename T, typename GT, int VPL, bool DROP = false>
__global__ __launch_bounds__(64 * kRowsPerBlock) void add_layernorm_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const GT* __restrict__ gamma, const GT* __restrict__ beta,
    T* __restrict__ out, long long rows, int N, float eps, const bf_dropout_t drop) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = N >> 3;
    const T* xr = x + row * N;
    const T* rr = res ? res + row * N : nullptr;
    float v[VPL][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int vi = lane + 64 * c;
        if (vi < nvec) {
            load8(xr + vi * 8, v[c]);
            if constexpr (DROP) drop8(drop, row, nvec, vi, v[c]);
            if (rr) {
                float r[8];
                load8(rr + vi * 8, r);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[c][i] += r[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += v[c][i];
        }
    }
    const float inv_n = 1.0f / (float)N;
    const float mean = wave_sum(sum) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        if (lane + 64 * c < nvec) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = v[c][i] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_n + eps);
    T* orow = out + row * N;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int vi = lane + 64 * c;
        if (vi < nvec) {
            float g[8], b[8], o[8];
            load8(gamma + vi * 8, g);
            load8(beta + vi * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf((v[c][i] - mean) * rstd, g[i], b[i]);
            store8(orow + vi * 8, o);
        }
    }
}

// Rows of 32 V vectors (N = 256 V: 768, 1024): HALF a wave per row, V vectors per lane, so every lane of every load,
// store and arithmetic instruction is busy (one wave per row leaves a quarter of the lanes of N = 768 idle), two rows
// per wave.
template <typename T, typename GT, int V, bool DROP = false>
__global__ __launch_bounds__(64 * kRowsPerBlock) void add_layernorm_half_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const GT* __restrict__ gamma, const GT* __restrict__ beta,
    T* __restrict__ out, long long rows, int N, float eps, const bf_dropout_t drop) {
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const long long row_raw = ((long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (row_raw - (lane >> 5) >= rows) return;           // the whole wave is past the end
    const bool live = row_raw < rows;
    const long long row = live ? row_raw : rows - 1;      // the idle half of the last wave reads a valid row
    const T* xr = x + row * N;
    const T* rr = res ? res + row * N : nullptr;
    float v[V][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < V; ++c) {
        const int vi = hl + 32 * c;
        load8(xr + vi * 8, v[c]);
        if constexpr (DROP) drop8(drop, row, N >> 3, vi, v[c]);
        if (rr) {
            float r[8];
            load8(rr + vi * 8, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] += r[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[c][i];
    }
    const float inv_n = 1.0f / (float)N;
    const float mean = half_sum(sum, lane) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < V; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = v[c][i] - mean;
            sq = fmaf(d, d, sq);
        }
    const float rstd = 1.0f / sqrtf(half_sum(sq, lane) * inv_n + eps);
    if (!live) return;
    T* orow = out + row * N;
#pragma unroll
    for (int c = 0; c < V; ++c) {
        const int vi = hl + 32 * c;
        float g[8], b[8], o[8];
        load8(gamma + vi * 8, g);
        load8(beta + vi * 8, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaf((v[c][i] - mean) * rstd, g[i], b[i]);
        store8(orow + vi * 8, o);
    }
}

// Backward of out = LayerNorm(x + residual) * gamma + beta.  z = x + residual and its row statistics are recomputed
// (nothing but the forward's inputs is kept), then with zh = (z - mean) * rstd and a = dy * gamma:
//     dz = rstd * (a - mean_row(a) - zh * mean_row(a * zh))      (= dx = dresidual)
//     dgamma = sum_rows dy * zh,   dbeta = sum_rows dy
// One wave per row, a workgroup walks rows blockIdx, blockIdx + gridDim, ...; every lane keeps the dgamma / dbeta
// partial sums of its own columns in registers, the 4 waves of a workgroup are combined through LDS and leave one
// [2][N] fp32 row per workgroup, reduced in a fixed order by layernorm_param_grad_kernel (deterministic).
// DROP: z = dropout(x) + residual with the forward's keep-mask regenerated (bf_philox.h); dz is the gradient of the
// residual, and dx = dz o keep / (1 - p) goes to its own tensor.
// With dropout the kernel needs 132 VGPRs = 3 waves per SIMD; asked for 4 it fits in 128 (5 spilled dwords) and this
// latency-bound pass runs 109 -> 79 us (two consumers: 129 -> 92 us) at the BERT-base training shape (tools/ln_bwd_bench.py, round 5).
// (Folding the next dense layer's bias column sums into this kernel was built and measured in round 5: 16 more accumulators cost a
// wave per SIMD again (+25-45 us), LDS ds_add_f32 accumulation 190-207 us — a column-sum pass of its own, 16 us, stays cheaper.)
template <typename T, typename GT, int VPL, bool DROP = false>
__global__ __launch_bounds__(64 * kRowsPerBlock, (DROP && VPL <= 2) ? 4 : 1) void add_layernorm_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const GT* __restrict__ gamma, const T* __restrict__ dy,
    T* __restrict__ dz, float* __restrict__ partial, long long rows, int N, float eps, const bf_dropout_t drop,
    T* __restrict__ dx, const T* __restrict__ dy2) {
    extern __shared__ float sh[];  // [kRowsPerBlock][2][N]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = N >> 3;
    float gm[VPL][8], dgam[VPL][8], dbet[VPL][8];
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int vi = lane + 64 * c;
#pragma unroll
        for (int i = 0; i < 8; ++i) gm[c][i] = dgam[c][i] = dbet[c][i] = 0.f;
        if (vi < nvec) load8(gamma + vi * 8, gm[c]);
    }
    const float inv_n = 1.0f / (float)N;
    for (long long row = (long long)blockIdx.x * kRowsPerBlock + wave; row < rows; row += (long long)gridDim.x * kRowsPerBlock) {
        float v[VPL][8], g[VPL][8];
        uint32_t kept[DROP ? VPL : 1];  // the row's keep decisions: drawn once, used on load and on store
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < VPL; ++c) {
            const int vi = lane + 64 * c;
            if (vi < nvec) {
                load8(x + row * N + vi * 8, v[c]);
                if constexpr (DROP) {
                    kept[c] = drop8_bits(drop, row, nvec, vi);
                    drop8_apply(drop, kept[c], v[c]);
                }
                if (res) {
                    float r[8];
                    load8(res + row * N + vi * 8, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[c][i] += r[i];
                }
                load8(dy + row * N + vi * 8, g[c]);
                if (dy2) {  // the output had two consumers: their gradients are summed here, in fp32, not by a pass of their own
                    float h[8];
                    load8(dy2 + row * N + vi * 8, h);
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[c][i] += h[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += v[c][i];
            }
        }
        const float mean = wave_sum(sum) * inv_n;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < VPL; ++c)
            if (lane + 64 * c < nvec)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[c][i] -= mean;
                    sq = fmaf(v[c][i], v[c][i], sq);
                }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_n + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < VPL; ++c)
            if (lane + 64 * c < nvec)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[c][i] *= rstd;  // zh
                    dbet[c][i] += g[c][i];
                    dgam[c][i] = fmaf(g[c][i], v[c][i], dgam[c][i]);
                    g[c][i] *= gm[c][i];  // a
                    s1 += g[c][i];
                    s2 = fmaf(g[c][i], v[c][i], s2);
                }
        s1 = wave_sum(s1) * inv_n;
        s2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int c = 0; c < VPL; ++c) {
            const int vi = lane + 64 * c;
            if (vi < nvec) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = rstd * (g[c][i] - s1 - v[c][i] * s2);
                store8(dz + row * N + vi * 8, o);
                if constexpr (DROP) {
                    drop8_apply(drop, kept[c], o);  // the same groups, the same decisions: dx = dz o keep / (1 - p)
                    store8(dx + row * N + vi * 8, o);
                }
            }
        }
    }
    // combine the workgroup's 4 waves
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int vi = lane + 64 * c;
        if (vi < nvec)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sh[(wave * 2 + 0) * N + vi * 8 + i] = dgam[c][i];
                sh[(wave * 2 + 1) * N + vi * 8 + i] = dbet[c][i];
            }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < 2 * N; n += blockDim.x) {
        const int which = n / N, col = n - which * N;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < kRowsPerBlock; ++w) acc += sh[(w * 2 + which) * N + col];
        partial[((long long)blockIdx.x * 2 + which) * N + col] = acc;
    }
}

__global__ __launch_bounds__(1024) void layernorm_param_grad_kernel(const float* __restrict__ partial, int nblocks, int N,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // block = 16 columns x 64 lanes over the workgroup axis (2 N / 16 blocks: 96 for N = 768 — with 64 columns per
    // block the 24 blocks of a launch took 21 us for 6 MB of partials); fixed order -> deterministic
    __shared__ float sh[64][16];
    const int which = blockIdx.y, c = threadIdx.x & 15, n = blockIdx.x * 16 + c, cl = threadIdx.x >> 4;
    float acc = 0.f;
    if (n < N)
        for (int b = cl; b < nblocks; b += 64) acc += partial[((long long)b * 2 + which) * N + n];
    sh[cl][c] = acc;
    __syncthreads();
    if (cl == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) t += sh[i][c];
        (which ? dbeta : dgamma)[n] = t;
    }
}

constexpr int kBwdBlocks = 1024;  // workgroups (each leaves one [2][N] partial row)

template <typename T, typename GT>
int launch_bwd_vpl(const void* x, const void* res, const void* gamma, const void* dy, void* dz, float* partial,
                   int nblocks, long long rows, int N, float eps, hipStream_t stream, const bf_dropout_t* drop, void* dx,
                   const void* dy2) {
    const int nvec = N >> 3;
    const size_t lds = (size_t)kRowsPerBlock * 2 * N * sizeof(float);
    const dim3 grid((unsigned)nblocks), block(64 * kRowsPerBlock);
    const bf_dropout_t d = drop ? *drop : bf_dropout_t{0, 0, 0, 0, 0, 1.0f, 0, 0, nullptr};
#define BF_LNB_LAUNCH(VPL)                                                                                                  \
    do {                                                                                                                    \
        if (d.thresh)                                                                                                       \
            hipLaunchKernelGGL((add_layernorm_bwd_kernel<T, GT, VPL, true>), grid, block, lds, stream, (const T*)x,         \
                               (const T*)res, (const GT*)gamma, (const T*)dy, (T*)dz, partial, rows, N, eps, d, (T*)dx,    \
                               (const T*)dy2);                                                                              \
        else                                                                                                                \
            hipLaunchKernelGGL((add_layernorm_bwd_kernel<T, GT, VPL, false>), grid, block, lds, stream, (const T*)x,        \
                               (const T*)res, (const GT*)gamma, (const T*)dy, (T*)dz, partial, rows, N, eps, d, (T*)dx,    \
                               (const T*)dy2);                                                                              \
    } while (0)
    if (nvec <= 64) BF_LNB_LAUNCH(1);
    else if (nvec <= 128) BF_LNB_LAUNCH(2);
    else if (nvec <= 256) BF_LNB_LAUNCH(4);
    else BF_LNB_LAUNCH(8);
#undef BF_LNB_LAUNCH
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, typename GT>
int launch_vpl(const void* x, const void* res, const void* gamma, const void* beta, void* out, long long rows, int N,
               float eps, hipStream_t stream, const bf_dropout_t* drop) {
    const int nvec = N >> 3;
    const bf_dropout_t d = drop ? *drop : bf_dropout_t{0, 0, 0, 0, 0, 1.0f, 0, 0, nullptr};
    if (nvec % 32 == 0 && nvec <= 128) {  // N = 256, 512, 768, 1024: half a wave per row
        const dim3 hgrid((unsigned)((rows + 2 * kRowsPerBlock - 1) / (2 * kRowsPerBlock))), hblock(64 * kRowsPerBlock);
#define BF_LNH_LAUNCH(V)                                                                                                \
    do {                                                                                                                \
        if (d.thresh)                                                                                                   \
            hipLaunchKernelGGL((add_layernorm_half_kernel<T, GT, V, true>), hgrid, hblock, 0, stream, (const T*)x,      \
                               (const T*)res, (const GT*)gamma, (const GT*)beta, (T*)out, rows, N, eps, d);             \
        else                                                                                                            \
            hipLaunchKernelGGL((add_layernorm_half_kernel<T, GT, V, false>), hgrid, hblock, 0, stream, (const T*)x,     \
                               (const T*)res, (const GT*)gamma, (const GT*)beta, (T*)out, rows, N, eps, d);             \
    } while (0)
        switch (nvec / 32) {
            case 1: BF_LNH_LAUNCH(1); break;
            case 2: BF_LNH_LAUNCH(2); break;
            case 3: BF_LNH_LAUNCH(3); break;
            default: BF_LNH_LAUNCH(4); break;
        }
#undef BF_LNH_LAUNCH
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), block(64 * kRowsPerBlock);
#define BF_LN_LAUNCH(VPL)                                                                                              \
    do {                                                                                                               \
        if (d.thresh)                                                                                                  \
            hipLaunchKernelGGL((add_layernorm_kernel<T, GT, VPL, true>), grid, block, 0, stream, (const T*)x,          \
                               (const T*)res, (const GT*)gamma, (const GT*)beta, (T*)out, rows, N, eps, d);            \
        else                                                                                                           \
            hipLaunchKernelGGL((add_layernorm_kernel<T, GT, VPL, false>), grid, block, 0, stream, (const T*)x,         \
                               (const T*)res, (const GT*)gamma, (const GT*)beta, (T*)out, rows, N, eps, d);            \
    } while (0)
    if (nvec <= 64) BF_LN_LAUNCH(1);
    else if (nvec <= 128) BF_LN_LAUNCH(2);
    else if (nvec <= 256) BF_LN_LAUNCH(4);
    else if (nvec <= 512) BF_LN_LAUNCH(8);
    else BF_LN_LAUNCH(16);
#undef BF_LN_LAUNCH
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int launch_gt(const void* x, const void* res, const void* gamma, const void* beta, int param_dtype, int dtype,
              void* out, long long rows, int N, float eps, hipStream_t stream, const bf_dropout_t* drop) {
    if (param_dtype == BF_DT_F32) return launch_vpl<T, float>(x, res, gamma, beta, out, rows, N, eps, stream, drop);
    if (param_dtype == dtype) return launch_vpl<T, T>(x, res, gamma, beta, out, rows, N, eps, stream, drop);
    BF_FAIL("bf_add_layernorm: gamma/beta must be fp32 or have the activation dtype");
}


// Embedding block in one pass (HF BertEmbeddings: LayerNorm(word[ids] + type[type_ids] + pos[pos_ids]), the block
// that feeds the first Bayesian layers of the converted model): one wave per token gathers its three table rows,
// adds them in fp32, normalises.  Replaces three gather launches, two full-size adds and the LayerNorm's own read
// of their result.  pos_ids / type_ids may be NULL: position = token index within its sequence of `seq_len` tokens,
// type 0 (the module's defaults).
template <typename T, typename GT, int VPL>
__global__ __launch_bounds__(64 * kRowsPerBlock) void embed_layernorm_kernel(
    const long long* __restrict__ ids, const long long* __restrict__ type_ids, const long long* __restrict__ pos_ids,
    const T* __restrict__ word, const T* __restrict__ type, const T* __restrict__ pos, const GT* __restrict__ gamma,
    const GT* __restrict__ beta, T* __restrict__ out, long long rows, int N, int seq_len, long long pos_rows,
    long long word_rows, long long type_rows, long long pos_table_rows, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = N >> 3;
    long long wi = ids[row], ti = type_ids ? type_ids[row] : 0, pi = pos_ids ? pos_ids[row % pos_rows] : row % seq_len;
    // an id outside its table: read row 0 instead and poison the output row (NaN) — never an out-of-bounds access
    const bool bad = (unsigned long long)wi >= (unsigned long long)word_rows ||
                     (unsigned long long)ti >= (unsigned long long)type_rows ||
                     (unsigned long long)pi >= (unsigned long long)pos_table_rows;
    if (bad) wi = ti = pi = 0;
    const T* wr = word + wi * N;
    const T* tr = type + ti * N;
    const T* pr = pos + pi * N;
    float v[VPL][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int vi = lane + 64 * c;
        if (vi < nvec) {
            float a[8], b[8];
            load8(wr + vi * 8, v[c]);
            load8(tr + vi * 8, a);
            load8(pr + vi * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[c][i] = (v[c][i] + a[i]) + b[i];
                sum += v[c][i];
            }
        }
    }
    const float inv_n = 1.0f / (float)N;
    const float mean = wave_sum(sum) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        if (lane + 64 * c < nvec) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = v[c][i] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    }
    const float rstd = bad ? __builtin_nanf("") : 1.0f / sqrtf(wave_sum(sq) * inv_n + eps);
    T* orow = out + row * N;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int vi = lane + 64 * c;
        if (vi < nvec) {
            float g[8], b[8], o[8];
            load8(gamma + vi * 8, g);
            load8(beta + vi * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf((v[c][i] - mean) * rstd, g[i], b[i]);
            store8(orow + vi * 8, o);
        }
    }
}

template <typename T, typename GT>
int launch_embed(const long long* ids, const long long* type_ids, const long long* pos_ids, const void* word,
                 const void* type, const void* pos, const void* gamma, const void* beta, void* out, long long rows, int N,
                 int seq_len, long long pos_rows, long long word_rows, long long type_rows, long long pos_table_rows,
                 float eps, hipStream_t stream) {
    const dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), block(64 * kRowsPerBlock);
    const int nvec = N / 8;
#define BF_EMB(V)                                                                                                      \
    hipLaunchKernelGGL((embed_layernorm_kernel<T, GT, V>), grid, block, 0, stream, ids, type_ids, pos_ids,              \
                       (const T*)word, (const T*)type, (const T*)pos, (const GT*)gamma, (const GT*)beta, (T*)out, rows, \
                       N, seq_len, pos_rows, word_rows, type_rows, pos_table_rows, eps)
    if (nvec <= 64) BF_EMB(1);
    else if (nvec <= 128) BF_EMB(2);
    else if (nvec <= 256) BF_EMB(4);
    else BF_EMB(8);
#undef BF_EMB
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int bf_launch_add_layernorm(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta,
                            int param_dtype, void* d_out, int dtype, long long rows, int N, float eps,
                            hipStream_t stream, const bf_dropout_t* drop) {
    if (rows < 0 || N <= 0) BF_FAIL("bf_add_layernorm: bad shape rows=%lld N=%d", rows, N);
    if (rows == 0) return 0;
    if (!d_x || !d_gamma || !d_beta || !d_out) BF_FAIL("bf_add_layernorm: null pointer");
    if (N % 8 || N > 8192) BF_FAIL("bf_add_layernorm: N=%d must be a multiple of 8 and at most 8192", N);
    if (rows > 0x7fffffffLL * kRowsPerBlock) BF_FAIL("bf_add_layernorm: too many rows");
    const uintptr_t al = (uintptr_t)d_x | (uintptr_t)d_residual | (uintptr_t)d_gamma | (uintptr_t)d_beta | (uintptr_t)d_out;
    if (al & 15) BF_FAIL("bf_add_layernorm: pointers must be 16-byte aligned");
    switch (dtype) {
        case BF_DT_BF16: return launch_gt<__bf16>(d_x, d_residual, d_gamma, d_beta, param_dtype, dtype, d_out, rows, N, eps, stream, drop);
        case BF_DT_F16: return launch_gt<_Float16>(d_x, d_residual, d_gamma, d_beta, param_dtype, dtype, d_out, rows, N, eps, stream, drop);
        case BF_DT_F32: return launch_gt<float>(d_x, d_residual, d_gamma, d_beta, param_dtype, dtype, d_out, rows, N, eps, stream, drop);
    }
    BF_FAIL("bf_add_layernorm: unknown dtype %d", dtype);
}


int bf_launch_embed_layernorm(const long long* d_ids, const long long* d_type_ids, const long long* d_pos_ids,
                              const void* d_word, const void* d_type, const void* d_pos, const void* d_gamma,
                              const void* d_beta, int param_dtype, void* d_out, int dtype, long long rows, int N,
                              int seq_len, long long pos_rows, long long word_rows, long long type_rows,
                              long long pos_table_rows, float eps, hipStream_t stream) {
    if (word_rows < 1 || type_rows < 1 || pos_table_rows < 1)
        BF_FAIL("bf_embed_layernorm: empty table (%lld / %lld / %lld rows)", word_rows, type_rows, pos_table_rows);
    if (!d_pos_ids && seq_len > pos_table_rows)
        BF_FAIL("bf_embed_layernorm: seq_len=%d exceeds the position table (%lld rows)", seq_len, pos_table_rows);
    if (rows < 0 || N <= 0 || seq_len < 1) BF_FAIL("bf_embed_layernorm: bad shape rows=%lld N=%d seq_len=%d", rows, N, seq_len);
    if (rows == 0) return 0;
    if (!d_ids || !d_word || !d_type || !d_pos || !d_gamma || !d_beta || !d_out) BF_FAIL("bf_embed_layernorm: null pointer");
    if (d_pos_ids && pos_rows < 1) BF_FAIL("bf_embed_layernorm: pos_rows must be >= 1 with explicit position ids");
    if (N % 8 || N > 4096) BF_FAIL("bf_embed_layernorm: N=%d must be a multiple of 8 and at most 4096", N);
    if (rows > 0x7fffffffLL * kRowsPerBlock) BF_FAIL("bf_embed_layernorm: too many rows");
    const uintptr_t al = (uintptr_t)d_word | (uintptr_t)d_type | (uintptr_t)d_pos | (uintptr_t)d_gamma | (uintptr_t)d_beta | (uintptr_t)d_out;
    if (al & 15) BF_FAIL("bf_embed_layernorm: pointers must be 16-byte aligned");
    if (param_dtype != dtype && param_dtype != BF_DT_F32) BF_FAIL("bf_embed_layernorm: gamma/beta must be fp32 or the table dtype");
    const bool pf = param_dtype == BF_DT_F32;
#define BF_EMB_T(T)                                                                                                     \
    return pf ? launch_embed<T, float>(d_ids, d_type_ids, d_pos_ids, d_word, d_type, d_pos, d_gamma, d_beta, d_out, rows, N, \
                                       seq_len, pos_rows, word_rows, type_rows, pos_table_rows, eps, stream)               \
              : launch_embed<T, T>(d_ids, d_type_ids, d_pos_ids, d_word, d_type, d_pos, d_gamma, d_beta, d_out, rows, N,    \
                                   seq_len, pos_rows, word_rows, type_rows, pos_table_rows, eps, stream)
    switch (dtype) {
        case BF_DT_BF16: BF_EMB_T(__bf16);
        case BF_DT_F16: BF_EMB_T(_Float16);
        case BF_DT_F32: BF_EMB_T(float);
    }
#undef BF_EMB_T
    BF_FAIL("bf_embed_layernorm: unknown dtype %d", dtype);
}

static int bwd_blocks(long long rows) {
    const long long need = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    return (int)(need < kBwdBlocks ? (need < 1 ? 1 : need) : kBwdBlocks);
}

size_t bf_add_layernorm_bwd_ws_bytes(long long rows, int N) {
    if (rows < 1 || N < 1) return 0;
    return (size_t)bwd_blocks(rows) * 2 * N * sizeof(float);
}

int bf_launch_add_layernorm_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype,
                                const void* d_dy, void* d_dz, float* d_dgamma, float* d_dbeta, void* d_workspace,
                                size_t workspace_bytes, int dtype, long long rows, int N, float eps, hipStream_t stream,
                                const bf_dropout_t* drop, void* d_dx, const void* d_dy2) {
    if ((uintptr_t)d_dy2 & 15) BF_FAIL("bf_add_layernorm_bwd: the second gradient must be 16-byte aligned");
    if (drop && drop->thresh && (!d_dx || ((uintptr_t)d_dx & 15)))
        BF_FAIL("bf_add_layernorm_bwd: dropout needs a 16-byte aligned d_dx (the gradient of the dropped input)");
    if (rows < 0 || N <= 0) BF_FAIL("bf_add_layernorm_bwd: bad shape rows=%lld N=%d", rows, N);
    if (N % 8 || N > 4096) BF_FAIL("bf_add_layernorm_bwd: N=%d must be a multiple of 8 and at most 4096", N);
    if (!d_dgamma || !d_dbeta) BF_FAIL("bf_add_layernorm_bwd: null parameter gradient");
    if (rows == 0) {
        BF_HIP_CHECK(hipMemsetAsync(d_dgamma, 0, (size_t)N * sizeof(float), stream));
        BF_HIP_CHECK(hipMemsetAsync(d_dbeta, 0, (size_t)N * sizeof(float), stream));
        return 0;
    }
    if (!d_x || !d_gamma || !d_dy || !d_dz) BF_FAIL("bf_add_layernorm_bwd: null pointer");
    const uintptr_t al = (uintptr_t)d_x | (uintptr_t)d_residual | (uintptr_t)d_gamma | (uintptr_t)d_dy | (uintptr_t)d_dz;
    if (al & 15) BF_FAIL("bf_add_layernorm_bwd: pointers must be 16-byte aligned");
    const size_t need = bf_add_layernorm_bwd_ws_bytes(rows, N);
    if (!d_workspace || workspace_bytes < need) BF_FAIL("bf_add_layernorm_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    const int nb = bwd_blocks(rows);
    float* partial = reinterpret_cast<float*>(d_workspace);
    int rc = 1;
#define BF_LNB_DISPATCH(T)                                                                                              \
    rc = param_dtype == BF_DT_F32 ? launch_bwd_vpl<T, float>(d_x, d_residual, d_gamma, d_dy, d_dz, partial, nb, rows, N, eps, stream, drop, d_dx, d_dy2) \
         : param_dtype == dtype   ? launch_bwd_vpl<T, T>(d_x, d_residual, d_gamma, d_dy, d_dz, partial, nb, rows, N, eps, stream, drop, d_dx, d_dy2)     \
                                  : (bf_set_error("bf_add_layernorm_bwd: gamma must be fp32 or have the activation dtype"), 1)
    switch (dtype) {
        case BF_DT_BF16: BF_LNB_DISPATCH(__bf16); break;
        case BF_DT_F16: BF_LNB_DISPATCH(_Float16); break;
        case BF_DT_F32: BF_LNB_DISPATCH(float); break;
        default: BF_FAIL("bf_add_layernorm_bwd: unknown dtype %d", dtype);
    }
#undef BF_LNB_DISPATCH
    if (rc) return rc;
    hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3((N + 15) / 16, 2), dim3(1024), 0, stream, partial, nb, N, d_dgamma,
                       d_dbeta);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
