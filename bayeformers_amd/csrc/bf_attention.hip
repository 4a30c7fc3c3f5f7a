// bf_attention.hip — softmax(Q K^T * scale + mask) V for the attention block that sits between the Bayesian
// query/key/value projections and the Bayesian output projection of the transformers the reference converts
// (HF BertSelfAttention around bnn.Linear.forward, /root/reference/bayeformers/nn/layers/linear.py:83-104; the
// reference itself runs whatever the wrapped model runs).  Inference-time forward only.
//
// Shape of the kernel (head size 64, keys/queries in tiles of 128, bf16 or fp16):
//   * one 256-thread workgroup per (128 queries, head, sequence); each of its 4 waves owns 32 queries;
//   * a 128-key tile of K ([key][d], 16-byte chunks XOR-swizzled by key & 7) and of V ([key][d], 160-byte rows) is
//     staged in LDS with 16-byte accesses;
//   * S^T = K Q^T on v_mfma_f32_16x16x32 with K as the row operand: a lane ends up with 4 consecutive keys of ONE query
//     per 16-key block, so the softmax statistics of a query are 32 in-lane values + two cross-lane steps, and the
//     probabilities of two neighbouring key blocks are already the 8-element column operand of the P V product —
//     the k index of that product is a fixed permutation of the keys, applied identically to the V^T fragments, which
//     come straight out of the row-major V tile through gfx950's LDS transpose read (ds_read_b64_tr_b16);
//   * O^T = V^T P^T accumulates in fp32; longer sequences walk the key tiles with the usual running max / sum rescale;
//   * a lane owns 4 consecutive features of one query at the end: 8-byte stores into the [B, T, H, 64] output.
// Algorithmic HBM bytes: (3 reads + 1 write) * B*T*H*64 * 2 B.
#include <stdlib.h>

#include "bf_common.h"
#include "bf_philox.h"

namespace {

// The attention output rows (the x of the attention-out GEMM that runs next) leave through plain stores: nontemporal, sc1 and
// sc0 sc1 (write-through) stores were measured in the BERT-base step, one box, interleaved runs
// (profiles/r5e_gemm_variants_in_step_ab.txt): no policy beats plain stores.
template <typename V>
__device__ __forceinline__ void attn_st8(V* p, V v) { *p = v; }


constexpr int HD = 64;          // head size
constexpr int TQ = 128;         // queries per workgroup
constexpr int TKEY = 128;       // keys per tile
constexpr int K_ROW = HD * 2;   // 128 B
constexpr int V_ROW = HD * 2 + 32;     // 160 B: the 8 key rows a 32-lane half of a transpose-read touches tile one bank row
constexpr int K_BYTES = TKEY * K_ROW;  // 16 KiB
constexpr int V_BYTES = TKEY * V_ROW;  // 20 KiB

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

template <typename T>
struct Mfma;
template <>
struct Mfma<__bf16> {
    using frag = bf16x8_t;
    using half4 = bf16x4_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mfma<_Float16> {
    using frag = f16x8_t;
    using half4 = f16x4_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

struct AttnParams {
    const void* q;
    const void* k;
    const void* v;
    const float* mask;  // [B][T] additive, nullable
    const unsigned char* mask_off;  // nullable device flag: non-zero = the mask is all zeros, skip it
    void* out;
    float* lse;  // nullable: [B][H][T] log2-sum-exp2 of the scaled, masked scores (for the backward kernels)
    long long tok_stride;  // elements between consecutive tokens of q / k / v (H * 64 for packed heads)
    int B, T, H;
    float scale_log2e;  // scaling * log2(e)
    int ablate;         // developer ablation bits (BF_ATTN_ABLATE): 1 = no output stores, 2 = K/V staged once from tile 0
    // DROP instantiation (training, HF attention_probs_dropout): the probabilities are dropped AFTER the softmax
    // normalisation (the row sums keep every term) with the Philox keep-mask of bf_philox.h.  A dropout group is the 8
    // probabilities of one P^T fragment: query q, keys tile * 128 + (2c + e) * 16 + 4 lg + j (e = 0, 1; j = 0..3) ->
    //   group index g = (((b * H + h) * T + q) * (T / 32) + tile * 4 + c) * 4 + lg,   field = e * 4 + j.
    // keep_bits (nullable): the decisions as one 32-bit word per (b, h, q, tile, lg), bit c * 8 + e * 4 + j — 1 bit per
    // probability, 3 % of the layer's q/k/v/o traffic — for the backward kernel, whose lanes own 4 QUERIES of one key and
    // would have to evaluate eight Philox blocks per group of eight to regenerate them.
    bf_dropout_t drop;
    uint32_t* keep_bits;  // [B][H][T][T / 32]
};

// 3 workgroups per CU, for the DROP instantiation too: at 3 it spills 12 registers (168 VGPRs + 48 B of scratch), at 2 it does
// not (178 VGPRs); measured back to back, same box: 69.9 vs 72.0 us at the BERT-base shape, 278 vs 304 us at 160 x 16 heads x
// 384 tokens — the third workgroup is worth more than the spills cost.
template <typename T, bool DROP = false>
__global__ __launch_bounds__(256, 3) void attention_fwd_kernel(const AttnParams p) {
    using frag = typename Mfma<T>::frag;
    using half4 = typename Mfma<T>::half4;
    __shared__ __attribute__((aligned(16))) char smem[K_BYTES + V_BYTES + TKEY * 4];
    char* const ks = smem;
    char* const vs = smem + K_BYTES;
    float* const ms = reinterpret_cast<float*>(smem + K_BYTES + V_BYTES);  // this tile's key mask, in log2 units
    // the caller may not know on the host whether its padding mask hides anything: a device flag says so (uniform)
    const float* const mask = (p.mask && !(p.mask_off && *p.mask_off)) ? p.mask : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * TQ + wid * 32, h = blockIdx.y, b = blockIdx.z;
    const T* qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.T * p.tok_stride + (long long)h * HD;
    const T* kb = reinterpret_cast<const T*>(p.k) + (long long)b * p.T * p.tok_stride + (long long)h * HD;
    const T* vb = reinterpret_cast<const T*>(p.v) + (long long)b * p.T * p.tok_stride + (long long)h * HD;

    // Q^T column operand: lane (query = li, k group = lg) holds 8 consecutive features of its query
    frag qf[2][2];
#pragma unroll
    for (int qb_i = 0; qb_i < 2; ++qb_i)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
            qf[qb_i][dh] = *reinterpret_cast<const frag*>(qb + (long long)(q0 + qb_i * 16 + li) * p.tok_stride + dh * 32 + lg * 8);

    f32x4_t o[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float run_max[2] = {-INFINITY, -INFINITY}, run_sum[2] = {0.f, 0.f};

    for (int key0 = 0; key0 < p.T; key0 += TKEY) {
        if (key0) __syncthreads();  // the previous tile's fragment reads are done
#ifdef BF_DEV
        if (!((p.ablate & 2) && (blockIdx.x | blockIdx.y | blockIdx.z)))
#endif
        // stage K ([key][d], chunk ^= key & 7) and V ([key][d], 160-byte rows), 16 bytes per lane
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, c8 = c & 7;
            const f32x4_t kv = *reinterpret_cast<const f32x4_t*>(kb + (long long)(key0 + row) * p.tok_stride + c8 * 8);
            *reinterpret_cast<f32x4_t*>(ks + row * K_ROW + ((c8 ^ (row & 7)) << 4)) = kv;
            const f32x4_t vv = *reinterpret_cast<const f32x4_t*>(vb + (long long)(key0 + row) * p.tok_stride + c8 * 8);
            *reinterpret_cast<f32x4_t*>(vs + row * V_ROW + (c8 << 4)) = vv;
        }
        if (mask && tid < TKEY / 4)
            *reinterpret_cast<f32x4_t*>(ms + tid * 4) =
                *reinterpret_cast<const f32x4_t*>(mask + (long long)b * p.T + key0 + tid * 4) * 1.4426950408889634f;
        __syncthreads();

        // One block of 16 queries at a time (keeps the live scores at 32 registers, 4 waves per SIMD fit).
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            // S^T[key][query]: lane (query li, group lg) holds keys kb*16 + 4*lg + 0..3 of each 16-key block
            f32x4_t s[8];
#pragma unroll
            for (int kbk = 0; kbk < 8; ++kbk) {
                s[kbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    const int row = kbk * 16 + li;
                    const frag kf = *reinterpret_cast<const frag*>(ks + row * K_ROW + (((dh * 4 + lg) ^ (row & 7)) << 4));
                    s[kbk] = Mfma<T>::run(kf, qf[qi][dh], s[kbk]);
                }
            }
            // scale (+ additive key mask: a lane's 4 keys of a block are one 16-byte LDS read), in log2 units
            float mx = -INFINITY;
#pragma unroll
            for (int kbk = 0; kbk < 8; ++kbk) {
                f32x4_t mk = {0.f, 0.f, 0.f, 0.f};
                if (mask) mk = *reinterpret_cast<const f32x4_t*>(ms + kbk * 16 + lg * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[kbk][j] = fmaf(s[kbk][j], p.scale_log2e, mk[j]);
                    mx = fmaxf(mx, s[kbk][j]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float new_max = fmaxf(run_max[qi], mx);
            // a fully masked row so far keeps max = -inf: use 0 as the reference point (all terms become 0)
            const float ref = new_max == -INFINITY ? 0.f : new_max;
            const float corr = __builtin_amdgcn_exp2f(run_max[qi] - ref);  // exp2(-inf) = 0 on the first tile
            float sum = 0.f;
#pragma unroll
            for (int kbk = 0; kbk < 8; ++kbk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[kbk][j] = __builtin_amdgcn_exp2f(s[kbk][j] - ref);
                    sum += s[kbk][j];
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            run_sum[qi] = run_sum[qi] * corr + sum;
            run_max[qi] = new_max;
            if constexpr (DROP) {
                // drop probabilities (the sum above saw all of them); the 1 / (1 - p) factor joins the final normalisation
                const unsigned long long qrow = ((unsigned long long)b * p.H + h) * p.T + (q0 + qi * 16 + li);
                const unsigned long long g0 = (qrow * (unsigned)(p.T >> 5) + (unsigned)(key0 >> 7) * 4u) * 4u + lg +
                                              (((unsigned long long)p.drop.g0_hi << 32) | p.drop.g0_lo);
                uint32_t word = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned long long g = g0 + 4ull * c;
                    const uint32_t keep = bf_dropout_keep8((uint32_t)g, (uint32_t)(g >> 32), bf_dropout_call(p.drop), p.drop.site,
                                                           p.drop.k0, p.drop.k1, p.drop.thresh);
                    word |= keep << (8 * c);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (!((keep >> (e * 4 + j)) & 1u)) s[2 * c + e][j] = 0.f;
                }
                if (p.keep_bits) p.keep_bits[(qrow * (unsigned)(p.T >> 7) + (unsigned)(key0 >> 7)) * 4u + lg] = word;
            }
#pragma unroll
            for (int db = 0; db < 4; ++db) o[qi][db] *= corr;
            // O^T[d][query] += V^T[d][k] P^T[k][query], k walking 32 keys at a time in the order
            // (group lg, slot j):  j < 4 -> key (2c)*16 + 4*lg + j,  j >= 4 -> key (2c+1)*16 + 4*lg + (j - 4)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x8_t pv = {s[2 * c][0],     s[2 * c][1],     s[2 * c][2],     s[2 * c][3],
                                    s[2 * c + 1][0], s[2 * c + 1][1], s[2 * c + 1][2], s[2 * c + 1][3]};
                const frag pf = __builtin_convertvector(pv, frag);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    // V^T fragment by the LDS transpose read: the 16 lanes of a group point at the [4 keys][16 d]
                    // block (lane -> key li >> 2, features 4 * (li & 3) ..), each gets its column = 4 keys of feature li
                    const char* vblk = vs + (lg * 4 + (li >> 2)) * V_ROW + (db * 16 + (li & 3) * 4) * 2;
                    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vblk + (2 * c) * 16 * V_ROW));
                    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vblk + (2 * c + 1) * 16 * V_ROW));
                    const s16x8_t v01 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[qi][db] = Mfma<T>::run(__builtin_bit_cast(frag, v01), pf, o[qi][db]);
                }
            }
        }
    }

    // lane (query li, group lg) holds features db*16 + 4*lg + 0..3 of its query: 8-byte stores
    T* ob = reinterpret_cast<T*>(p.out) + ((long long)b * p.T * p.H + h) * HD;
#ifdef BF_DEV
    if (!(p.ablate & 1) || run_sum[0] == 12345.f)
#endif
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const float inv = run_sum[qi] > 0.f ? (DROP ? p.drop.inv_keep : 1.0f) / run_sum[qi] : 0.f;
        // a fully masked query has no probabilities: +inf makes exp2(score - lse) = 0 in the backward kernels
        if (p.lse && lg == 0)
            p.lse[((long long)b * p.H + h) * p.T + q0 + qi * 16 + li] =
                run_sum[qi] > 0.f ? run_max[qi] + __builtin_amdgcn_logf(run_sum[qi]) : INFINITY;
        T* orow = ob + (long long)(q0 + qi * 16 + li) * p.H * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const f32x4_t r = o[qi][db] * inv;
            attn_st8(reinterpret_cast<half4*>(orow + db * 16 + lg * 4), __builtin_convertvector(r, half4));
        }
    }
}

}  // namespace

int bf_launch_attention_fwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask,
                            const unsigned char* d_mask_off, void* d_out, float* d_lse, int dtype, int B, int T, int H,
                            int head_dim, long long token_stride, float scaling, hipStream_t stream,
                            const bf_dropout_t* drop, uint32_t* d_keep_bits) {
    if (!d_q || !d_k || !d_v || !d_out) BF_FAIL("bf_attention_fwd: NULL argument");
    if (dtype != BF_DT_BF16 && dtype != BF_DT_F16) BF_FAIL("bf_attention_fwd: dtype must be bf16 or fp16");
    if (head_dim != HD) BF_FAIL("bf_attention_fwd: head size %d (only %d)", head_dim, HD);
    if (B < 1 || H < 1 || T < TKEY || T % TKEY) BF_FAIL("bf_attention_fwd: T=%d must be a positive multiple of %d", T, TKEY);
    if (B > 65535 || H > 65535) BF_FAIL("bf_attention_fwd: B or H exceeds the grid");
    if (token_stride < (long long)H * HD || token_stride % 8) BF_FAIL("bf_attention_fwd: bad token stride %lld", token_stride);
    if (((uintptr_t)d_q | (uintptr_t)d_k | (uintptr_t)d_v | (uintptr_t)d_out) & 15) BF_FAIL("bf_attention_fwd: pointers must be 16-byte aligned");
    AttnParams p;
    p.q = d_q;
    p.k = d_k;
    p.v = d_v;
    p.mask = d_mask;
    p.mask_off = d_mask_off;
    p.out = d_out;
    p.lse = d_lse;
    p.tok_stride = token_stride;
    p.B = B;
    p.T = T;
    p.H = H;
    p.scale_log2e = scaling * 1.4426950408889634f;
#ifdef BF_DEV
    {
        const char* e = getenv("BF_ATTN_ABLATE");
        p.ablate = e ? atoi(e) : 0;
    }
#else
    p.ablate = 0;
#endif
    const dim3 grid(T / TQ, H, B);
    p.keep_bits = nullptr;
    p.drop = bf_dropout_t{0, 0, 0, 0, 0, 1.0f, 0, 0, nullptr};
    if (drop && drop->thresh) {
        if (d_keep_bits && ((uintptr_t)d_keep_bits & 3)) BF_FAIL("bf_attention_fwd: keep bits must be 4-byte aligned");
        p.drop = *drop;
        p.keep_bits = d_keep_bits;
        if (dtype == BF_DT_BF16) hipLaunchKernelGGL((attention_fwd_kernel<__bf16, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attention_fwd_kernel<_Float16, true>), grid, dim3(256), 0, stream, p);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (dtype == BF_DT_BF16) hipLaunchKernelGGL(attention_fwd_kernel<__bf16>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(attention_fwd_kernel<_Float16>, grid, dim3(256), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
