// bf_attention_bwd.hip — backward of bf_attention_fwd (csrc/bf_attention.hip): autograd through the attention
// block between the Bayesian query/key/value projections and the Bayesian output projection in the reference's
// training loop (/root/reference/examples/bert_glue.py:239, `loss.backward()` through HF BertSelfAttention).
//
// With P = softmax(scale Q K^T + mask) (recomputed from Q, K and the log-sum-exp rows the forward saved),
// delta_q = sum_d dO[q][d] O[q][d] and dS = P o (dO V^T - delta):
//     dV = P^T dO,   dK = scale dS^T Q,   dQ = scale dS K.
// Two kernels, no atomics (deterministic), head size 64, queries / keys in tiles of 128, bf16 or fp16:
//   * dq kernel — one workgroup per (128 queries, head, sequence), each of its 4 waves owns 32 queries and walks the
//     key tiles.  Same orientation as the forward: S^T = K Q^T and dP^T = V dO^T on v_mfma_f32_16x16x32 with the key
//     operand as rows, so a lane holds 4 consecutive keys of ONE query per 16-key block; dS^T of two neighbouring
//     blocks is the column operand of dQ^T = K^T dS^T, and the K^T fragments come out of a row-major K tile through the
//     LDS transpose read (ds_read_b64_tr_b16), exactly as V^T does in the forward.  Also leaves delta for the second kernel.
//   * dk/dv kernel — one workgroup per (128 keys, head, sequence), each of its 8 waves owns 16 keys and walks the query tiles
//     with the roles swapped: S = Q K^T and dP = dO V^T with the QUERY operand as rows, so a lane holds 4 consecutive
//     queries of one key; P and dS are the column operands of dV^T = dO^T P and dK^T = Q^T dS, with dO^T / Q^T through
//     the transpose read.
// Algorithmic HBM bytes: dq kernel 5 reads (Q, K, V, O, dO) + 1 write; dk/dv kernel 4 reads + 2 writes, x B*T*H*64*2 B.
#include "bf_common.h"
#include "bf_device.h"

namespace {

constexpr int HD = 64;
constexpr int TT = 128;                 // queries / keys per tile
constexpr int S_ROW = HD * 2;           // 128 B: swizzled image for direct 16-byte fragment reads (chunk ^= row & 7)
constexpr int P_ROW = HD * 2 + 32;      // 160 B: padded image for the transpose reads (see bf_attention.hip)
constexpr int S_BYTES = TT * S_ROW;     // 16 KiB
constexpr int P_BYTES = TT * P_ROW;     // 20 KiB

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

struct BwdParams {
    const void* q;
    const void* k;
    const void* v;
    const float* mask;              // [B][T] additive over the keys, nullable
    const unsigned char* mask_off;  // nullable device flag: non-zero = skip the mask
    const void* o;                  // [B][T][H][64]
    const void* dout;               // [B][T][H][64]
    const float* lse;               // [B][H][T], log2 units
    float* delta;                   // [B][H][T]
    void* dq;                       // [B][T][H][64]
    void* dk;
    void* dv;
    long long tok_stride;           // elements between consecutive tokens of q / k / v
    int B, T, H;
    float scale, scale_log2e;
    // attention_probs_dropout (the forward's DROP instantiation): the keep decisions the forward stored, one bit per
    // probability — word (b, h, q, tile, lg), bit c * 8 + e * 4 + j <-> key tile * 128 + (2c + e) * 16 + 4 lg + j — and the
    // 1 / (1 - p) factor.  NULL: no dropout.  With P~ = P o keep / (1 - p):  dV = P~^T dO,  dS = P o (keep / (1 - p) o dP~ - delta).
    const uint32_t* keep_bits;
    float inv_keep;
    // one-tile kernel only (round 5): per (sequence, head) the sums over the 128 tokens of dq, dk, dv AS STORED —
    // [B][H][3][64] fp32 — from which attention_colsum_finish_kernel makes the per-sample column sums that the Bayesian
    // query / key / value layers take as their bias gradient (bf_linear_bwd: d_dy_colsum).  NULL: not wanted.
    float* cs_partial;
};

// stage a [128][64] tile (rows `row0`.. of a [tokens][stride] tensor) into the swizzled and / or the padded image
template <typename T, int NT = 256>
__device__ __forceinline__ void stage_tile(const T* base, long long stride, int row0, char* swz, char* pad, int tid) {
#pragma unroll
    for (int i = 0; i < 1024 / NT; ++i) {
        const int c = tid + NT * i, row = c >> 3, c8 = c & 7;
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(base + (long long)(row0 + row) * stride + c8 * 8);
        if (swz) *reinterpret_cast<f32x4_t*>(swz + row * S_ROW + ((c8 ^ (row & 7)) << 4)) = x;
        if (pad) *reinterpret_cast<f32x4_t*>(pad + row * P_ROW + (c8 << 4)) = x;
    }
}

// row-operand fragment (16 rows x 32 features) of block `blk` of a swizzled tile: lane (row li, k group lg)
template <typename T>
__device__ __forceinline__ typename Mfma<T>::frag row_frag(const char* swz, int blk, int dh, int li, int lg) {
    const int row = blk * 16 + li;
    return *reinterpret_cast<const typename Mfma<T>::frag*>(swz + row * S_ROW + (((dh * 4 + lg) ^ (row & 7)) << 4));
}

// transposed fragment: rows = features db*16 + li, k = the 32 tile rows {(2c)*16 + 4 lg + 0..3, (2c+1)*16 + 4 lg + 0..3}
// of a padded tile, by two LDS transpose reads (the 16 lanes of a group point at a [4 rows][16 features] block)
template <typename T, int ROW = P_ROW>
__device__ __forceinline__ typename Mfma<T>::frag tr_frag(const char* pad, int c, int db, int li, int lg) {
    const char* blk = pad + (lg * 4 + (li >> 2)) * ROW + (db * 16 + (li & 3) * 4) * 2;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + (2 * c) * 16 * ROW));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + (2 * c + 1) * 16 * ROW));
    const s16x8_t ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(typename Mfma<T>::frag, ab);
}

template <typename T>
__device__ __forceinline__ typename Mfma<T>::frag pack2(const f32x4_t a, const f32x4_t b) {
    const f32x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_convertvector(v, typename Mfma<T>::frag);
}

// ---------------------------------------------------------------------------------------------------- dQ (+ delta)
template <typename T, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attention_bwd_dq_kernel(const BwdParams p) {
    using frag = typename Mfma<T>::frag;
    using half4 = typename Mfma<T>::half4;
    __shared__ __attribute__((aligned(16))) char smem[2 * S_BYTES + P_BYTES + TT * 4];
    char* const ks = smem;                     // K, swizzled: row operand of S^T
    char* const vs = smem + S_BYTES;           // V, swizzled: row operand of dP^T
    char* const kp = smem + 2 * S_BYTES;       // K, padded: K^T through the transpose read
    float* const ms = reinterpret_cast<float*>(smem + 2 * S_BYTES + P_BYTES);
    const float* const mask = (p.mask && !(p.mask_off && *p.mask_off)) ? p.mask : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * TT + wid * 32, h = blockIdx.y, b = blockIdx.z;
    const long long hoff = (long long)h * HD;
    const T* qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.T * p.tok_stride + hoff;
    const T* kb = reinterpret_cast<const T*>(p.k) + (long long)b * p.T * p.tok_stride + hoff;
    const T* vb = reinterpret_cast<const T*>(p.v) + (long long)b * p.T * p.tok_stride + hoff;
    const long long ostride = (long long)p.H * HD;
    const T* ob = reinterpret_cast<const T*>(p.o) + (long long)b * p.T * ostride + hoff;
    const T* dob = reinterpret_cast<const T*>(p.dout) + (long long)b * p.T * ostride + hoff;

    // column operands: lane (query li, k group lg) holds 8 consecutive features of its query, per 32-feature half
    frag qf[2][2], dof[2][2];
    float delta[2], lse[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const long long q = q0 + qi * 16 + li;
        float part = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            qf[qi][dh] = *reinterpret_cast<const frag*>(qb + q * p.tok_stride + dh * 32 + lg * 8);
            dof[qi][dh] = *reinterpret_cast<const frag*>(dob + q * ostride + dh * 32 + lg * 8);
            const frag of = *reinterpret_cast<const frag*>(ob + q * ostride + dh * 32 + lg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf((float)dof[qi][dh][e], (float)of[e], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        delta[qi] = part;
        lse[qi] = p.lse[((long long)b * p.H + h) * p.T + q];
        if (lg == 0) p.delta[((long long)b * p.H + h) * p.T + q] = part;
    }

    f32x4_t dq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dq[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int key0 = 0; key0 < p.T; key0 += TT) {
        if (key0) __syncthreads();
        stage_tile<T>(kb, p.tok_stride, key0, ks, kp, tid);
        stage_tile<T>(vb, p.tok_stride, key0, vs, nullptr, tid);
        if (mask && tid < TT / 4)
            *reinterpret_cast<f32x4_t*>(ms + tid * 4) =
                *reinterpret_cast<const f32x4_t*>(mask + (long long)b * p.T + key0 + tid * 4) * 1.4426950408889634f;
        __syncthreads();
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            // DROP: the lane's 32 probabilities of this (query, key tile) are the 32 bits of ONE keep word, in the
            // forward's own layout (word (query, key tile, lg), bit 4 kbk + j) — loaded ahead of the products
            uint32_t kw = 0;
            if constexpr (DROP)
                kw = p.keep_bits[((((long long)b * p.H + h) * p.T + (q0 + qi * 16 + li)) * (p.T / TT) + key0 / TT) * 4 + lg];
            // S^T and dP^T [key][query]: lane (query li, group lg) holds keys kbk*16 + 4 lg + 0..3 of each block
            f32x4_t s[8], dp[8];
#pragma unroll
            for (int kbk = 0; kbk < 8; ++kbk) {
                s[kbk] = dp[kbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    s[kbk] = Mfma<T>::run(row_frag<T>(ks, kbk, dh, li, lg), qf[qi][dh], s[kbk]);
                    dp[kbk] = Mfma<T>::run(row_frag<T>(vs, kbk, dh, li, lg), dof[qi][dh], dp[kbk]);
                }
            }
#pragma unroll
            for (int kbk = 0; kbk < 8; ++kbk) {
                f32x4_t mk = {0.f, 0.f, 0.f, 0.f};
                if (mask) mk = *reinterpret_cast<const f32x4_t*>(ms + kbk * 16 + lg * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[kbk][j], p.scale_log2e, mk[j]) - lse[qi]);
                    if constexpr (DROP) {
                        const float m = ((kw >> (kbk * 4 + j)) & 1u) ? p.inv_keep : 0.f;
                        s[kbk][j] = pr * (dp[kbk][j] * m - delta[qi]);  // dS^T = P o (keep / (1 - p) o dP~ - delta)
                    } else {
                        s[kbk][j] = pr * (dp[kbk][j] - delta[qi]);  // dS^T
                    }
                }
            }
            // dQ^T[d][query] += K^T[d][k] dS^T[k][query], k walking 32 keys at a time in the order the blocks give
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const frag dsf = pack2<T>(s[2 * c], s[2 * c + 1]);
#pragma unroll
                for (int db = 0; db < 4; ++db) dq[qi][db] = Mfma<T>::run(tr_frag<T>(kp, c, db, li, lg), dsf, dq[qi][db]);
            }
        }
    }
    // lane (query li, group lg) holds features db*16 + 4 lg + 0..3 of its query
    T* dqb = reinterpret_cast<T*>(p.dq) + (long long)b * p.T * ostride + hoff;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        T* row = dqb + (long long)(q0 + qi * 16 + li) * ostride;
#pragma unroll
        for (int db = 0; db < 4; ++db)
            *reinterpret_cast<half4*>(row + db * 16 + lg * 4) = __builtin_convertvector(dq[qi][db] * p.scale, half4);
    }
}

// ---------------------------------------------------------------------------------------------------- dK, dV
// 8 waves, each owning 16 keys of the workgroup's 128 (32 keys per wave need 117 spilled registers).
template <typename T, bool DROP = false>
__global__ __launch_bounds__(512, 2) void attention_bwd_dkv_kernel(const BwdParams p) {
    using frag = typename Mfma<T>::frag;
    using half4 = typename Mfma<T>::half4;
    __shared__ __attribute__((aligned(16))) char smem[2 * S_BYTES + 2 * P_BYTES + 2 * TT * 4 + (DROP ? TT * 16 : 0)];
    char* const qs = smem;                               // Q, swizzled: row operand of S
    char* const dos = smem + S_BYTES;                    // dO, swizzled: row operand of dP
    char* const qp = smem + 2 * S_BYTES;                 // Q, padded: Q^T through the transpose read
    char* const dop = smem + 2 * S_BYTES + P_BYTES;      // dO, padded: dO^T through the transpose read
    float* const lse_s = reinterpret_cast<float*>(smem + 2 * S_BYTES + 2 * P_BYTES);
    float* const del_s = lse_s + TT;
    uint32_t* const keep_s = reinterpret_cast<uint32_t*>(del_s + TT);  // DROP: [128 queries][4 lg] keep words of this (query tile, key tile)
    const float* const mask = (p.mask && !(p.mask_off && *p.mask_off)) ? p.mask : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const long long key = blockIdx.x * TT + wid * 16 + li;
    // DROP: this lane's key sits in word (query, key tile, lg_f) at one fixed bit (the forward's layout)
    const int key_i = wid * 16 + li;
    const int keep_word = (key_i >> 2) & 3, keep_bit = ((key_i >> 5) << 3) | (((key_i >> 4) & 1) << 2) | (key_i & 3);
    const long long hoff = (long long)h * HD;
    const T* qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.T * p.tok_stride + hoff;
    const T* kb = reinterpret_cast<const T*>(p.k) + (long long)b * p.T * p.tok_stride + hoff;
    const T* vb = reinterpret_cast<const T*>(p.v) + (long long)b * p.T * p.tok_stride + hoff;
    const long long ostride = (long long)p.H * HD;
    const T* dob = reinterpret_cast<const T*>(p.dout) + (long long)b * p.T * ostride + hoff;
    const float* lse_g = p.lse + ((long long)b * p.H + h) * p.T;
    const float* del_g = p.delta + ((long long)b * p.H + h) * p.T;

    // column operands: lane (key li, k group lg) holds 8 consecutive features of its key
    frag kf[2], vf[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
        kf[dh] = *reinterpret_cast<const frag*>(kb + key * p.tok_stride + dh * 32 + lg * 8);
        vf[dh] = *reinterpret_cast<const frag*>(vb + key * p.tok_stride + dh * 32 + lg * 8);
    }
    const float mk = mask ? mask[(long long)b * p.T + key] * 1.4426950408889634f : 0.f;
    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dk[j] = dv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int q0 = 0; q0 < p.T; q0 += TT) {
        if (q0) __syncthreads();
        stage_tile<T, 512>(qb, p.tok_stride, q0, qs, qp, tid);
        stage_tile<T, 512>(dob, ostride, q0, dos, dop, tid);
        if (tid < TT / 4) {
            *reinterpret_cast<f32x4_t*>(lse_s + tid * 4) = *reinterpret_cast<const f32x4_t*>(lse_g + q0 + tid * 4);
            *reinterpret_cast<f32x4_t*>(del_s + tid * 4) = *reinterpret_cast<const f32x4_t*>(del_g + q0 + tid * 4);
        }
        if constexpr (DROP)  // 512 words = 128 queries x 4 groups of this tile pair
            keep_s[tid] = p.keep_bits[((((long long)b * p.H + h) * p.T + q0 + (tid >> 2)) * (p.T / TT) + blockIdx.x) * 4 + (tid & 3)];
        __syncthreads();
        // S and dP [query][key]: lane (key li, group lg) holds queries qbk*16 + 4 lg + 0..3 of each block
        f32x4_t s[8], dp[8];
#pragma unroll
        for (int qbk = 0; qbk < 8; ++qbk) {
            s[qbk] = dp[qbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                s[qbk] = Mfma<T>::run(row_frag<T>(qs, qbk, dh, li, lg), kf[dh], s[qbk]);
                dp[qbk] = Mfma<T>::run(row_frag<T>(dos, qbk, dh, li, lg), vf[dh], dp[qbk]);
            }
        }
#pragma unroll
        for (int qbk = 0; qbk < 8; ++qbk) {
            const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + qbk * 16 + lg * 4);
            const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(del_s + qbk * 16 + lg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[qbk][j], p.scale_log2e, mk) - l4[j]);
                if constexpr (DROP) {
                    const uint32_t wrd = keep_s[(qbk * 16 + lg * 4 + j) * 4 + keep_word];
                    const float m = ((wrd >> keep_bit) & 1u) ? p.inv_keep : 0.f;
                    s[qbk][j] = pr * m;                          // P~ = P o keep / (1 - p)
                    dp[qbk][j] = pr * (dp[qbk][j] * m - d4[j]);  // dS
                } else {
                    s[qbk][j] = pr;                          // P
                    dp[qbk][j] = pr * (dp[qbk][j] - d4[j]);  // dS
                }
            }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const frag pf = pack2<T>(s[2 * c], s[2 * c + 1]);
            const frag dsf = pack2<T>(dp[2 * c], dp[2 * c + 1]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[db] = Mfma<T>::run(tr_frag<T>(dop, c, db, li, lg), pf, dv[db]);
                dk[db] = Mfma<T>::run(tr_frag<T>(qp, c, db, li, lg), dsf, dk[db]);
            }
        }
    }
    T* dkb = reinterpret_cast<T*>(p.dk) + (long long)b * p.T * ostride + hoff;
    T* dvb = reinterpret_cast<T*>(p.dv) + (long long)b * p.T * ostride + hoff;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        *reinterpret_cast<half4*>(dkb + key * ostride + db * 16 + lg * 4) = __builtin_convertvector(dk[db] * p.scale, half4);
        *reinterpret_cast<half4*>(dvb + key * ostride + db * 16 + lg * 4) = __builtin_convertvector(dv[db], half4);
    }
}


// ---------------------------------------------------------------------------------------------------- T = 128: all three
// One key tile = one query tile: dQ, dK and dV of a (sequence, head) in ONE workgroup, P and dS computed once.
// The first part is the dK / dV kernel (8 waves x 16 keys; S = Q K^T, dP = dO V^T with the queries as rows), with delta
// computed here (wave w: queries 16 w .. 16 w + 15).  Every wave leaves its 16 K rows (still in registers as the column
// operand) as rows of a padded K image once S and dP are done, and the dS of its 16 keys — a lane holds 4 consecutive
// queries of one key: 8-byte writes — as rows of a padded dS^T[key][query] image, over the Q / dO images that are dead by then;
// wave w takes queries 16 w .. 16 w + 15 of dQ^T = K^T dS^T with BOTH operands through the LDS transpose read (the
// same contraction order on either side: keys (2c) 16 + 4 lg + 0..3, (2c + 1) 16 + 4 lg + 0..3).  Deterministic: no
// atomics, a fixed summation order.  HBM: 5 reads (Q, K, V, O, dO) + 3 writes, against 9 + 3 of the two-kernel path.
constexpr int DS_ROW = TT * 2 + 32;  // 288 B: a row of dS^T (128 queries) + the transpose read's padding

template <typename T, bool DROP = false>
__global__ __launch_bounds__(512, 4) void attention_bwd_tile_kernel(const BwdParams p) {
    using frag = typename Mfma<T>::frag;
    using half4 = typename Mfma<T>::half4;
    __shared__ __attribute__((aligned(16))) char smem[2 * S_BYTES + 2 * P_BYTES + 2 * TT * 4 + (DROP ? TT * 16 : 0)];
    char* const qs = smem;                               // Q, swizzled: row operand of S
    char* const dos = smem + S_BYTES;                    // dO, swizzled: row operand of dP
    char* const qp = smem + 2 * S_BYTES;                 // Q, padded: Q^T through the transpose read
    char* const dop = smem + 2 * S_BYTES + P_BYTES;      // dO, padded: dO^T through the transpose read
    float* const lse_s = reinterpret_cast<float*>(smem + 2 * S_BYTES + 2 * P_BYTES);
    float* const del_s = lse_s + TT;
    uint32_t* const keep_s = reinterpret_cast<uint32_t*>(del_s + TT);  // DROP: [128 queries][4 lg] keep words of this (b, h)
    char* const kp = smem;                               // K, padded, over qs | dos once S and dP are done (K is still in registers)
    char* const dst = smem + 2 * S_BYTES;                // dS^T [128 keys][DS_ROW] over qp | dop once dK and dV are done
    static_assert(P_BYTES <= 2 * S_BYTES && TT * DS_ROW <= 2 * P_BYTES, "the K and dS^T images alias dead tiles only");
    const float* const mask = (p.mask && !(p.mask_off && *p.mask_off)) ? p.mask : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const long long row = wid * 16 + li;                 // this lane's key in the first part, its query in the second
    const long long hoff = (long long)h * HD;
    const T* qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.T * p.tok_stride + hoff;
    const T* kb = reinterpret_cast<const T*>(p.k) + (long long)b * p.T * p.tok_stride + hoff;
    const T* vb = reinterpret_cast<const T*>(p.v) + (long long)b * p.T * p.tok_stride + hoff;
    const long long ostride = (long long)p.H * HD;
    const T* ob = reinterpret_cast<const T*>(p.o) + (long long)b * p.T * ostride + hoff;
    const T* dob = reinterpret_cast<const T*>(p.dout) + (long long)b * p.T * ostride + hoff;
    const float* lse_g = p.lse + ((long long)b * p.H + h) * p.T;

    frag kf[2], vf[2], of[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
        kf[dh] = *reinterpret_cast<const frag*>(kb + row * p.tok_stride + dh * 32 + lg * 8);
        vf[dh] = *reinterpret_cast<const frag*>(vb + row * p.tok_stride + dh * 32 + lg * 8);
        of[dh] = *reinterpret_cast<const frag*>(ob + row * ostride + dh * 32 + lg * 8);
    }
    const float mk = mask ? mask[(long long)b * p.T + row] * 1.4426950408889634f : 0.f;

    stage_tile<T, 512>(qb, p.tok_stride, 0, qs, qp, tid);
    stage_tile<T, 512>(dob, ostride, 0, dos, dop, tid);
    if (tid < TT / 4)
        *reinterpret_cast<f32x4_t*>(lse_s + tid * 4) = *reinterpret_cast<const f32x4_t*>(lse_g + tid * 4);
    if constexpr (DROP) keep_s[tid] = p.keep_bits[(((long long)b * p.H + h) * p.T) * 4 + tid];  // 512 words = 128 x 4
    __syncthreads();
    {   // delta of query `row` = <dO, O>: the dO row comes out of the staged tile (a row fragment of block wid is the
        // lane's 8 features of that row); it is first read behind the barrier that follows the S / dP products
        float part = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            const frag df = row_frag<T>(dos, wid, dh, li, lg);
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf((float)df[e], (float)of[dh][e], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (lg == 0) {
            del_s[row] = part;
            p.delta[((long long)b * p.H + h) * p.T + row] = part;
        }
    }
    // S and dP [query][key]: lane (key li, group lg) holds queries qbk*16 + 4 lg + 0..3 of each block
    f32x4_t s[8], dp[8];
#pragma unroll
    for (int qbk = 0; qbk < 8; ++qbk) {
        s[qbk] = dp[qbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            s[qbk] = Mfma<T>::run(row_frag<T>(qs, qbk, dh, li, lg), kf[dh], s[qbk]);
            dp[qbk] = Mfma<T>::run(row_frag<T>(dos, qbk, dh, li, lg), vf[dh], dp[qbk]);
        }
        if (qbk & 1) __builtin_amdgcn_sched_barrier(0);  // (keeps the fragment reads of later blocks from piling up in registers)
    }
    __syncthreads();  // delta of all 128 queries is in LDS; the swizzled Q and dO images are dead
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) *reinterpret_cast<frag*>(kp + row * P_ROW + (dh * 32 + lg * 8) * 2) = kf[dh];
    // P and dS leave the fp32 accumulators as 16-bit column operands right away (half the registers through the dV /
    // dK products: no spills at 128 VGPRs); pf[c] / dsf[c] = query blocks 2c, 2c + 1
    frag pf[4], dsf[4];
    // DROP: this lane's key sits in word (query, lg_f) at one fixed bit
    const int key_i = (int)row;
    const int keep_word = (key_i >> 2) & 3, keep_bit = ((key_i >> 5) << 3) | (((key_i >> 4) & 1) << 2) | (key_i & 3);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int qbk = 2 * c + e;
            const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + qbk * 16 + lg * 4);
            const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(del_s + qbk * 16 + lg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[qbk][j], p.scale_log2e, mk) - l4[j]);
                if constexpr (DROP) {
                    const uint32_t wrd = keep_s[(qbk * 16 + lg * 4 + j) * 4 + keep_word];
                    const float m = ((wrd >> keep_bit) & 1u) ? p.inv_keep : 0.f;
                    s[qbk][j] = pr * m;                          // P~ = P o keep / (1 - p)
                    dp[qbk][j] = pr * (dp[qbk][j] * m - d4[j]);  // dS
                } else {
                    s[qbk][j] = pr;                          // P
                    dp[qbk][j] = pr * (dp[qbk][j] - d4[j]);  // dS
                }
            }
        }
        pf[c] = pack2<T>(s[2 * c], s[2 * c + 1]);
        dsf[c] = pack2<T>(dp[2 * c], dp[2 * c + 1]);
    }
    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dk[j] = dv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dv[db] = Mfma<T>::run(tr_frag<T>(dop, c, db, li, lg), pf[c], dv[db]);
            dk[db] = Mfma<T>::run(tr_frag<T>(qp, c, db, li, lg), dsf[c], dk[db]);
        }
    }
    T* dkb = reinterpret_cast<T*>(p.dk) + (long long)b * p.T * ostride + hoff;
    T* dvb = reinterpret_cast<T*>(p.dv) + (long long)b * p.T * ostride + hoff;
    // column sums of this tile's dq / dk / dv: [8 waves][3][64] floats in the 12 KiB between the K image (kp, 20 KiB over
    // qs | dos) and the padded images — dead since S and dP were formed
    float* const cs = reinterpret_cast<float*>(smem + P_BYTES);
    auto colsum_rows = [&](const half4 (&v)[4], int t) {  // the wave's 16 rows of one tensor, as stored
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            f32x4_t r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = row_sum((float)v[db][j]);
            if (li == 15) *reinterpret_cast<f32x4_t*>(cs + (wid * 3 + t) * HD + db * 16 + lg * 4) = r;
        }
    };
    {
        half4 hk[4], hv[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            hk[db] = __builtin_convertvector(dk[db] * p.scale, half4);
            hv[db] = __builtin_convertvector(dv[db], half4);
            *reinterpret_cast<half4*>(dkb + row * ostride + db * 16 + lg * 4) = hk[db];
            *reinterpret_cast<half4*>(dvb + row * ostride + db * 16 + lg * 4) = hv[db];
        }
        if (p.cs_partial) {
            colsum_rows(hk, 1);
            colsum_rows(hv, 2);
        }
    }

    __syncthreads();  // every wave has read the last of Q, dO and their transposes
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // elements 0..3 / 4..7 of dsf[c]: queries (2c) 16 + 4 lg + 0..3 / (2c + 1) 16 + 4 lg + 0..3
        const half4 lo = {dsf[c][0], dsf[c][1], dsf[c][2], dsf[c][3]}, hi = {dsf[c][4], dsf[c][5], dsf[c][6], dsf[c][7]};
        *reinterpret_cast<half4*>(dst + row * DS_ROW + ((2 * c) * 16 + lg * 4) * 2) = lo;
        *reinterpret_cast<half4*>(dst + row * DS_ROW + ((2 * c + 1) * 16 + lg * 4) * 2) = hi;
    }
    __syncthreads();
    // dQ^T[d][query] = sum_k K^T[d][k] dS^T[k][query] for the wave's query block `wid`
    f32x4_t dq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dq[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const frag dsf = tr_frag<T, DS_ROW>(dst, c, wid, li, lg);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = Mfma<T>::run(tr_frag<T>(kp, c, db, li, lg), dsf, dq[db]);
    }
    T* dqb = reinterpret_cast<T*>(p.dq) + (long long)b * p.T * ostride + hoff;
    half4 hq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        hq[db] = __builtin_convertvector(dq[db] * p.scale, half4);
        *reinterpret_cast<half4*>(dqb + row * ostride + db * 16 + lg * 4) = hq[db];
    }
    if (p.cs_partial) {
        colsum_rows(hq, 0);
        __syncthreads();
        if (tid < 3 * HD) {  // fixed order over the 8 waves: deterministic
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) acc += cs[w * 3 * HD + tid];
            p.cs_partial[((long long)b * p.H + h) * 3 * HD + tid] = acc;
        }
    }
}

// out[t][s][h * 64 + d] = sum over the sequences of sample s of partial[b][h][t][d]; grid = (3 H, S), 64 threads
__global__ __launch_bounds__(64) void attention_colsum_finish_kernel(const float* __restrict__ partial, int seq_per_sample, int H,
                                                                     int S, float* __restrict__ out) {
    const int t = blockIdx.x / H, h = blockIdx.x % H, s = blockIdx.y, d = threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < seq_per_sample; ++i)
        acc += partial[(((long long)s * seq_per_sample + i) * H + h) * 3 * HD + t * HD + d];
    out[((long long)t * S + s) * H * HD + h * HD + d] = acc;
}

}  // namespace

int bf_launch_attention_bwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask,
                            const unsigned char* d_mask_off, const void* d_out, const void* d_dout, const float* d_lse,
                            float* d_delta, void* d_dq, void* d_dk, void* d_dv, int dtype, int B, int T, int H,
                            int head_dim, long long token_stride, float scaling, hipStream_t stream,
                            const uint32_t* d_keep_bits, float inv_keep, int samples, float* d_cs_partial, float* d_colsum) {
    if (!d_q || !d_k || !d_v || !d_out || !d_dout || !d_lse || !d_delta || !d_dq || !d_dk || !d_dv)
        BF_FAIL("bf_attention_bwd: NULL argument");
    if (d_colsum && (T != TT || !d_cs_partial || samples < 1 || B % samples || samples > 65535))
        BF_FAIL("bf_attention_bwd: column sums need one-tile sequences (T = %d), a partial buffer and B %% samples == 0", TT);
    if (dtype != BF_DT_BF16 && dtype != BF_DT_F16) BF_FAIL("bf_attention_bwd: dtype must be bf16 or fp16");
    if (head_dim != HD) BF_FAIL("bf_attention_bwd: head size %d (only %d)", head_dim, HD);
    if (B < 1 || H < 1 || T < TT || T % TT) BF_FAIL("bf_attention_bwd: T=%d must be a positive multiple of %d", T, TT);
    if (B > 65535 || H > 65535) BF_FAIL("bf_attention_bwd: B or H exceeds the grid");
    if (token_stride < (long long)H * HD || token_stride % 8) BF_FAIL("bf_attention_bwd: bad token stride %lld", token_stride);
    const uintptr_t al = (uintptr_t)d_q | (uintptr_t)d_k | (uintptr_t)d_v | (uintptr_t)d_out | (uintptr_t)d_dout |
                         (uintptr_t)d_dq | (uintptr_t)d_dk | (uintptr_t)d_dv | (uintptr_t)d_lse | (uintptr_t)d_delta;
    if (al & 15) BF_FAIL("bf_attention_bwd: pointers must be 16-byte aligned");
    if (d_mask && ((uintptr_t)d_mask & 15)) BF_FAIL("bf_attention_bwd: mask must be 16-byte aligned");
    BwdParams p;
    p.q = d_q;
    p.k = d_k;
    p.v = d_v;
    p.mask = d_mask;
    p.mask_off = d_mask_off;
    p.o = d_out;
    p.dout = d_dout;
    p.lse = d_lse;
    p.delta = d_delta;
    p.dq = d_dq;
    p.dk = d_dk;
    p.dv = d_dv;
    p.tok_stride = token_stride;
    p.B = B;
    p.T = T;
    p.H = H;
    p.scale = scaling;
    p.scale_log2e = scaling * 1.4426950408889634f;
    p.keep_bits = d_keep_bits;
    p.inv_keep = inv_keep;
    p.cs_partial = d_colsum ? d_cs_partial : nullptr;
    const dim3 grid(T / TT, H, B);
    auto finish = [&]() -> int {
        if (d_colsum)
            hipLaunchKernelGGL(attention_colsum_finish_kernel, dim3(3 * H, samples), dim3(64), 0, stream, d_cs_partial, B / samples,
                               H, samples, d_colsum);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    };
    if (d_keep_bits) {
        if ((uintptr_t)d_keep_bits & 3) BF_FAIL("bf_attention_bwd: keep bits must be 4-byte aligned");
        if (T == TT) {
            if (dtype == BF_DT_BF16) hipLaunchKernelGGL((attention_bwd_tile_kernel<__bf16, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((attention_bwd_tile_kernel<_Float16, true>), grid, dim3(512), 0, stream, p);
        } else if (dtype == BF_DT_BF16) {
            hipLaunchKernelGGL((attention_bwd_dq_kernel<__bf16, true>), grid, dim3(256), 0, stream, p);
            hipLaunchKernelGGL((attention_bwd_dkv_kernel<__bf16, true>), grid, dim3(512), 0, stream, p);
        } else {
            hipLaunchKernelGGL((attention_bwd_dq_kernel<_Float16, true>), grid, dim3(256), 0, stream, p);
            hipLaunchKernelGGL((attention_bwd_dkv_kernel<_Float16, true>), grid, dim3(512), 0, stream, p);
        }
        return finish();
    }
    if (T == TT) {  // one tile: dQ, dK, dV in one launch
        if (dtype == BF_DT_BF16) hipLaunchKernelGGL(attention_bwd_tile_kernel<__bf16>, grid, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL(attention_bwd_tile_kernel<_Float16>, grid, dim3(512), 0, stream, p);
        return finish();
    }
    if (dtype == BF_DT_BF16) {
        hipLaunchKernelGGL(attention_bwd_dq_kernel<__bf16>, grid, dim3(256), 0, stream, p);
        hipLaunchKernelGGL(attention_bwd_dkv_kernel<__bf16>, grid, dim3(512), 0, stream, p);
    } else {
        hipLaunchKernelGGL(attention_bwd_dq_kernel<_Float16>, grid, dim3(256), 0, stream, p);
        hipLaunchKernelGGL(attention_bwd_dkv_kernel<_Float16>, grid, dim3(512), 0, stream, p);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
