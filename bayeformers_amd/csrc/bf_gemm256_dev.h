// bf_gemm256_dev.h — device helpers and the tile-schedule interface shared by the translation units of the 256-wide
// sampled-weight GEMM (bf_gemm256.hip: burst / unit-ring forms; bf_gemm256_r5.hip: the five-slot ring of the forward).
// Everything here is internal (anonymous namespace per translation unit); the C-ABI is include/bayeformers_amd.h.
#pragma once
#include <type_traits>

#include "bf_common.h"
#include "bf_gemm_params.h"

// schedule policy bits (build_schedule): 1 = odd workgroups run their tiles in reverse order, 2 = fixed full-height
// tiles (no balancing), 4 = XCD-chunked dealing, 8 = every XCD walks a contiguous share of each height class, bits 4-7 =
// tallest tile (0 = 8 units), bits 8-11 = columns per group of the locality order (0 = 4), 0x1000 = the columns that get one
// tile more are the first ones (columns of a sample share their row cuts), 0x2000 = the column-group size is chosen per shape
// by the modelled fabric fetch.  0x300C since round 6 (12 before: same times, 9 % more fabric reads per BERT-base step:
// profiles/r6b_sched_l2_model.md)
#ifndef BF_SCHED_POLICY
#define BF_SCHED_POLICY 0x300C
#endif

// The tile schedule of a shape (built on the host once per device, kept in device memory): bf_gemm256.hip.
struct Gemm256Sched {
    int4* d_table;
    int rounds, grid;
};
int bf_gemm256_get_schedule(int S, int layers, int tiles_n, int M, int policy, hipStream_t stream, Gemm256Sched& out);

// the five-slot-ring forward kernel (bf_gemm256_r5.hip)
bool bf_gemm256_r5_supported(const GemmParams& p, int w_dtype, int y_dtype);
int bf_launch_gemm256_r5(const GemmParams& p, int w_dtype, hipStream_t stream, int grid);
int bf_launch_gemm256_r5_nn(const GemmParams& p, int dtype, hipStream_t stream, int grid);
bool bf_gemm256_r5_tn_supported(const GemmParams& p);
int bf_launch_gemm256_r5_tn(const GemmParams& p, int dtype, hipStream_t stream, int grid);

namespace {

constexpr int TN = 256, TK = 64;
constexpr int UNIT = 32;                          // rows per schedule unit (one 16-row block per wave group)
constexpr int HMAX = 8;                           // tallest tile: 256 rows = 128 fp32 accumulators per lane (9 and 10
                                                  // were measured: they spill and run 6-8 % slower per flop)
constexpr int HMIN = 4;                           // shortest tile the k-loop is instantiated for
constexpr int TM = HMAX * UNIT;                   // rows of x a stage holds
constexpr int XPIECES = TM / 64;                  // 1 KiB DMA pieces per wave for the x rows of a stage: 4
constexpr int ROW_BYTES = TK * 2;                 // 128
constexpr int X_BYTES = TM * ROW_BYTES;           // 32 KiB
constexpr int STAGE_BYTES = (TM + TN) * ROW_BYTES;  // 64 KiB

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

template <>
struct Mfma16<float> {  // a 16-byte fragment = 4 consecutive k of fp32: four 16x16x4 MFMAs (bf_gemm256_r5.hip orders them itself)
    using frag = f32x4_t;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], c, 0, 0, 0);
        return c;
    }
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    // 64 lanes x 16 B: lane i lands at lds_wave_base + 16*i (wave-uniform base)
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// acc[nb][mb][j] starts at bias[n] (n = the lane's 4 consecutive features of fragment nb): the bias add costs no
// epilogue work and its loads overlap the first DMA wait of the tile.
template <int H>
__device__ __forceinline__ void init_acc(f32x4_t (&acc)[4][H], const float* bias, int n0, int N, int wn, int lane) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        f32x4_t b = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const int n = n0 + wn * 64 + nb * 16 + (lane >> 4) * 4;
            if (n + 3 < N) {
                b = *reinterpret_cast<const f32x4_t*>(bias + n);  // [S][N] fp32 rows, N % 4 == 0 checked on the host
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < N) b[j] = bias[n + j];
            }
        }
#pragma unroll
        for (int mb = 0; mb < H; ++mb) acc[nb][mb] = b;
    }
}

// compile-time loop (immediate offsets for the inline-asm LDS reads)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// output stores are nontemporal (plain and sc1 / sc0 sc1 write-through stores were measured: profiles/r3t_gemm_output_store_policy.txt)
__device__ __forceinline__ void gemm_st16(f32x4_t* p, f32x4_t v) { __builtin_nontemporal_store(v, p); }


// act() of a 16-byte chunk of YT outputs (8 x 16-bit or 4 x fp32), computed in fp32
template <typename YT>
__device__ __forceinline__ f32x4_t act_chunk(f32x4_t c, int act) {
    if constexpr (sizeof(YT) == 4) {
        return bf_apply_act(c, act);
    } else {
        typedef __attribute__((ext_vector_type(8))) YT yt8;
        typedef __attribute__((ext_vector_type(4))) YT yt4;
        const yt8 h = __builtin_bit_cast(yt8, c);
        const f32x4_t lo = bf_apply_act<true>(f32x4_t{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, act);
        const f32x4_t hi = bf_apply_act<true>(f32x4_t{(float)h[4], (float)h[5], (float)h[6], (float)h[7]}, act);
        const yt4 a = __builtin_convertvector(lo, yt4), b = __builtin_convertvector(hi, yt4);
        const yt8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_bit_cast(f32x4_t, r);
    }
}

// Wave-private epilogue: a wave's part of the tile is 16 h rows x 64 features — for every
// 16-row block a full 128-byte line (256 bytes of fp32) per row.  The block goes through a 2 / 4 KiB slice of LDS that
// only this wave touches (XOR-swizzled like the operand rows: conflict-free ds_write_b64 / ds_read_b128) and leaves as
// 16-byte stores of whole lines, 8 (4) rows per instruction.  Nothing is exchanged between waves, so the epilogue has
// no barrier at all: wave group 0 starts it a slot before group 1, and every wave runs it at its own pace.
// `scratch` = this wave's 8 KiB of the stage buffer the k-loop consumed last (every fragment read of it is complete
// when a wave gets here, see the kernel).
template <typename YT, int H, int SLICES = 2>
__device__ __forceinline__ void epilogue_wave(char* scratch, const f32x4_t (&acc)[4][H], YT* y, YT* y2, int m0,
                                              int m_end, int n0, int N, int wm, int wn, int lane, int act,
                                              const YT* gpre = nullptr) {  // gpre: the stored rows are multiplied by act'(gpre)
    constexpr int ROWB = 64 * (int)sizeof(YT);   // 128 or 256
    constexpr int BLK = 16 * ROWB;               // 2 or 4 KiB
    constexpr int CH = ROWB / 16;                // 16-byte chunks per row: 8 or 16
    constexpr int RPI = 64 / CH;                 // rows per store instruction: 8 or 4
    constexpr int NI = 16 / RPI;                 // store instructions per block: 2 or 4
    constexpr int EPC = 16 / (int)sizeof(YT);
    asm volatile("" : "+v"(lane));               // offsets recomputed per tile, not hoisted into the k-loop's registers
    const bool vec_ok = (N % EPC) == 0 && ((uintptr_t)y % 16) == 0 && (!y2 || ((uintptr_t)y2 % 16) == 0);
    const int m_l = lane & 15, q = lane >> 4;
    // write side: lane holds 4 consecutive features 16 nb + 4 q of row m_l
    const int wsw = sizeof(YT) == 2 ? (m_l >> 1) & 7 : m_l;
    int wr_off[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int c = sizeof(YT) == 2 ? nb * 2 + (q >> 1) : nb * 4 + q;
        wr_off[nb] = m_l * ROWB + ((c ^ wsw) << 4) + (sizeof(YT) == 2 ? (q & 1) * 8 : 0);
    }
    // read side: instruction it covers rows it * RPI + lane / CH, chunk lane % CH
    const int rr = lane / CH, rc = lane % CH;
    const int n = n0 + wn * 64 + rc * EPC;
    const bool n_ok = n < N, n_full = vec_ok && n + EPC <= N;
    auto write_block = [&](int mb) {
        char* R = scratch + (SLICES == 2 ? (mb & 1) * BLK : 0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const f32x4_t v = (y2 || gpre) ? acc[nb][mb] : bf_apply_act<sizeof(YT) == 2>(acc[nb][mb], act);
            char* dst = R + wr_off[nb];
            if constexpr (sizeof(YT) == 4)
                *reinterpret_cast<f32x4_t*>(dst) = v;
            else if constexpr (__is_same(YT, __bf16))
                *reinterpret_cast<bf16x4_t*>(dst) = __builtin_convertvector(v, bf16x4_t);
            else
                *reinterpret_cast<f16x4_t*>(dst) = __builtin_convertvector(v, f16x4_t);
        }
    };
    // software pipeline over the blocks: block mb + 1 is converted and written (other slice) behind the row reads of
    // block mb — one LDS round trip per block instead of two
    // gpre: the 16-byte chunks of the pre-activation that belong to the rows of block mb are fetched TWO blocks ahead (they come
    // from HBM: written by the forward pass long ago) into registers the accumulators of the blocks already written out gave up
    constexpr int PF = 2;
    f32x4_t gp[PF + 1][NI];
    auto fetch_gpre = [&](int mb) {
        if constexpr (sizeof(YT) == 2) {
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int m = m0 + (2 * mb + wm) * 16 + it * RPI + rr;
                gp[mb % (PF + 1)][it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (m < m_end && n_full) gp[mb % (PF + 1)][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(gpre + (unsigned)(m * N + n)));
            }
        }
    };
    write_block(0);
    if (gpre) {
#pragma unroll
        for (int b = 0; b < PF && b < H; ++b) fetch_gpre(b);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mb = 0; mb < H; ++mb) {
        const char* R = scratch + (SLICES == 2 ? (mb & 1) * BLK : 0);
        f32x4_t rows[NI];
        if (gpre && mb + PF < H) fetch_gpre(mb + PF);
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int r = it * RPI + rr;
            const int rsw = sizeof(YT) == 2 ? (r >> 1) & 7 : r;
            rows[it] = *reinterpret_cast<const f32x4_t*>(R + r * ROWB + ((rc ^ rsw) << 4));
        }
        if (SLICES == 2 && mb + 1 < H) write_block(mb + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (SLICES == 1 && mb + 1 < H) {  // one slice: the next block goes in once this one's rows are in registers
            write_block(mb + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int m = m0 + (2 * mb + wm) * 16 + it * RPI + rr;
            if (m < m_end && n_ok) {
                f32x4_t v = rows[it];
                if (y2) {
                    YT* o2 = y2 + (unsigned)(m * N + n);
                    if (n_full) {
                        gemm_st16(reinterpret_cast<f32x4_t*>(o2), v);
                    } else {
                        const YT* e = reinterpret_cast<const YT*>(&v);
                        for (int j = 0; j < EPC; ++j)
                            if (n + j < N) o2[j] = e[j];
                    }
                    v = act_chunk<YT>(v, act);
                }
                YT* o = y + (unsigned)(m * N + n);
                if constexpr (sizeof(YT) == 2) {
                    if (gpre && n_full) {  // (whole 16-byte chunks only: the host requires N % 8 == 0 for this form)
                        typedef __attribute__((ext_vector_type(8))) YT yt8;
                        const yt8 g = __builtin_bit_cast(yt8, v);
                        const yt8 x = __builtin_bit_cast(yt8, gp[mb % (PF + 1)][it]);
                        yt8 r;
#pragma unroll
                        for (int j = 0; j < 8; ++j) r[j] = (YT)((float)g[j] * bf_gelu_grad((float)x[j]));
                        v = __builtin_bit_cast(f32x4_t, r);
                    }
                }
                if (n_full) {
                    gemm_st16(reinterpret_cast<f32x4_t*>(o), v);
                } else {
                    const YT* e = reinterpret_cast<const YT*>(&v);
                    for (int j = 0; j < EPC; ++j)
                        if (n + j < N) o[j] = e[j];
                }
            }
        }
    }
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

}  // namespace
