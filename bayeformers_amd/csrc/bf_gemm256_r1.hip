// bf_gemm256_r1.hip — DEVELOPER BUILDS ONLY (-DBF_DEV): the round-1 kernel, kept as the same-run A/B baseline of
// bf_gemm256.hip (tools/gemm_bench).  Not part of the product library.
// the fast path of the sampled-weight GEMM on gfx950: y[s] = x[s] W_s^T + b_s
// (F.linear at /root/reference/bayeformers/nn/layers/linear.py:104, all S samples in one launch).
//
// Shape of the kernel (MI355X: 256 CUs, 160 KiB LDS/CU, wave64, v_mfma_f32_16x16x32_{bf16,f16}):
//   * one 256(m) x 256(n) output tile per 512-thread workgroup (8 waves = 2(m) x 4(n), 128 x 64 per wave,
//     128 fp32 accumulator registers per lane), K walked in steps of 64;
//   * both operands are K-contiguous ([M][K] activations, [N][K] sampled weights) and are DMA'd straight into LDS
//     with global_load_lds_dwordx4 (no VGPR round trip), double-buffered: 2 x (256+256) rows x 128 B = 128 KiB;
//   * LDS rows are 128 B; the 16-byte chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7).  The DMA
//     destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address and again on the
//     fragment read, which makes every ds_read_b128 of an MFMA fragment bank-conflict-free;
//   * the MFMA runs with swapped operands (D rows = n, cols = m) so each lane owns 4 consecutive output features
//     of one row of y and the epilogue stores 8 B (bf16/fp16) or 16 B (fp32) per lane;
//   * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs (private 4 MiB L2) gets a contiguous run of tiles
//     in (sample, n-tile, m-tile) order, so the W_s n-panel and the x m-panels it re-reads stay in its own L2.
// Requirements: K % 64 == 0, 16-byte aligned operands; M and N are arbitrary (edge rows are clamped on load and
// masked on store).  Everything else goes to the generic kernel in bf_gemm.hip.
#include <stdlib.h>

#include "bf_common.h"
#include "bf_gemm_params.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
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

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    // 64 lanes x 16 B: lane i lands at lds_wave_base + 16*i (wave-uniform base)
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}


// acc[nb][mb][j] starts at bias[n] (n = the lane's 4 consecutive features of fragment nb): the bias add costs no
// epilogue work and its loads overlap the first DMA wait of the tile.
__device__ __forceinline__ void init_acc(f32x4_t (&acc)[4][8], const float* bias, int n0, int N, int wn, int lane) {
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
        for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = b;
    }
}

// XCD-aware bijective remap of the flat block id (block b runs on XCD b % 8, observed; speed only).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg) {
    const unsigned xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}


// logical tile index -> (sample, tm, tn).  Inside a sample tiles are ordered (group of 4 m-tiles, tn, tm in group):
// any 32 consecutive logical tiles (one XCD's concurrent set) span about 4-11 m-panels x 3-8 n-panels, which keeps
// the panels every CU of the XCD re-reads in that XCD's private L2.
__device__ __forceinline__ void tile_coords(unsigned lt, int tiles_m, int tiles_n, int& s, int& tm, int& tn) {
    const unsigned per_s = (unsigned)(tiles_m * tiles_n);
    s = lt / per_s;
    const unsigned r = lt - s * per_s;
    const unsigned grp = r / (4u * tiles_n);
    const unsigned in = r - grp * 4u * tiles_n;
    const unsigned gm = min(4u, (unsigned)tiles_m - grp * 4u);
    tn = in / gm;
    tm = grp * 4 + (in - tn * gm);
}

// Epilogue: accumulators -> LDS as row-major rows of YT (in passes, see epilogue_passes) -> whole-row 16-byte global
// stores.  A lane's fragment registers are 4 consecutive n of one m (8 B for 16-bit outputs): written with
// ds_write_b64 into rows padded by 16 B (2-way bank aliasing only), then every wave streams rows back with
// ds_read_b128 and stores them with dwordx4 (512 contiguous bytes per row for bf16).
constexpr int EPI_PAD = 16;
#ifndef BF_NT_STORES
#define BF_NT_STORES 1
#endif
constexpr bool NT_STORES = BF_NT_STORES;
// ------------------------------------------------------------------------------------------------------------
// The kernel: persistent ping-pong.
// The two waves that share a SIMD belong to different wave groups (G0 = waves 0-3 = rows 0..127 of the tile,
// G1 = waves 4-7 = rows 128..255) and run the same slot sequence one slot apart:
//
//      slot:   4t        4t+1      4t+2      4t+3      4t+4
//      G0:     L0(t)     M0(t)     L1(t)     M1(t)     L0(t+1) ...
//      G1:     M1(t-1)   L0(t)     M0(t)     L1(t)     M1(t)   ...
//
// L = 12 ds_read_b128 (the fragments of one 32-deep half of the k-tile) + lgkmcnt(0); M = 32 MFMAs on registers.
// Every slot ends in one workgroup barrier, so a SIMD always has one wave on the matrix pipe while its partner is
// on the LDS pipe.  The LDS DMA of k-step t+1 is issued by each wave at the start of its own L0(t) and is only
// waited for at the last barrier before slot 4(t+1), i.e. it has 3-4 slots (>= 1500 cycles) to land.
// One workgroup per CU walks tiles b, b+grid, b+2*grid, ... (same XCD every time, so the L2-aware tile order is
// preserved), and:
//   * the LDS DMA issued in the LAST k-step of a tile fetches k-step 0 of the workgroup's NEXT tile into the buffer
//     that would otherwise idle, and is retired by the k-loop's existing waits — the next tile starts without a
//     cold-start load;
//   * the epilogue stages the accumulators through the just-consumed buffer in passes of 64 rows (32 for fp32
//     outputs), so the prefetched stage in the other buffer survives it; its global stores are not waited for
//     until the next tile's first k-step retires them together with that step's DMA.
template <typename YT>
__device__ __forceinline__ void epilogue_passes(char* region, const f32x4_t (&acc)[4][8], const float* bias, YT* y,
                                                int m0, int n0, int M, int N, int wm, int wn, int wid, int lane,
                                                int act) {
    constexpr int ROW = TN * (int)sizeof(YT) + EPI_PAD;
    constexpr int PASS_ROWS = sizeof(YT) == 4 ? 32 : 64;
    constexpr int PASSES = TM / PASS_ROWS;            // 4 or 8
    constexpr int PASSES_PER_GROUP = 128 / PASS_ROWS;  // 2 or 4
    constexpr int MB_PER_PASS = PASS_ROWS / 16;        // 4 or 2
    constexpr int CHUNKS = TN * (int)sizeof(YT) / 16;  // 16-byte chunks per row: 32 or 64
    constexpr int EPC = 16 / (int)sizeof(YT);
    constexpr int ROWS_PER_INST = 64 / CHUNKS > 0 ? 64 / CHUNKS : 1;  // 2 or 1
    constexpr int INSTS = PASS_ROWS / 8 / ROWS_PER_INST;                // per wave per pass: 4
    const bool vec_ok = (N % EPC) == 0 && ((uintptr_t)y % 16) == 0;
    // per-lane constants of the two access patterns (32-bit: one sample's y has < 2^31 elements, checked on the host)
    const int wr_off = (lane & 15) * ROW + (wn * 64 + (lane >> 4) * 4) * (int)sizeof(YT);
    const int rd_row = wid * (PASS_ROWS / 8) + (CHUNKS < 64 ? lane / CHUNKS : 0);
    const int rd_q = CHUNKS < 64 ? lane % CHUNKS : lane;
    const int rd_off = rd_row * ROW + rd_q * 16;
    const int n = n0 + rd_q * EPC;
    const bool n_ok = n < N, n_full = vec_ok && n + EPC <= N;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        __builtin_amdgcn_s_barrier();  // the region is free (k-loop reads / previous pass's row reads are done)
        if (wm == pass / PASSES_PER_GROUP) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                for (int k = 0; k < MB_PER_PASS; ++k) {
                    const int mb = (pass % PASSES_PER_GROUP) * MB_PER_PASS + k;
                    const f32x4_t v = bf_apply_act(acc[nb][mb], act);
                    char* dst = region + wr_off + k * 16 * ROW + nb * 16 * (int)sizeof(YT);
                    if constexpr (sizeof(YT) == 4)
                        *reinterpret_cast<f32x4_t*>(dst) = v;
                    else if constexpr (__is_same(YT, __bf16))
                        *reinterpret_cast<bf16x4_t*>(dst) = __builtin_convertvector(v, bf16x4_t);
                    else
                        *reinterpret_cast<f16x4_t*>(dst) = __builtin_convertvector(v, f16x4_t);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int mrow = m0 + pass * PASS_ROWS + rd_row;
#pragma unroll
        for (int it = 0; it < INSTS; ++it) {
            const int m = mrow + it * ROWS_PER_INST;
            if (m < M && n_ok) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(region + rd_off + it * ROWS_PER_INST * ROW);
                YT* o = y + (unsigned)(m * N + n);
                if (n_full) {
                    // streaming store: y is not re-read by this kernel, keep it from evicting operand panels in L2
                    if (NT_STORES) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(o));
                    else *reinterpret_cast<f32x4_t*>(o) = v;
                } else {
                    const YT* e = reinterpret_cast<const YT*>(&v);
                    for (int j = 0; j < EPC; ++j)
                        if (n + j < N) o[j] = e[j];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // row reads done before the region is rewritten
    }
}

template <typename T, typename YT>
__global__ __launch_bounds__(512, 2) void gemm256_persist_kernel(const GemmParams p) {
    using frag = typename Mfma16<T>::frag;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_nl = p.tiles_n * p.layers;  // the L layers sharing x form one wide row of n-tiles per m-panel
    const unsigned total = (unsigned)(p.tiles_m * tiles_nl * p.S);

    const int prow = lane >> 3;
    const int kc8 = ((lane & 7) ^ ((((wid & 1) << 2) + (lane >> 4)) & 7)) * 8;

    // One tile's DMA sources: wave-uniform sample bases (SGPRs) + eight 32-bit per-lane element offsets.
    struct Src {
        const T* xb;
        const T* wb;
        unsigned xo[4], wo[4];
    };
    // s = index of the (layer, sample) pair in w / bias / y; the activations only depend on the sample
    auto tile_setup = [&](unsigned vb, Src& t, int& s, int& m0, int& n0) {
        int tm, tn, xs;
        tile_coords(xcd_remap(vb, total), p.tiles_m, tiles_nl, xs, tm, tn);
        const int layer = tn / p.tiles_n;
        tn -= layer * p.tiles_n;
        s = layer * p.S + xs;
        m0 = tm * TM;
        n0 = tn * TN;
        t.xb = reinterpret_cast<const T*>(p.x) + (long long)xs * p.x_sstride;
        t.wb = reinterpret_cast<const T*>(p.w) + (long long)s * N * K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + wid) * 8 + prow;
            t.xo[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)K + kc8;
            t.wo[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)K + kc8;
        }
    };
    auto stage = [&](const Src& t, int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const T* xk = t.xb + kt * TK;
        const T* wk = t.wb + kt * TK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(xk + t.xo[i], base + (i * 8 + wid) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wk + t.wo[i], base + X_BYTES + (i * 8 + wid) * 1024);
    };

    const int fsw = (lane >> 1) & 7;
    const int foff0 = (lane & 15) * ROW_BYTES + ((((lane >> 4)) ^ fsw) << 4);
    const int foff1 = (lane & 15) * ROW_BYTES + (((4 + (lane >> 4)) ^ fsw) << 4);
    const int xfrag_base = wm * 128 * ROW_BYTES;
    const int wfrag_base = X_BYTES + wn * 64 * ROW_BYTES;

    const int nk = K / TK;
    unsigned vb = blockIdx.x;
    Src cur;
    int s, m0, n0;
    tile_setup(vb, cur, s, m0, n0);
    int g = 0;  // running k-step counter: step g lives in LDS buffer g & 1
    stage(cur, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4_t acc[4][8];
    frag wf[4], xf[8];

    // the four slots of one k-step; `dma()` issues this step's LDS DMA at the top of L0
    auto kstep = [&](auto&& dma) {
        const char* sb = smem + (g & 1) * STAGE_BYTES;
        dma();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wf[i] = *reinterpret_cast<const frag*>(sb + wfrag_base + i * 16 * ROW_BYTES + foff0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xf[j] = *reinterpret_cast<const frag*>(sb + xfrag_base + j * 16 * ROW_BYTES + foff0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wf[i] = *reinterpret_cast<const frag*>(sb + wfrag_base + i * 16 * ROW_BYTES + foff1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xf[j] = *reinterpret_cast<const frag*>(sb + xfrag_base + j * 16 * ROW_BYTES + foff1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        ++g;
    };

    for (;;) {
        const bool has_next = vb + gridDim.x < total;
        if (wm == 1) __builtin_amdgcn_s_barrier();  // G1 runs one slot behind G0
        init_acc(acc, p.bias ? p.bias + (long long)s * N : nullptr, n0, N, wn, lane);

        for (int kt = 0; kt + 1 < nk; ++kt)
            kstep([&] { if (!(p.flags & 1)) stage(cur, (p.flags & 64) ? 0 : kt + 1, (g & 1) ^ 1); });
        // last k-step: its DMA slot fetches k-step 0 of this workgroup's next tile
        unsigned vbn = vb + gridDim.x;
        asm volatile("" : "+s"(vbn));  // keep the next tile's address arithmetic out of the k-loop's live ranges
        kstep([&] {
            if (has_next && !(p.flags & 1)) {
                Src nxt;
                int s2, m2, n2;
                tile_setup(vbn, nxt, s2, m2, n2);
                stage(nxt, 0, (g & 1) ^ 1);
            }
        });
        if (wm == 0) __builtin_amdgcn_s_barrier();  // rejoin: balance G1's leading barrier

        // the last consumed buffer is (g-1)&1; buffer g&1 already holds k-step 0 of the next tile
        YT* y = reinterpret_cast<YT*>(p.y) + (long long)s * M * N;
        if (!(p.flags & 8))
            epilogue_passes<YT>(smem + ((g - 1) & 1) * STAGE_BYTES, acc, nullptr, y, m0, n0, (p.flags & 16) ? 0 : M, N, wm,
                                wn, wid, lane, p.act);
        if (!has_next) break;
        vb = vbn;
        tile_setup(vb, cur, s, m0, n0);
        __builtin_amdgcn_s_barrier();  // every wave has left the epilogue before the region takes DMA again
    }
}

template <typename T>
int launch256(const GemmParams& p, int y_dtype, hipStream_t stream) {
    const uint32_t tiles = (uint32_t)(p.tiles_m * p.tiles_n * p.S * p.layers);
    // persistent: one workgroup per CU (a grid that is a multiple of 8 keeps a workgroup's tiles on one XCD)
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
        else n_cu = prop.multiProcessorCount / 8 * 8;
        if (n_cu < 8) n_cu = 8;
    }
    const dim3 grid(tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu);
    if (y_dtype == BF_DT_F32)
        hipLaunchKernelGGL((gemm256_persist_kernel<T, float>), grid, dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm256_persist_kernel<T, T>), grid, dim3(512), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

#if 0
bool bf_gemm256_supported(int x_dtype, int w_dtype, int y_dtype, int S, int M, int N, int K, const void* d_x,
                          const void* d_w, int64_t x_sample_stride) {
    if (w_dtype != BF_DT_BF16 && w_dtype != BF_DT_F16) return false;
    if (x_dtype != w_dtype) return false;
    if (y_dtype != w_dtype && y_dtype != BF_DT_F32) return false;
    if (K % TK != 0 || K < TK) return false;
    if (N % 4 != 0) return false;  // 16-byte bias rows / output chunks
    if (((uintptr_t)d_x | (uintptr_t)d_w) & 15) return false;
    if (((size_t)x_sample_stride * 2) % 16 != 0) return false;
    if ((long long)M * K >= (1ll << 32) || (long long)N * K >= (1ll << 32) || (long long)M * N >= (1ll << 31)) return false;
    const long long tiles = (long long)((M + TM - 1) / TM) * ((N + TN - 1) / TN) * S;  // S counts (layer, sample) pairs
    if (tiles > 0x7FFFFFFFll) return false;
    (void)S;
    return true;
}

#endif
int bf_launch_gemm256_r1(const GemmParams& p0, int w_dtype, int y_dtype, hipStream_t stream) {
    GemmParams p = p0;
    const char* ab = getenv("BF_GEMM_ABLATE");
    p.flags = ab ? atoi(ab) : 0;
    if (p.layers < 1) p.layers = 1;
    p.tiles_m = (p.M + TM - 1) / TM;
    p.tiles_n = (p.N + TN - 1) / TN;
    if (w_dtype == BF_DT_BF16) return launch256<__bf16>(p, y_dtype, stream);
    return launch256<_Float16>(p, y_dtype, stream);
}
