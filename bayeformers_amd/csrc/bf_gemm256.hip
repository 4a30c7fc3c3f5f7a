// bf_gemm256.hip — the fast path of the sampled-weight GEMM on gfx950: y[s] = x[s] W_s^T + b_s
// (F.linear at /root/reference/bayeformers/nn/layers/linear.py:104, all S samples in one launch).
//
// Shape of the kernel (MI355X: 256 CUs, 160 KiB LDS/CU, wave64, v_mfma_f32_16x16x32_{bf16,f16}):
//   * one (32 h)(m) x 256(n) output tile per 512-thread workgroup, h = 4..8 (128..256 rows), 8 waves = 2(m) x 4(n);
//     the two wave groups take the 16-row blocks of the tile alternately (group g owns blocks g, g+2, ...), so a wave
//     holds 16 h x 64 outputs = 16 h fp32 accumulator registers per lane; K walked in steps of 64;
//   * both operands are K-contiguous ([M][K] activations, [N][K] sampled weights) and are DMA'd straight into LDS
//     with global_load_lds_dwordx4 (no VGPR round trip), double-buffered: 2 x (256+256) rows x 128 B = 128 KiB;
//   * LDS rows are 128 B; the 16-byte chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7).  The DMA
//     destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address and again on the
//     fragment read, which makes every ds_read_b128 of an MFMA fragment bank-conflict-free;
//   * the MFMA runs with swapped operands (D rows = n, cols = m) so each lane owns 4 consecutive output features
//     of one row of y; the epilogue moves a wave's 16 h x 64 part through a private LDS slice and stores whole
//     128-byte lines (epilogue_wave below);
//   * WHICH tiles a workgroup runs is decided on the host (build_schedule below): the output is cut into columns
//     (sample, layer, n-tile) of ceil(M/32) units of 32 rows, every column into tiles of near-equal height, and the
//     tiles are dealt to the 256 persistent workgroups so that all of them carry the same number of units.  With
//     fixed 256-row tiles BERT-base's launches have 480 k tiles = 1.875 k rounds of 256 CUs — 1/16 of the CU-time is
//     a partial last round; with heights {8, 7} every CU gets exactly 15 k units.  Every XCD walks a contiguous
//     share of each height class in (sample, column group, m-band, column) order, so the W_s n-panels and the x
//     m-bands its 32 CUs re-read stay in its private 4 MiB L2 from one round to the next.
// Requirements: K % 64 == 0, 16-byte aligned operands; M and N are arbitrary (edge rows are clamped on load and
// masked on store).  Everything else goes to the generic kernel in bf_gemm.hip.
// Since round 3 the forward with 16-bit outputs and the NN input-gradient form run bf_gemm256_r5.hip (the same tile, waves,
// schedule and epilogue with the operands streamed through a five-slot LDS ring, K >= 128); this file keeps the schedule
// builder, the fp32-output forward, K = 64, operands past 2^30 elements, and the TN weight-gradient form with its unit ring.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

#include "bf_gemm256_dev.h"

#ifndef BF_NT_FORM
#define BF_NT_FORM 2
#endif
#ifndef BF_NN_FORM
#define BF_NN_FORM 1
#endif

namespace {


// ------------------------------------------------------------------------------------------------------------
// The kernel: persistent ping-pong over a host-built tile schedule.
// The two waves that share a SIMD belong to different wave groups (G0 = waves 0-3 = even 16-row blocks of the tile,
// G1 = waves 4-7 = odd blocks) and run the same slot sequence one slot apart:
//
//      slot:   4t        4t+1      4t+2      4t+3      4t+4
//      G0:     L0(t)     M0(t)     L1(t)     M1(t)     L0(t+1) ...
//      G1:     M1(t-1)   L0(t)     M0(t)     L1(t)     M1(t)   ...
//
// L = 4 + h ds_read_b128 (the fragments of one 32-deep half of the k-tile) + lgkmcnt(0); M = 4 h MFMAs on registers.
// Every slot ends in one workgroup barrier, so a SIMD always has one wave on the matrix pipe while its partner is
// on the LDS pipe.  The LDS DMA of k-step t+1 is issued by each wave at the start of its own L0(t) and is only
// waited for at the last barrier before slot 4(t+1), i.e. it has 3-4 slots (>= 1500 cycles) to land.
//   * the LDS DMA issued in the LAST k-step of a tile fetches k-step 0 of the workgroup's NEXT tile into the buffer
//     that would otherwise idle, and is retired by the k-loop's existing waits — the next tile starts without a
//     cold-start load;
//   * the epilogue is wave-private (epilogue_wave): no barrier between a tile's last k-step and the next tile's
//     first slot barrier; the next tile's first DMA (into the buffer whose slices were epilogue scratch) waits for
//     that barrier (`defer` in kstep).
// Schedule entry (int4): x = (layer, sample) pair index into w / bias / y, y = sample index into x,
// z = n-tile | height << 24 (height in 32-row units; 0 = no tile), w = first row.
// TRX / TRW = the operand is contraction-major.  Both: the TN form used by the weight-gradient GEMM of the backward pass,
// dW[n][k] = sum_m dy[m][n] x[m][k]: both operands are CONTRACTION-major in memory — p.x is [batch][K][M] (dy: rows = contraction index m, M = output rows n),
// p.w is [batch][K][N] (x) — so no transposed copies of dy and x are ever made.  A stage then holds two [64][256]
// tiles (64 contraction rows of 512 B); their 32-byte granules are XOR-swizzled by the row (again on the DMA source
// address) and the MFMA fragments come out through the LDS transpose read ds_read_b64_tr_b16: two reads give a lane
// the 8 contraction values 8 lg + 0..7 of its row, the same ones a ds_read_b128 of a K-contiguous operand delivers, so
// the two operand forms mix:
//   TRX TRW
//    0   0   NT   y = x W^T          (forward; x [M][K], W [N][K])
//    1   1   TN   dW = dy^T x        (weight gradient; dy [m][N], x [m][K], contraction over the batch rows m)
//    0   1   NN   dx = dy W          (input gradient; dy [M][N], W [N][K] read as it was sampled: no transposed copy)
//
// RING (TN form, >= 2 k-steps): the DMA of a k-step is not issued in one burst of 8 pieces per wave at L0 but as
// four UNITS of [32 contraction rows][256] (16 KiB: X0, W0 = the halves read in L0, X1, W1 = the halves read in L1),
// one unit per wave group per L slot, each into the half-buffer whose last read is one barrier behind:
//      G0:  L0(t): X1(t+1)    L1(t): X0(t+2)          G1:  L0(t): W1(t+1)    L1(t): W0(t+2)
// so a wave has at most 12 pieces in flight and never more than 4 are issued into one slot; the waits are counted
// (vmcnt(8) = "everything but the two youngest units"): G0 at the end of its M slots, G1 at the end of its L slots.
// The units of the next tile's first two k-steps are issued by the last two k-steps of a tile; the epilogue scratch
// is a separate 32 KiB region, so nothing of the next tile has to wait for the epilogue.
// Measured (dW GEMMs of the BERT-base step, one box): 940-958 -> 1067-1117 TFLOP/s.  The same ring over row-major
// operands (units = the 32-deep halves of all rows, i.e. 64-byte row segments: code below, -DBF_RING_ROWMAJOR) is
// 3-7 % SLOWER than the burst form at every K: half-line DMA requests cost more than the spread issue gains.
template <typename T, typename YT, bool TRX = false, bool TRW = false, bool SEG = false, bool RING = false>
__global__ __launch_bounds__(512, 2) void gemm256_sched_kernel(const GemmParams p) {
    using frag = typename Mfma16<T>::frag;

    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES + (RING ? 32768 : 0)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int M = p.M, N = p.N, K = p.K;


    // One tile's DMA sources: wave-uniform sample bases (SGPRs) + eight 32-bit per-lane element offsets.
    struct Src {
        const T* xb;
        const T* wb;
        const T* ob;  // RING: the base of the operand this wave's group fetches (x: group 0, w: group 1)
        unsigned xo[XPIECES], wo[4];
    };
    auto tile_setup = [&](const int4 d, Src& t, int& s, int& m0, int& n0, int& h) {
        // the lane's place in a DMA piece, recomputed per tile from an opaque copy of the lane id: as kernel-lifetime
        // values the two would sit in VGPRs through every k-loop (the fp32-output instantiations spilled two registers)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int prow = ln >> 3;
        const int kc8 = ((ln & 7) ^ ((((wid & 1) << 2) + (ln >> 4)) & 7)) * 8;
        // the entry is the same for every lane: say so, so that everything derived from it lives in SGPRs
        s = __builtin_amdgcn_readfirstlane(d.x);
        const int z = __builtin_amdgcn_readfirstlane(d.z);
        h = z >> 24;
        m0 = __builtin_amdgcn_readfirstlane(d.w);
        n0 = (z & 0xFFFFFF) * TN;
        const T* xb = reinterpret_cast<const T*>(p.x) + (long long)__builtin_amdgcn_readfirstlane(d.y) * p.x_sstride;
        const T* wb = reinterpret_cast<const T*>(p.w) + (long long)s * N * K;
        t.xb = xb;
        t.wb = wb;
        t.ob = wm == 0 ? xb : wb;
        // contraction-major operand: piece q = i * 8 + wid = contraction rows 2 q, 2 q + 1 of the k-step; lane -> (row,
        // 16-byte position); the position holds source chunk c = position ^ (key(row) << 1), key = row bits {0, 1, 3};
        // columns past the edge are clamped (they only feed output rows / columns that are masked on store).  The key
        // does not depend on i, so one offset per operand serves all four pieces (piece i = + 16 i rows, added to the
        // wave-uniform base)
        if constexpr (RING) {
            // Group 0 only ever fetches x units, group 1 w units: each wave keeps its own operand's offsets in xo[].
            // Contraction-major operand: a unit = 16 pieces of 2 rows; wave wn of the group fetches pieces i * 4 + wn = rows
            // i * 8 + rb (+ lane >> 5 inside rb).  key(row) = (rb & 3) | (i & 1) << 2.
            // Row-major operand: a unit = the 32-deep half of the k-step of all 256 rows = [256][64 B]; piece q = i * 4 + wn
            // = rows 16 q .. 16 q + 15, lane -> (row lane >> 2, 16-byte position lane & 3) holding source chunk
            // position ^ ((row >> 2) & 3) (conflict-free ds_read_b128 of a [16 rows][4 chunks] fragment block).
            auto offsets = [&](auto tr, int rows, int c0) {
                if constexpr (decltype(tr)::value) {
                    const int rb = wn * 2 + (lane >> 5);
                    // (xo[2], xo[3] repeat xo[0], xo[1]: both operand forms then index xo[] by the piece number, which
                    // keeps the struct in registers when the two groups of a kernel use different forms)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = (lane & 31) ^ (((rb & 3) | ((e & 1) << 2)) << 1);
                        t.xo[e] = (unsigned)rb * (unsigned)rows + (unsigned)min(c0 + c * 8, rows - 8);
                    }
                } else {
                    const int c = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        t.xo[i] = (unsigned)min(c0 + (i * 4 + wn) * 16 + (lane >> 2), rows - 1) * (unsigned)K + c * 8;
                }
            };
            if (wm == 0) offsets(std::integral_constant<bool, TRX>{}, M, m0);
            else offsets(std::integral_constant<bool, TRW>{}, N, n0);
            return;
        }
        const int tr_r = wid * 2 + (lane >> 5);
        const int tr_c = (lane & 31) ^ (((tr_r & 3) | ((tr_r >> 1) & 4)) << 1);
        if constexpr (TRX) {
            t.xo[0] = (unsigned)tr_r * (unsigned)M + (unsigned)min(m0 + tr_c * 8, M - 8);
        } else {
#pragma unroll
            for (int i = 0; i < XPIECES; ++i)
                t.xo[i] = (unsigned)min(m0 + (i * 8 + wid) * 8 + prow, M - 1) * (unsigned)K + kc8;
        }
        if constexpr (TRW) {
            t.wo[0] = (unsigned)tr_r * (unsigned)N + (unsigned)min(n0 + tr_c * 8, N - 8);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                t.wo[i] = (unsigned)min(n0 + (i * 8 + wid) * 8 + prow, N - 1) * (unsigned)K + kc8;
        }
    };
    // piece q = i * 8 + wid covers rows 8 q .. 8 q + 7 of the stage; only the 4 h pieces of a tile's rows are fetched
    // (inside a tile's k-loop h is the compile-time height, so a full-height tile issues its pieces without branches)
    auto stage = [&](const Src& t, int kt, int buf, auto h) {
        char* base = smem + buf * STAGE_BYTES;
        const T* xk = t.xb;
        const T* wk = t.wb;
        if constexpr (SEG) {
            // segmented contraction (NN form): k-step kt lies in segment kt / (K / TK), all wave-uniform arithmetic
            const int nks = K / TK;
            const int seg = (kt >= nks ? 1 : 0) + (kt >= 2 * nks ? 1 : 0) + (kt >= 3 * nks ? 1 : 0);
            kt -= seg * nks;
            xk += (long long)seg * p.x_seg_stride;
            wk += (long long)seg * p.w_seg_stride;
        }
        xk += TRX ? (long long)kt * TK * M : (long long)kt * TK;
        wk += TRW ? (long long)kt * TK * N : (long long)kt * TK;
        if constexpr (TRX) {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(xk + (long long)i * 16 * M + t.xo[0], base + (i * 8 + wid) * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < XPIECES; ++i)
                if (i * 8 + 7 < 4 * h || i * 8 + wid < 4 * h) glds16(xk + t.xo[i], base + (i * 8 + wid) * 1024);
        }
        if constexpr (TRW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(wk + (long long)i * 16 * N + t.wo[0], base + X_BYTES + (i * 8 + wid) * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(wk + t.wo[i], base + X_BYTES + (i * 8 + wid) * 1024);
        }
    };

    // RING: this wave's pieces (the first `cnt` of four) of its group's unit `half` (0 / 1) of k-step kt of tile t, into
    // buffer buf
    auto issue_unit = [&](const Src& t, int kt, int buf, int half, int cnt) {
#ifdef BF_DEV
        if (p.flags & 1) return;     // ablation: no DMA in the k-loop (tools/r6g_tn_probe.sh)
        if (p.flags & 64) kt = 0;    // ablation: every k-step re-reads k-step 0 (operands L2-hot)
#endif
        char* dst = smem + buf * STAGE_BYTES + (wm == 0 ? 0 : X_BYTES) + half * 16384 + wn * 1024;
        const T* b = t.ob;
        if constexpr (SEG) {
            const int nks = K / TK;
            const int seg = (kt >= nks ? 1 : 0) + (kt >= 2 * nks ? 1 : 0) + (kt >= 3 * nks ? 1 : 0);
            kt -= seg * nks;
            b += (long long)seg * (wm == 0 ? p.x_seg_stride : p.w_seg_stride);
        }
        auto go = [&](auto tr, long long ld) {
            if constexpr (decltype(tr)::value) {
                // contraction-major unit: pieces through a buffer descriptor over the sample's operand — 32-bit per-lane byte
                // offset + a scalar row offset, no 64-bit vector address arithmetic per piece (what the forward ring kernel
                // gained 3-5 % from at K = 768, bf_gemm256_r5.hip).  A sample's operand is < 2^31 bytes (host check).
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(b), 0, 0x7FFFFFFF, 0x00020000);
                const int row0 = kt * TK + half * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + i * 4096), 16, (int)(t.xo[i] * 2u),
                                                             (int)((row0 + i * 8) * ld) * 2, 0, 0);
            } else {
                const T* colp = b + kt * TK + half * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < cnt) glds16(colp + t.xo[i], dst + i * 4096);
            }
        };
        if (wm == 0) go(std::integral_constant<bool, TRX>{}, M);
        else go(std::integral_constant<bool, TRW>{}, N);
    };
    auto wait_pieces = [&](int n) {  // wave-uniform: all but this wave's n most recent pieces have landed
        switch (n) {
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    // row-major operands under RING: lane -> row lane & 15 of a 16-row block, chunk (lane >> 4) ^ ((row >> 2) & 3)

    const int fsw = (lane >> 1) & 7;
    const int foff0 = (lane & 15) * ROW_BYTES + ((((lane >> 4)) ^ fsw) << 4);
    const int foff1 = (lane & 15) * ROW_BYTES + (((4 + (lane >> 4)) ^ fsw) << 4);
    const int xfrag_base = wm * 16 * ROW_BYTES;  // + j * 32 rows: wave group wm owns blocks wm, wm + 2, ...
    const int wfrag_base = X_BYTES + wn * 64 * ROW_BYTES;
    // contraction-major tiles: the 16 lanes of a group point at a [4 contraction rows][16 columns] block: lane -> row
    // 8 lg + (li >> 2) of the 32-row half (the second read takes the 4 rows below: + 2 KiB, same key), 8 bytes (li & 3)
    // of the block's 32-byte granule; granule' = granule ^ key(row)
    const int tr_rl = ((lane & 15) >> 2) | (((lane >> 4) & 1) << 2);
    const int tr_lane = ((lane >> 4) * 8 + ((lane & 15) >> 2)) * 512 + (lane & 3) * 8;
    // LDS byte addresses of the wave's first w block (wn * 4) and first x block (wm) in buffer 0; the other blocks are
    // one XOR with a constant away (smem is 1 KiB aligned, the granule index sits alone in address bits 5..8), so the
    // k-loop holds two address registers instead of twelve
    // The reads are inline asm: after an LDS DMA hipcc waits vmcnt(0) before any LDS load it can see through the
    // builtin, which would expose the whole DMA latency in every k-step; the k-loop's own s_waitcnt lgkmcnt(0) /
    // vmcnt(0) statements order these reads against the DMA and the MFMAs.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned tr_w0 = lds0 + X_BYTES + tr_lane + (((wn * 4) ^ tr_rl) << 5);
    const unsigned tr_x0 = lds0 + tr_lane + ((wm ^ tr_rl) << 5);
    // row-major operands under RING: lane -> row lane & 15 of a 16-row block, chunk (lane >> 4) ^ ((row >> 2) & 3); the
    // blocks of a wave are immediate offsets away (x: wm + 2 j -> 2 KiB apart, w: wn * 4 + i -> 1 KiB apart)
    const unsigned ring_foff = (lane & 15) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4);
    const unsigned ring_x0 = lds0 + ring_foff + wm * 1024;
    const unsigned ring_w0 = lds0 + X_BYTES + ring_foff + wn * 4096;
    auto ring_read = [&](unsigned a, auto off) -> frag {
        frag v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(decltype(off)::value));
        return v;
    };
    auto tr_read = [&](unsigned a0, int blk_xor, auto half) -> frag {
        const unsigned a = a0 ^ (unsigned)(blk_xor << 5);
        constexpr int off = decltype(half)::value * 32 * 512;
        s16x4_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(off + 4 * 512));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(frag, v);
    };

    const int nk = SEG ? p.segs * (K / TK) : K / TK;
    const int4* __restrict__ sched = p.sched + blockIdx.x;
    const unsigned G = gridDim.x;
    int4 d = sched[0];
    if ((d.z >> 24) == 0) return;
    Src cur;
    int s, m0, n0, h;
    tile_setup(d, cur, s, m0, n0, h);
    int g = 0;  // running k-step counter: step g lives in LDS buffer g & 1
    if constexpr (RING) {
        issue_unit(cur, 0, 0, 0, 4);
        issue_unit(cur, 0, 0, 1, 4);
        issue_unit(cur, 1, 1, 0, 4);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        stage(cur, 0, 0, h);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int round = 0;
    for (;;) {
        int4 dn = {0, 0, 0, 0};
        if (round + 1 < p.sched_rounds) dn = sched[(unsigned)(round + 1) * G];
        const bool has_next = (dn.z >> 24) != 0;

        auto body = [&](auto hc) {
            constexpr int H = decltype(hc)::value;
            f32x4_t acc[4][H];
            frag wf[4], xf[H];
            // the four slots of one k-step; `dma()` issues this step's LDS DMA at the top of L0
            // `defer`: the first k-step of a tile that follows another one.  There is no barrier between the tiles: the
            // DMA of this step lands in the buffer whose slices the waves used as epilogue scratch, so group 0 issues
            // it only after the step's first barrier (every wave of both groups reaches that barrier — group 1 as
            // its start-of-tile barrier — after its epilogue); group 1's own issue point already lies behind it.
            auto kstep = [&](auto&& dma, auto last, bool defer) {
                const char* sb = smem + (g & 1) * STAGE_BYTES;
                if (!(defer && wm == 0)) dma();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (TRW) wf[i] = tr_read(tr_w0 + (g & 1) * STAGE_BYTES, i, std::integral_constant<int, 0>{});
                    else wf[i] = *reinterpret_cast<const frag*>(sb + wfrag_base + i * 16 * ROW_BYTES + foff0);
                }
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    if constexpr (TRX) xf[j] = tr_read(tr_x0 + (g & 1) * STAGE_BYTES, 2 * j, std::integral_constant<int, 0>{});
                    else xf[j] = *reinterpret_cast<const frag*>(sb + xfrag_base + j * 32 * ROW_BYTES + foff0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                if (defer && wm == 0) dma();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < H; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (TRW) wf[i] = tr_read(tr_w0 + (g & 1) * STAGE_BYTES, i, std::integral_constant<int, 1>{});
                    else wf[i] = *reinterpret_cast<const frag*>(sb + wfrag_base + i * 16 * ROW_BYTES + foff1);
                }
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    if constexpr (TRX) xf[j] = tr_read(tr_x0 + (g & 1) * STAGE_BYTES, 2 * j, std::integral_constant<int, 1>{});
                    else xf[j] = *reinterpret_cast<const frag*>(sb + xfrag_base + j * 32 * ROW_BYTES + foff1);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < H; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // (with the wave-private epilogue group 1 does not meet group 0 again before the tile-end barrier: its
                // last slot ends without one, which also keeps the two groups' barrier counts equal)
                if (!(decltype(last)::value && wm == 1)) __builtin_amdgcn_s_barrier();
                ++g;
            };

            // the RING k-step: same slots and barriers; one unit issued per L slot, counted waits.
            // A row-major x unit only needs the tile's 32 H rows: every wave of group 0 fetches the first CX of its four
            // pieces (64 CX >= 32 H rows).  Units issued across a tile boundary (and by the prologue) are always whole:
            // pieces(j, half) below is what this wave issued for that unit of k-step j (j >= nk = the next tile).
            // `steady`: kt + 2 < nk — every unit issued is this tile's own, no conditions in the loop body
            constexpr int CX = TRX ? 4 : (2 * H + 3) / 4;
            const int cw = wm == 0 ? CX : 4;
            auto pieces = [&](int j, int half) { return (j >= nk || j == 0 || (j == 1 && half == 0)) ? 4 : cw; };
            auto read_frags = [&](unsigned sboff, auto half) {
                constexpr int HF = decltype(half)::value;
                static_for<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (TRW) wf[i] = tr_read(tr_w0 + sboff, i, half);
                    else wf[i] = ring_read(ring_w0 + sboff, std::integral_constant<int, HF * 16384 + i * 1024>{});
                });
                static_for<0, H>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (TRX) xf[j] = tr_read(tr_x0 + sboff, 2 * j, half);
                    else xf[j] = ring_read(ring_x0 + sboff, std::integral_constant<int, HF * 16384 + j * 2048>{});
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            Src nxt = cur;  // the next tile's sources, set up before the last two k-steps
            auto kstep_ring = [&](int kt, auto last, auto steady) {
                constexpr bool ST = decltype(steady)::value;
                const bool e1 = ST || kt + 1 < nk || has_next, e2 = ST || kt + 2 < nk || has_next;
                const unsigned sboff = (g & 1) * STAGE_BYTES;
                if (e1) {
                    if (ST || kt + 1 < nk) issue_unit(cur, kt + 1, (g & 1) ^ 1, 1, cw);
                    else issue_unit(nxt, 0, (g & 1) ^ 1, 1, 4);
                }
                read_frags(sboff, std::integral_constant<int, 0>{});
                if (wm == 1) {  // W1(kt)
                    if constexpr (ST) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else wait_pieces(e1 ? 8 : 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < H; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                if (wm == 0) {  // X1(kt)
                    if constexpr (ST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CX) : "memory");
                    else wait_pieces(e1 ? pieces(kt + 1, 0) + pieces(kt + 1, 1) : 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                if (e2) {
                    if (ST || kt + 2 < nk) issue_unit(cur, kt + 2, g & 1, 0, cw);
                    else issue_unit(nxt, kt + 2 - nk, g & 1, 0, 4);
                }
                read_frags(sboff, std::integral_constant<int, 1>{});
                if (wm == 1) {  // W0(kt + 1)
                    if constexpr (ST) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else wait_pieces(e1 ? (e2 ? 8 : 4) : 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < H; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                if (wm == 0) {  // X0(kt + 1)
                    if constexpr (ST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CX) : "memory");
                    else wait_pieces(e1 ? pieces(kt + 1, 1) + (e2 ? pieces(kt + 2, 0) : 0) : 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(decltype(last)::value && wm == 1)) __builtin_amdgcn_s_barrier();
                ++g;
            };

            if (wm == 1) __builtin_amdgcn_s_barrier();  // G1 runs one slot behind G0
            init_acc<H>(acc, p.bias ? p.bias + (long long)s * N : nullptr, n0, N, wn, lane);

            if constexpr (RING) {
                // (k-step 0 runs the steady code too: its wait at the end of M0 then counts the whole unit X0(1) as 2 CX
                // <= 4 + CX pieces — stricter than needed, and that unit was issued before the previous epilogue)
                for (int kt = 0; kt + 2 < nk; ++kt) kstep_ring(kt, std::false_type{}, std::true_type{});
                if (has_next) {
                    int s2, m2, n2, h2;
                    tile_setup(dn, nxt, s2, m2, n2, h2);
                }
                kstep_ring(nk - 2, std::false_type{}, std::false_type{});
                kstep_ring(nk - 1, std::true_type{}, std::false_type{});
            } else {
            for (int kt = 0; kt + 1 < nk; ++kt)
                kstep([&] {
#ifdef BF_DEV
                    if (p.flags & 1) return;
                    stage(cur, (p.flags & 64) ? 0 : kt + 1, (g & 1) ^ 1, hc);
#else
                    stage(cur, kt + 1, (g & 1) ^ 1, hc);
#endif
                }, std::false_type{}, round > 0 && kt == 0);
            // last k-step: its DMA slot fetches k-step 0 of this workgroup's next tile
            kstep([&] {
#ifdef BF_DEV
                if (p.flags & 1) return;
#endif
                if (has_next) {
                    Src nxt;
                    int s2, m2, n2, h2;
                    tile_setup(dn, nxt, s2, m2, n2, h2);
                    stage(nxt, 0, (g & 1) ^ 1, h2);
                }
            }, std::true_type{}, round > 0 && nk == 1);
            }
            // the last consumed buffer is (g-1)&1; buffer g&1 already holds k-step 0 of the next tile.  Group 0 is one
            // slot ahead here and stays ahead through its epilogue
            YT* y = reinterpret_cast<YT*>(p.y) + (long long)s * M * N;
            int m_end = min(M, m0 + h * UNIT);
#ifdef BF_DEV
            if (p.flags & 16) m_end = 0;
            const bool skip = (p.flags & 8) != 0;
#else
            constexpr bool skip = false;
#endif
            YT* y2 = p.y2 ? reinterpret_cast<YT*>(p.y2) + (long long)s * M * N : nullptr;
            // every fragment read of buffer (g - 1) & 1 is complete: group 0 passed its last barrier together with the
            // end of group 1's last LDS slot, group 1 comes from its last MFMA slot
            if (!skip) {
                if constexpr (RING)
                    epilogue_wave<YT, H, sizeof(YT) == 2 ? 2 : 1>(smem + 2 * STAGE_BYTES + wid * 4096, acc, y, y2, m0, m_end, n0, N, wm, wn, lane,
                                            p.act);
                else
                    epilogue_wave<YT, H>(smem + ((g - 1) & 1) * STAGE_BYTES + wid * 8192, acc, y, y2, m0, m_end, n0, N, wm,
                                         wn, lane, p.act);
            }
        };
        switch (h) {
            case 8: body(std::integral_constant<int, 8>{}); break;
            case 7: body(std::integral_constant<int, 7>{}); break;
            case 6: body(std::integral_constant<int, 6>{}); break;
            case 5: body(std::integral_constant<int, 5>{}); break;
            default: body(std::integral_constant<int, 4>{}); break;  // h <= 4: rows past 32 h are masked on store
        }
        if (!has_next) break;
        d = dn;
        ++round;
        tile_setup(d, cur, s, m0, n0, h);
        // (no barrier between tiles: see `defer` in kstep)
    }
}

// ------------------------------------------------------------------------------------------------------------
// Host side: the tile schedule.
struct Tile {
    int pair, xs, tn, m0, h;
    long long key;  // locality order inside a height class
};

// XCD-aware bijective map of a block id to its position in the logical workgroup order (block b runs on XCD b % 8,
// observed; speed only): XCD x owns a contiguous run of logical positions.
unsigned xcd_remap(unsigned b, unsigned nwg) {
    const unsigned xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// Modelled fabric fetch of a schedule, in rows of K elements: every XCD's workgroups (block b runs on XCD b % 8, observed)
// run their j-th tiles together and in k-lockstep, so a panel shared by several of them is fetched into the XCD's L2 once
// per round — and nothing survives to the next round: the k-slices a round touches (a + b panels for a x b tiles, 12 x
// 393 KB at K = 768) exceed the 4 MiB L2 under LRU.  Validated against TCC_EA0_RDREQ on the four BERT-base launches
// (308 / 107 / 386 / 427 MB modelled, 308 / 108 / 384 / 428 MB counted: profiles/r6b_sched_l2_model.md), independent of
// the modelled L2 size between 2 and 4 MiB.  tools/sched_l2_sim.py is the same model with an explicit LRU.
long long schedule_fetch_rows(const std::vector<int4>& table, int rounds, int grid) {
    long long rows = 0;
    std::vector<long long> seen;
    for (int xcd = 0; xcd < 8; ++xcd)
        for (int j = 0; j < rounds; ++j) {
            seen.clear();
            for (int b = xcd; b < grid; b += 8) {
                const int4 d = table[(size_t)j * grid + b];
                const int h = d.z >> 24;
                if (!h) continue;
                seen.push_back(((long long)d.x << 32) | (unsigned)(d.z & 0xFFFFFF) | (1ll << 62));  // W panel (pair, n-tile)
                for (int u = 0; u < h; ++u) seen.push_back(((long long)d.y << 32) | (unsigned)(d.w / UNIT + u));  // x unit
            }
            std::sort(seen.begin(), seen.end());
            seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
            for (long long v : seen) rows += (v >> 62) ? TN : UNIT;
        }
    return rows;
}

// Cut S * layers * tiles_n columns of ceil(M / 32) units into tiles of 1..8 units and deal them to at most n_cu
// workgroups.  Returns the table ([rounds][grid] int4) and the launch grid.
// policy bit 0: workgroups at odd logical positions run their tiles in reverse order (short tiles first), which
// spreads the workgroups' epilogue store bursts over time instead of all of them ending a tile in the same
// microsecond.
// policy bit 12 (round 6): the columns that get one tile more than the others (the Bresenham remainder of the target tile
// count) are the FIRST columns instead of being spread evenly: columns of one sample then share their row cuts, so the
// tiles of a band of x rows that run together on an XCD fetch the same units (modelled fetch of the BERT-base FFN-up
// launch 3.50 -> 2.90 x its operands, BERT-large Q/K/V 4.07 -> 3.33).
// `cg` = columns per group of the locality order (policy bit 3).
void build_schedule_cg(int S, int layers, int tiles_n, int M, int n_cu, int policy, int cg, std::vector<int4>& table,
                       int& rounds, int& grid) {
    const int hmax = ((policy >> 4) & 15) ? std::min(HMAX, std::max(HMIN, (policy >> 4) & 15)) : HMAX;
    const int C = S * layers * tiles_n;
    const int Hc = (M + UNIT - 1) / UNIT;
    const long long U = (long long)C * Hc;
    const int n_min = (Hc + hmax - 1) / hmax;
    const int n_max = std::max(n_min, Hc / HMIN);
    long long T = std::max<long long>(1, (U + (long long)hmax * n_cu - 1) / ((long long)hmax * n_cu));  // tiles per CU
    std::vector<Tile> tiles;
    for (int iter = 0; iter < 4; ++iter) {
        const long long target = T * n_cu;
        tiles.clear();
        for (int c = 0; c < C; ++c) {
            // spread of the target tile count over the columns: Bresenham, or (bit 12) the remainder on the first columns
            long long n = (target * (c + 1)) / C - (target * c) / C;
            if (policy & 0x1000) n = target / C + (c < target % C ? 1 : 0);
            n = std::min<long long>(std::max<long long>(n, n_min), n_max);
            if (policy & 2) n = n_min;  // fixed full-height tiles (the round-1 decomposition)
            const int q = Hc / (int)n, r = Hc % (int)n;
            const int xs = c / (tiles_n * layers), cn = c % (tiles_n * layers);
            const int layer = cn / tiles_n, tn = cn % tiles_n;
            int u = 0;
            for (int i = 0; i < (int)n; ++i) {
                const int hh = q + (i < r ? 1 : 0);
                Tile t;
                t.pair = layer * S + xs;
                t.xs = xs;
                t.tn = tn;
                t.m0 = u * UNIT;
                t.h = hh;
                // (sample, band of 1024 rows, column, row): the 32 concurrent tiles of an XCD share few panels
                t.key = (((long long)xs * 4096 + t.m0 / 1024) * 4096 + cn) * 65536 + (t.m0 / UNIT);
                // policy bit 3: (sample, group of cg columns, 256-row band, column) — an XCD that walks this order keeps
                // cg W panels in its L2 while the x bands stream past them (groups of 3 / 4 / 6 / 9 / 12 columns measured
                // in the BERT-base step on one box, round 3: GEMM 7.13 / 7.10 / 7.20 / 7.29 / 7.22 ms, L2 fills 302 / 307 /
                // 299 / 310 / 318 MB per launch)
                if (policy & 8) t.key = ((((long long)xs * 4096 + cn / cg) * 65536 + t.m0 / 256) * 4096 + cn) * 8 + (t.m0 / UNIT) % 8;
                tiles.push_back(t);
                u += hh;
            }
        }
        if ((long long)tiles.size() <= target) break;
        T = ((long long)tiles.size() + n_cu - 1) / n_cu;  // the height cap forced more tiles than T rounds hold
    }
    const int total = (int)tiles.size();
    grid = std::min(total, n_cu);
    std::vector<std::vector<int>> lists(grid);  // per logical workgroup: indices into `tiles`, in running order
    // class-by-class dealing: tiles sorted tallest first (locality order inside a height class); round r takes the next
    // `grid` tiles.  Odd rounds are dealt backwards so that a workgroup that drew a tall tile in one round draws a
    // short one in the next — either over all workgroups, or (policy bit 4, the default) only among the 32 workgroups of
    // each XCD, which keeps an XCD on the same range of every height class (measured in the BERT-base step: 7.18 vs
    // 7.25 ms of GEMM time).
    std::stable_sort(tiles.begin(), tiles.end(), [](const Tile& a, const Tile& b) {
        return a.h != b.h ? a.h > b.h : a.key < b.key;
    });
    const bool per_xcd = (policy & 4) && grid % 8 == 0 && total >= grid;
    // (only when every height class fills whole rounds: then a workgroup that takes every 32nd tile of its XCD's list
    // gets the same number of tiles of every class; otherwise the span dealing below, which balances odd classes)
    bool whole_classes = (policy & 8) && grid % 8 == 0 && total >= grid;
    for (int a = 0; a < total && whole_classes;) {
        int b = a;
        while (b < total && tiles[b].h == tiles[a].h) ++b;
        if ((b - a) % grid) whole_classes = false;
        a = b;
    }
    if (whole_classes) {
        // Every XCD takes a CONTIGUOUS share of each height class (shares rotate so that the XCDs' tile counts stay
        // within one of each other) and its 32 workgroups walk that share 32 tiles at a time: consecutive rounds of an
        // XCD are neighbours in the locality order.  A workgroup draws every 32nd tile of its XCD's list, i.e. the same
        // number of tiles of every class: the unit balance of the class-by-class dealing is kept.
        const int span = grid / 8;
        std::vector<std::vector<int>> share(8);
        int carry = 0;
        for (int a = 0; a < total;) {
            int b = a;
            while (b < total && tiles[b].h == tiles[a].h) ++b;
            const int n = b - a;
            int start = a;
            for (int i = 0; i < 8; ++i) {
                const int x = (i + carry) % 8;
                const int cnt = (int)(((long long)n * (i + 1)) / 8 - ((long long)n * i) / 8);
                for (int k = start; k < start + cnt; ++k) share[x].push_back(k);
                start += cnt;
            }
            carry = (carry + n % 8) % 8;
            a = b;
        }
        for (int x = 0; x < 8; ++x)
            for (size_t j = 0; j < share[x].size(); ++j) {
                const int r = (int)(j / span), in = (int)(j % span);
                lists[x * span + ((r & 1) ? span - 1 - in : in)].push_back(share[x][j]);
            }
    } else if (!per_xcd) {
        for (int k = 0; k < total; ++k) {
            const int r = k / grid, pos = k % grid;
            lists[(r & 1) ? grid - 1 - pos : pos].push_back(k);
        }
    } else {
        // round r = tiles [r grid, (r+1) grid) cut into 8 spans of grid/8; XCD x takes span x of every round unless
        // swapping two XCDs' spans of some round evens out their totals (only where a height class ends inside a round)
        const int span = grid / 8, nr = (total + grid - 1) / grid;
        std::vector<std::vector<long long>> sum(nr, std::vector<long long>(8, 0));
        for (int k = 0; k < total; ++k) sum[k / grid][(k % grid) / span] += tiles[k].h;
        std::vector<std::vector<int>> perm(nr, std::vector<int>(8));
        for (int r = 0; r < nr; ++r)
            for (int x = 0; x < 8; ++x) perm[r][x] = x;
        auto tot = [&](int x) {
            long long t = 0;
            for (int r = 0; r < nr; ++r) t += sum[r][perm[r][x]];
            return t;
        };
        for (int it = 0; it < 64; ++it) {
            int hi = 0, lo = 0;
            for (int x = 1; x < 8; ++x) {
                if (tot(x) > tot(hi)) hi = x;
                if (tot(x) < tot(lo)) lo = x;
            }
            const long long th = tot(hi), tl = tot(lo);
            int best_r = -1;
            long long best = th;
            for (int r = 0; r < nr; ++r) {
                const long long d = sum[r][perm[r][hi]] - sum[r][perm[r][lo]];
                const long long m = std::max(th - d, tl + d);
                if (d > 0 && m < best) best = m, best_r = r;
            }
            if (best_r < 0) break;
            std::swap(perm[best_r][hi], perm[best_r][lo]);
        }
        for (int r = 0; r < nr; ++r)
            for (int x = 0; x < 8; ++x)
                for (int in = 0; in < span; ++in) {
                    const int k = r * grid + perm[r][x] * span + in;
                    if (k < total) lists[x * span + ((r & 1) ? span - 1 - in : in)].push_back(k);
                }
    }
    if (policy & 1)
        for (int li = 1; li < grid; li += 2) std::reverse(lists[li].begin(), lists[li].end());
    rounds = 0;
    for (const auto& l : lists) rounds = std::max(rounds, (int)l.size());
    table.assign((size_t)rounds * grid, int4{0, 0, 0, 0});
    for (int b = 0; b < grid; ++b) {
        const std::vector<int>& l = lists[xcd_remap((unsigned)b, (unsigned)grid)];
        for (size_t j = 0; j < l.size(); ++j) {
            const Tile& t = tiles[l[j]];
            table[j * grid + b] = int4{t.pair, t.xs, t.tn | (t.h << 24), t.m0};
        }
    }
}

// policy bits 8-11: columns per group of the locality order (0 = 4).  policy bit 13 (round 6): the group size is CHOSEN per
// shape — among 3, 4, 6, 8 and all columns of a sample — by the modelled fabric fetch of the resulting schedule
// (schedule_fetch_rows); the tiles, their heights and every workgroup's unit count are the same for every candidate.
void build_schedule(int S, int layers, int tiles_n, int M, int n_cu, int policy, std::vector<int4>& table, int& rounds,
                    int& grid) {
    const int cg0 = ((policy >> 8) & 15) ? (policy >> 8) & 15 : 4;
    if (!(policy & 0x2000) || !(policy & 8)) return build_schedule_cg(S, layers, tiles_n, M, n_cu, policy, cg0, table, rounds, grid);
    const int cols = tiles_n * layers;
    long long best = -1;
    for (int cg : {4, 3, 6, 8, cols}) {  // (the first candidate wins a tie: 4 is the round-3 default)
        if (cg > cols && cg != 4) continue;
        std::vector<int4> t;
        int r = 0, g = 0;
        build_schedule_cg(S, layers, tiles_n, M, n_cu, policy, cg, t, r, g);
        const long long f = schedule_fetch_rows(t, r, g);
        if (best < 0 || f < best) best = f, table.swap(t), rounds = r, grid = g;
    }
}

typedef Gemm256Sched Sched;
typedef std::tuple<int, int, int, int, int, int, int> SchedKey;  // device, S, layers, tiles_n, M, n_cu, policy
std::mutex g_sched_mu;
std::map<SchedKey, Sched> g_sched;

// The schedule of a shape is built once per device and kept in device memory for the life of the process (a few KB
// per shape).  The first launch of a shape therefore allocates: it cannot happen inside a stream capture.
int get_schedule(int S, int layers, int tiles_n, int M, int policy, hipStream_t stream, Sched& out) {
    int dev = 0;
    BF_HIP_CHECK(hipGetDevice(&dev));
    static std::map<int, int> cu_of;
    std::lock_guard<std::mutex> lk(g_sched_mu);
    int n_cu = cu_of[dev];
    if (!n_cu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
        else n_cu = prop.multiProcessorCount / 8 * 8;
        if (n_cu < 8) n_cu = 8;
        cu_of[dev] = n_cu;
    }
    const SchedKey key(dev, S, layers, tiles_n, M, n_cu, policy);
    auto it = g_sched.find(key);
    if (it != g_sched.end()) {
        out = it->second;
        return 0;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        BF_FAIL("bf_gemm_nt: the first launch of a shape (S=%d M=%d) builds its tile schedule and allocates device "
                "memory; call bf_gemm_prepare() for it, or run the step once, before capturing it into a graph", S, M);
    // Tables are NEVER freed or rewritten: a captured HIP graph keeps the raw d_table pointer in its kernel arguments
    // (bf_gemm_prepare / bench --graph), so a table must stay valid for the life of the process.  Variable-length batches
    // make every new M a new shape; a table is a few KB to a few hundred KB, so the cache is bounded by BYTES per device
    // (default 1 GiB, BF_GEMM_SCHED_CACHE_BYTES) and a shape past the bound is refused loudly instead of evicting.
    static std::map<int, size_t> bytes_of;
    static const size_t cap = [] {
        const char* e = getenv("BF_GEMM_SCHED_CACHE_BYTES");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? (size_t)v : ((size_t)1 << 30);
    }();
    std::vector<int4> table;
    Sched sc;
    build_schedule(S, layers, tiles_n, M, n_cu, policy, table, sc.rounds, sc.grid);
    if (bytes_of[dev] + table.size() * sizeof(int4) > cap)
        BF_FAIL("bf_gemm_nt: the tile schedules of this device already hold %zu bytes (%zu shapes in the process); raise "
                "BF_GEMM_SCHED_CACHE_BYTES (now %zu) or bucket the batch sizes", bytes_of[dev], g_sched.size(), cap);
    bytes_of[dev] += table.size() * sizeof(int4);
    BF_HIP_CHECK(hipMalloc((void**)&sc.d_table, table.size() * sizeof(int4)));
    BF_HIP_CHECK(hipMemcpy(sc.d_table, table.data(), table.size() * sizeof(int4), hipMemcpyHostToDevice));
    g_sched[key] = sc;
    out = sc;
    return 0;
}

template <typename T>
int launch256_tn(const GemmParams& p, hipStream_t stream, int grid) {
    // the unit ring needs two k-steps to wrap around (the contraction here is over the S * B * L batch rows: always)
    if (p.K >= 2 * TK)
        hipLaunchKernelGGL((gemm256_sched_kernel<T, float, true, true, false, true>), dim3(grid), dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm256_sched_kernel<T, float, true, true>), dim3(grid), dim3(512), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int launch256_nn(const GemmParams& p, hipStream_t stream, int grid) {
    // (the unit ring of the TN form was measured slower than the burst form for row-major operands: LABBOOK.md section 4.2)
    if (p.segs > 1) hipLaunchKernelGGL((gemm256_sched_kernel<T, T, false, true, true>), dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((gemm256_sched_kernel<T, T, false, true>), dim3(grid), dim3(512), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int launch256(const GemmParams& p, int y_dtype, hipStream_t stream, int grid) {
    if (y_dtype == BF_DT_F32)
        hipLaunchKernelGGL((gemm256_sched_kernel<T, float>), dim3(grid), dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm256_sched_kernel<T, T>), dim3(grid), dim3(512), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int bf_gemm256_get_schedule(int S, int layers, int tiles_n, int M, int policy, hipStream_t stream, Gemm256Sched& out) {
    return get_schedule(S, layers, tiles_n, M, policy, stream, out);
}

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
    if (M >= (1 << 24) || (N + TN - 1) / TN >= (1 << 24)) return false;  // schedule entry packing
    const long long tiles = (long long)((M + UNIT * HMIN - 1) / (UNIT * HMIN)) * ((N + TN - 1) / TN) * S;  // S counts (layer, sample) pairs
    if (tiles > 0x3FFFFFll) return false;  // the host-built schedule stays small
    return true;
}

extern "C" int bf_gemm_prepare(int S, int L, int M, int N, void* stream) {
    if (S < 1 || L < 1 || M < 1 || N < 1) BF_FAIL("bf_gemm_prepare: bad shape S=%d L=%d M=%d N=%d", S, L, M, N);
    Sched sc;
    return get_schedule(S, L, (N + TN - 1) / TN, M, BF_SCHED_POLICY, (hipStream_t)stream, sc);
}

extern "C" size_t bf_gemm_schedule(int S, int L, int M, int N, int n_cu, int32_t* out, size_t cap_values, int* rounds,
                                   int* grid) {
    return bf_gemm_schedule_policy(S, L, M, N, n_cu, -1, out, cap_values, rounds, grid);
}

extern "C" int64_t bf_gemm_schedule_fetch_rows(const int32_t* table, int rounds, int grid) {
    if (!table || rounds < 1 || grid < 1) return -1;
    std::vector<int4> t((size_t)rounds * grid);
    for (size_t i = 0; i < t.size(); ++i) t[i] = int4{table[4 * i], table[4 * i + 1], table[4 * i + 2], table[4 * i + 3]};
    return schedule_fetch_rows(t, rounds, grid);
}

extern "C" size_t bf_gemm_schedule_policy(int S, int L, int M, int N, int n_cu, int policy, int32_t* out, size_t cap_values,
                                          int* rounds, int* grid) {
    if (S < 1 || L < 1 || M < 1 || N < 1 || n_cu < 1) return 0;
    std::vector<int4> table;
    int r = 0, g = 0;
    build_schedule(S, L, (N + TN - 1) / TN, M, n_cu, policy < 0 ? BF_SCHED_POLICY : policy, table, r, g);
    if (rounds) *rounds = r;
    if (grid) *grid = g;
    const size_t n = table.size() * 4;
    if (out && cap_values >= n)
        for (size_t i = 0; i < table.size(); ++i) {
            out[4 * i] = table[i].x;
            out[4 * i + 1] = table[i].y;
            out[4 * i + 2] = table[i].z;
            out[4 * i + 3] = table[i].w;
        }
    return n;
}

int bf_launch_gemm256(const GemmParams& p0, int w_dtype, int y_dtype, hipStream_t stream) {
    GemmParams p = p0;
    int policy = BF_SCHED_POLICY;
#ifdef BF_DEV
    const char* ab = getenv("BF_GEMM_ABLATE");
    p.flags = ab ? atoi(ab) : 0;
    const char* pol = getenv("BF_GEMM_SCHED");
    if (pol) policy = atoi(pol);
#else
    p.flags = 0;
#endif
    if (p.layers < 1) p.layers = 1;
    p.tiles_m = (p.M + TM - 1) / TM;
    p.tiles_n = (p.N + TN - 1) / TN;
    Sched sc;
    if (get_schedule(p.S, p.layers, p.tiles_n, p.M, policy, stream, sc)) return 1;
    p.sched = sc.d_table;
    p.sched_rounds = sc.rounds;
    // forward form: the five-slot ring (bf_gemm256_r5.hip) where it applies.  BF_NT_FORM: 0 = burst kernel, non-zero = ring
    // (developer builds: BF_GEMM_NT_FORM overrides)
    int form = BF_NT_FORM;
#ifdef BF_DEV
    const char* fe = getenv("BF_GEMM_NT_FORM");
    if (fe) form = atoi(fe);
#endif
    if (form && bf_gemm256_r5_supported(p, w_dtype, y_dtype)) return bf_launch_gemm256_r5(p, w_dtype, stream, sc.grid);
    if (w_dtype == BF_DT_BF16) return launch256<__bf16>(p, y_dtype, stream, sc.grid);
    return launch256<_Float16>(p, y_dtype, stream, sc.grid);
}

// out[b][n][k] = sum_m a[b][m][n] * bmat[b][m][k]  (fp32 out): the weight-gradient GEMM dW = dy^T x of the backward pass
// without transposed copies of its operands (kernel form TR).  a: [batch][Mc][Nl], bmat: [batch][Mc][Kl], 16-bit.
bool bf_gemm256_tn_supported(int dtype, int batch, int Mc, int Nl, int Kl, const void* d_a, const void* d_b,
                             const void* d_out) {
    if (dtype != BF_DT_BF16 && dtype != BF_DT_F16) return false;
    if (batch < 1 || batch > 65535 || Mc < TK || Mc % TK || Nl < 8 || Nl % 8 || Kl < 8 || Kl % 8) return false;
    if (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_out) & 15) return false;
    // (a sample's operand is addressed by 32-bit byte offsets through a buffer descriptor: < 2^30 elements)
    if ((long long)Mc * Nl >= (1ll << 30) || (long long)Mc * Kl >= (1ll << 30) || (long long)Nl * Kl >= (1ll << 31)) return false;
    if (Nl >= (1 << 24) || (Kl + TN - 1) / TN >= (1 << 24)) return false;
    return true;
}

int bf_launch_gemm256_tn(const void* d_a, const void* d_b, float* d_out, int dtype, int batch, int Mc, int Nl, int Kl,
                         hipStream_t stream) {
    GemmParams p{};
    p.x = d_a;
    p.w = d_b;
    p.bias = nullptr;
    p.y = d_out;
    p.x_sstride = (long long)Mc * Nl;
    p.S = batch;
    p.M = Nl;
    p.N = Kl;
    p.K = Mc;
    p.act = BF_ACT_NONE;
    p.layers = 1;
    p.flags = 0;
#ifdef BF_DEV
    if (const char* ab = getenv("BF_GEMM_ABLATE")) p.flags = atoi(ab);
#endif
    p.tiles_m = (p.M + TM - 1) / TM;
    p.tiles_n = (p.N + TN - 1) / TN;
    Sched sc;
    if (get_schedule(p.S, 1, p.tiles_n, p.M, BF_SCHED_POLICY, stream, sc)) return 1;
    p.sched = sc.d_table;
    p.sched_rounds = sc.rounds;
#ifdef BF_DEV
    // developer builds, BF_GEMM_TN_FORM=1: the TN form on the five-slot ring (bf_gemm256_r5.hip, TRX) — built and measured in
    // round 6 (profiles/r6h_tn_ring5_ab.txt): bit-identical, 0-1.4 % per launch, nothing in the training step; the product
    // keeps the two-buffer unit ring of this file
    if (const char* fe = getenv("BF_GEMM_TN_FORM"))
        if (atoi(fe) && bf_gemm256_r5_tn_supported(p)) return bf_launch_gemm256_r5_tn(p, dtype, stream, sc.grid);
#endif
    if (dtype == BF_DT_BF16) return launch256_tn<__bf16>(p, stream, sc.grid);
    return launch256_tn<_Float16>(p, stream, sc.grid);
}

// y[s][m][k] = sum_n x[s][m][n] * w[s][n][k]  (16-bit in and out): the input-gradient GEMM dx = dy W of the backward pass
// with W_s [Nl][Kl] read as the sampling kernel wrote it (kernel form NN).  x: [S][M][Nl], w: [S][Nl][Kl], y: [S][M][Kl].
bool bf_gemm256_nn_supported(int dtype, int S, int M, int Nl, int Kl, const void* d_x, const void* d_w, const void* d_y) {
    if (dtype != BF_DT_BF16 && dtype != BF_DT_F16) return false;
    if (S < 1 || S > 65535 || M < 1 || Nl < TK || Nl % TK || Kl < 8 || Kl % 8) return false;
    if (((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_y) & 15) return false;
    if ((long long)M * Nl >= (1ll << 32) || (long long)Nl * Kl >= (1ll << 32) || (long long)M * Kl >= (1ll << 31)) return false;
    if (M >= (1 << 24) || (Kl + TN - 1) / TN >= (1 << 24)) return false;
    return (long long)M * Kl >= 128 * 128;
}

bool bf_gemm256_nn_actgrad_supported(int dtype, int S, int M, int Nl, int Kl, const void* d_x, const void* d_w, const void* d_y,
                                     const void* d_gpre) {
    if (!bf_gemm256_nn_supported(dtype, S, M, Nl, Kl, d_x, d_w, d_y) || !d_gpre || ((uintptr_t)d_gpre & 15)) return false;
    GemmParams p{};
    p.M = M, p.N = Kl, p.K = Nl;
    return bf_gemm256_r5_supported(p, dtype, dtype);  // the fused derivative lives in the ring kernel's epilogue only
}

int bf_launch_gemm256_nn(const void* d_x, const void* d_w, void* d_y, int dtype, int S, int M, int Nl, int Kl,
                         hipStream_t stream, int segs, const void* d_gpre, int act) {
    if (segs < 1 || segs > 4) BF_FAIL("bf_gemm_nn: 1 to 4 layers (got %d)", segs);
    if (d_gpre && (segs != 1 || act != BF_ACT_GELU || !bf_gemm256_nn_actgrad_supported(dtype, S, M, Nl, Kl, d_x, d_w, d_y, d_gpre)))
        BF_FAIL("bf_gemm_nn_actgrad: needs the ring form of the NN GEMM (16-bit, K %% 8 == 0, contraction >= 128), the GELU and one layer");
    GemmParams p{};
    p.x = d_x;
    p.w = d_w;
    p.bias = nullptr;
    p.y = d_y;
    p.y2 = nullptr;
    p.gpre = d_gpre;
    p.x_sstride = (long long)M * Nl;
    p.S = S;
    p.M = M;
    p.N = Kl;
    p.K = Nl;
    p.segs = segs;
    p.x_seg_stride = (long long)S * M * Nl;
    p.w_seg_stride = (long long)S * Nl * Kl;
    p.act = BF_ACT_NONE;
    p.layers = 1;
    p.flags = 0;
    p.tiles_m = (p.M + TM - 1) / TM;
    p.tiles_n = (p.N + TN - 1) / TN;
    Sched sc;
    if (get_schedule(p.S, 1, p.tiles_n, p.M, BF_SCHED_POLICY, stream, sc)) return 1;
    p.sched = sc.d_table;
    p.sched_rounds = sc.rounds;
    // the five-slot ring (bf_gemm256_r5.hip) where it applies; BF_NN_FORM 0 = burst kernel (developer builds: BF_GEMM_NN_FORM)
    int form = BF_NN_FORM;
#ifdef BF_DEV
    const char* fe = getenv("BF_GEMM_NN_FORM");
    if (fe) form = atoi(fe);
#endif
    if (form && bf_gemm256_r5_supported(p, dtype, dtype)) return bf_launch_gemm256_r5_nn(p, dtype, stream, sc.grid);
    if (dtype == BF_DT_BF16) return launch256_nn<__bf16>(p, stream, sc.grid);
    return launch256_nn<_Float16>(p, stream, sc.grid);
}
