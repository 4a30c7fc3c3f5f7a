// bf_gemm256_r5.hip — the forward form (NT: x [M][K], W_s [N][K], both K-contiguous) of the 256-wide sampled-weight GEMM
// with its LDS run as a FIVE-SLOT RING of 32 KiB operand units: y[s] = x[s] W_s^T + b_s, F.linear at
// /root/reference/bayeformers/nn/layers/linear.py:104 for all S samples in one launch.  The same body with the W unit
// contraction-major (template flag TRW) is the NN form of the backward pass, dx[s] = dy[s] W_s.
//
// Same tile, waves, fragments, swizzle, schedule and epilogue as bf_gemm256.hip (read its header first).  What differs is
// how a k-step's operands reach LDS.  There, a k-step's 64 KiB stage (x rows + W rows) is DMA'd in ONE burst of 8 pieces
// per wave at the top of the L0 slot into the other half of a double buffer: the L0 slot (8 DMA issues + 12 fragment
// reads) is about twice as long as the 32-MFMA slot it is paired with, and a piece has 3-4 slots to land.  Here all
// 160 KiB of the CU's LDS are five slots of one UNIT each — a unit = the 256 rows x 128 B of ONE operand of one k-step,
// in ring order W(0) X(0) W(1) X(1) ... — so the ring runs 1.5 k-steps ahead of the MFMAs:
//
//      slot sequence of a wave group:   L0(t)            M0(t)    L1(t)            M1(t)
//      DMA issued (4 pieces per wave):  X(t+1)                    W(t+2)
//      into the ring slot that held:    W(t-1)                    X(t-1)
//      waited for (vmcnt(4) = all but the newest unit) before the barrier that ends M1(t) (group 0) / L1(t) (group 1)
//
// Every L slot carries 4 pieces instead of 8 or 0, an x unit has 4 slots to land and a W unit 6 (W_s is the operand that
// comes cold from HBM: it is read once per step), and every piece is still 8 FULL 128-byte rows (the k-half ring of the
// burst kernel's -DBF_RING_ROWMAJOR build splits every line in two and loses).  The last two k-steps of a tile issue
// W(0), X(0), W(1) of the workgroup's next tile, so a tile boundary costs the ring nothing; the wave-private epilogue
// (16-bit outputs: 4 KiB of scratch per wave) fits in ONE of the two slots the last k-step consumed, the one whose next
// unit is issued two barriers into the next tile — no DMA ever waits for an epilogue (the burst kernel's `defer`).
// Requirements on top of bf_gemm256_supported(): K >= 128 (two k-steps: the ring wraps a tile boundary by 1.5 steps).
#include "bf_gemm256_dev.h"

namespace {

// Variants of this kernel that were built, measured in the BERT-base step and removed (LABBOOK.md section 4.2b; the sources are
// in the history up to round 5): all DMA pieces issued by one wave group (profiles/r5j_*: 4-6 % slower), the fused activation
// applied between the last k-step's MFMAs (r5e_*: 3-4 % slower), pieces issued behind the MFMA slot (r4l_*: 5 % slower), cache
// policy bits on the DMA (r4s_*), a register-only epilogue on permuted W fragments (r4i_*: 7-18 % slower), an L2 prefetch ahead of
// the DMA (r5v_*: 21 % slower), staggered workgroup starts, one barrier per k-step, no MFMA priority (r3aj_*, r3y_*).
constexpr int SLOT_BYTES = 32768;  // one unit: 256 rows x 128 B (= X_BYTES)
constexpr int NSLOT = 5;

// The pieces are fetched with buffer_load ... lds through a per-tile buffer descriptor (32-bit per-lane byte offset + scalar
// k offset: no 64-bit vector address arithmetic per piece; measured +3-5 % over global_load_lds at K = 768, whose
// instantiation also needed 9 spilled registers and is gone).
// TRW: W is contraction-major ([K][N] rows of the sampled weights as they lie: the NN form dx = dy W of the backward pass);
// a W unit is then a [64 contraction rows][256] tile of 512-byte rows whose 32-byte granules are XOR-swizzled by the row
// (on the DMA source address) and whose fragments come out through ds_read_b64_tr_b16, exactly as in bf_gemm256.hip.
// SEG: the contraction runs over p.segs segments of K (x: [segs][S][M][K], w: [segs][S][K][N]) — the one input gradient
// of the stacked query / key / value layers.
// T = float (round 5): the reference-precision form.  A unit is still 256 rows x 128 B — 32 fp32 values of k per row and
// k-step instead of 64 16-bit ones — so the ring, the DMA pieces, the swizzle and the fragment reads (16 bytes = 4
// consecutive k of a lane's row) are the 16-bit kernel's, byte for byte; a fragment feeds FOUR v_mfma_f32_16x16x4_f32
// (component c of both operands' vectors = k 4 (lane >> 4) + c of the 16-deep half: the same k in both operands, in a
// fixed order).  That MFMA runs 1/16 of the 16-bit rate (32 cycles per 2048 flop): an M slot is 4096 cycles beside an L
// slot of a few hundred, so the k-loop is bound by the matrix pipe alone.  fp32 outputs: the wave-private epilogue moves a
// 16-row block through ONE 4 KiB slice per wave (eight of them = the one consumed slot).
template <typename T>
struct FragOf { using type = typename Mfma16<T>::frag; };
template <>
struct FragOf<float> { using type = f32x4_t; };

// TRX (round 6): the x unit contraction-major too ([64 contraction rows][256] tile of 512-byte rows, swizzled and read through
// ds_read_b64_tr_b16 like a TRW unit) with fp32 outputs — the TN form dW[s] = dy[s]^T x[s] of the backward pass, which ran on
// the two-buffer unit ring of bf_gemm256.hip (1 k-step ahead, 32 KiB of LDS reserved for its epilogue) until round 6.
// AG (round 6, NN form): the epilogue multiplies the stored rows by act'(p.gpre) (bf_gemm_nn_actgrad) — an instantiation of its
// own, so that the plain input-gradient launches are the kernel they were.
template <typename T, typename YT, bool TRW = false, bool SEG = false, bool TRX = false, bool AG = false>
__global__ __launch_bounds__(512, 2) void gemm256_ring5_kernel(const GemmParams p) {
    static_assert(!AG || (TRW && !TRX && !SEG && sizeof(YT) == 2), "the activation-gradient epilogue: NN form, 16-bit outputs");
    static_assert(sizeof(T) == 2 || (!TRW && !SEG && !TRX && sizeof(YT) == 4), "fp32 operands: forward form, fp32 outputs");
    static_assert(!TRX || (TRW && !SEG), "contraction-major x: the TN form (both operands contraction-major, no segments)");
    using frag = typename FragOf<T>::type;
    constexpr unsigned ES = sizeof(T);            // bytes per operand element
    constexpr int TKE = ROW_BYTES / (int)ES;      // k-values per k-step: 64 (16-bit) or 32 (fp32)
    constexpr int CE = 16 / (int)ES;              // elements per 16-byte chunk

    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * SLOT_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int M = p.M, N = p.N, K = p.K;

    // piece q = i * 8 + wid of a unit = rows 8 q .. 8 q + 7; lane -> (row lane >> 3, 16-byte position lane & 7) holding
    // source chunk position ^ ((row >> 1) & 7)
    // (recomputed per tile from an opaque copy of the lane id — kernel-lifetime values would hold two VGPRs through every k-loop)
    auto piece_lane = [&](int& prow, int& kc8) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        prow = ln >> 3;
        kc8 = ((ln & 7) ^ ((((wid & 1) << 2) + (ln >> 4)) & 7)) * CE;
    };

    // DMA sources of the units still to be issued: wave-uniform operand bases + four per-lane BYTE offsets per operand.  Only ONE such set exists: the W half is switched to the workgroup's next tile before the k-step
    // that issues that tile's W(0), the x half one k-step later.
    const T* xb;
    const T* wb;
    unsigned xo, wo;  // per-lane byte offset of piece 0 (rows 8 wid .. 8 wid + 7); piece i lies 64 rows = `rowblk` bytes further
    const unsigned rowblk = 64u * (unsigned)K * ES;  // rows between a wave's consecutive pieces of a unit
    // An operand is fetched through a buffer descriptor that ends with the sample's operand: rows past M (N) of a partial
    // tile are out of range and arrive as zeros (their products land in rows / columns the epilogue masks) — no per-row
    // clamp, so ONE offset register per operand instead of four.
    unsigned x_bytes, w_bytes;
    auto setup_w = [&](const int4 d) {
        const int s = __builtin_amdgcn_readfirstlane(d.x);
        const int n0 = (__builtin_amdgcn_readfirstlane(d.z) & 0xFFFFFF) * TN;
        wb = reinterpret_cast<const T*>(p.w) + (long long)s * N * K;
        if constexpr (TRW) {
            // piece q = i * 8 + wid = contraction rows 2 q, 2 q + 1 of the k-step; lane -> (row, 16-byte position) holding
            // source chunk position ^ (key(row) << 1), key = row bits {0, 1, 3}; the key does not depend on i: one offset
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int tr_r = wid * 2 + (ln >> 5);
            const int tr_c = (ln & 31) ^ (((tr_r & 3) | ((tr_r >> 1) & 4)) << 1);
            wo = ((unsigned)tr_r * (unsigned)N + (unsigned)min(n0 + tr_c * 8, N - 8)) * 2u;
            w_bytes = 0x7FFFFFFF;
        } else {
            int prow, kc8;
            piece_lane(prow, kc8);
            wo = ((unsigned)(n0 + wid * 8 + prow) * (unsigned)K + kc8) * ES;
            w_bytes = (unsigned)N * (unsigned)K * ES;
        }
    };
    auto setup_x = [&](const int4 d) {
        const int m0 = __builtin_amdgcn_readfirstlane(d.w);
        xb = reinterpret_cast<const T*>(p.x) + (long long)__builtin_amdgcn_readfirstlane(d.y) * p.x_sstride;
        if constexpr (TRX) {  // as the W unit of the TRW form, over the M (= 512-byte row) direction of x
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int tr_r = wid * 2 + (ln >> 5);
            const int tr_c = (ln & 31) ^ (((tr_r & 3) | ((tr_r >> 1) & 4)) << 1);
            xo = ((unsigned)tr_r * (unsigned)M + (unsigned)min(m0 + tr_c * 8, M - 8)) * 2u;
            x_bytes = 0x7FFFFFFF;
        } else {
            int prow, kc8;
            piece_lane(prow, kc8);
            xo = ((unsigned)(m0 + wid * 8 + prow) * (unsigned)K + kc8) * ES;
            x_bytes = (unsigned)M * (unsigned)K * ES;
        }
    };
    // one 1 KiB piece: `base` (a buffer of `bytes`) + per-lane byte offset `off` + wave-uniform byte offset `soff` -> LDS `dst`
    auto piece = [&](const T* base, unsigned bytes, unsigned off, int soff, char* dst) {
#ifdef BF_DEV
        if (p.flags & 1) return;  // ablation: no DMA in the k-loop
#endif
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, (int)off, soff, 0, 0);
    };
    // segmented contraction: k-step kt lies in segment kt / (K / TK) (wave-uniform arithmetic)
    auto segment = [&](int& kt) -> int {
        if constexpr (SEG) {
            const int nks = K / TK;
            const int seg = (kt >= nks ? 1 : 0) + (kt >= 2 * nks ? 1 : 0) + (kt >= 3 * nks ? 1 : 0);
            kt -= seg * nks;
            return seg;
        }
        return 0;
    };
    // this wave's pieces of unit X(kt) / W(kt) into ring slot `slot`; only the 4 h pieces of the tile's rows of x are
    // fetched (h4 = 4 h: an integral_constant inside a tile's k-loop, so a full-height tile issues without branches)
    auto issue_x = [&](int kt, int slot, auto h4) {
#ifdef BF_DEV
        if (p.flags & 64) kt = 0;  // ablation: every k-step re-reads k-step 0 (operands L2-hot)
#endif
        char* base = smem + slot * SLOT_BYTES + wid * 1024;
        const int seg = segment(kt);
        const T* xs = SEG ? xb + (long long)seg * p.x_seg_stride : xb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (TRX) piece(xs, x_bytes, xo, (kt * TK + i * 16) * M * 2, base + i * 8192);
            else if (i * 8 + 7 < h4 || i * 8 + wid < h4) piece(xs, x_bytes, xo + i * rowblk, kt * (TK * 2), base + i * 8192);
        }
    };
    auto issue_w = [&](int kt, int slot) {
#ifdef BF_DEV
        if (p.flags & 64) kt = 0;
#endif
        char* base = smem + slot * SLOT_BYTES + wid * 1024;
        const int seg = segment(kt);
        const T* ws = SEG ? wb + (long long)seg * p.w_seg_stride : wb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (TRW) piece(ws, w_bytes, wo, (kt * TK + i * 16) * N * 2, base + i * 8192);
            else piece(ws, w_bytes, wo + i * rowblk, kt * (TK * 2), base + i * 8192);
        }
    };

    // fragment reads: inline asm (the k-loop's own waits order them against the DMA and the MFMAs; the compiler's
    // wait-count pass would otherwise drain every in-flight DMA before an LDS load it can see)
    const int fsw = (lane >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned foff0 = (lane & 15) * ROW_BYTES + ((((lane >> 4)) ^ fsw) << 4);
    const unsigned foff1 = (lane & 15) * ROW_BYTES + (((4 + (lane >> 4)) ^ fsw) << 4);
    const unsigned xrow0 = lds0 + wm * 16 * ROW_BYTES;  // + j * 32 rows: wave group wm owns blocks wm, wm + 2, ...
    const unsigned wrow0 = lds0 + wn * 64 * ROW_BYTES;  // + i * 16 rows
    auto lds_read = [&](unsigned a, auto off) -> frag {
        frag v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(decltype(off)::value));
        return v;
    };
    // contraction-major W unit: the 16 lanes of a group point at a [4 contraction rows][16 columns] block: lane -> row
    // 8 lg + (li >> 2) of the 32-row half (the second read takes the 4 rows below: + 2 KiB, same key), 8 bytes (li & 3) of
    // the block's 32-byte granule; granule' = granule ^ key(row).  Fragment block i of the wave: one XOR away.
    const int tr_rl = ((lane & 15) >> 2) | (((lane >> 4) & 1) << 2);
    const unsigned tr_w0 = lds0 + ((lane >> 4) * 8 + ((lane & 15) >> 2)) * 512 + (lane & 3) * 8 + (((wn * 4) ^ tr_rl) << 5);
    // contraction-major x unit: wave group wm owns the 16-row blocks wm, wm + 2, ... of the tile = granules wm + 2 j
    const unsigned tr_x0 = lds0 + ((lane >> 4) * 8 + ((lane & 15) >> 2)) * 512 + (lane & 3) * 8 + ((wm ^ tr_rl) << 5);
    auto tr_read = [&](unsigned a0, int blk, auto half) -> frag {
        const unsigned a = a0 ^ (unsigned)(blk << 5);
        constexpr int off = decltype(half)::value * 32 * 512;
        s16x4_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(off + 4 * 512));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(frag, v);
    };

    const int nk = SEG ? p.segs * (K / TK) : K / TKE;
    // the schedule is read through the scalar cache (it was written before the launch): entries arrive in SGPRs
    typedef int sched_i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) sched_i32x4 sched_entry;
    sched_entry* sched = (sched_entry*)(uintptr_t)(p.sched + blockIdx.x);
    auto entry = [&](unsigned i) -> int4 {
        const sched_i32x4 e = sched[i];
        return int4{e.x, e.y, e.z, e.w};
    };
    const unsigned G = gridDim.x;
    int4 d = entry(0);
    if ((d.z >> 24) == 0) return;
    int s = __builtin_amdgcn_readfirstlane(d.x);
    int h = __builtin_amdgcn_readfirstlane(d.z) >> 24;
    int m0 = __builtin_amdgcn_readfirstlane(d.w);
    int n0 = (__builtin_amdgcn_readfirstlane(d.z) & 0xFFFFFF) * TN;
    setup_w(d);
    setup_x(d);
    int a = 0;  // ring slot of W of the k-step about to run; X of that step sits in a + 1, X(+1) goes to a + 3, W(+2) to a + 4
    issue_w(0, 0);
    issue_x(0, 1, 4 * h);
    issue_w(1, 2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int round = 0;
    for (;;) {
        int4 dn = {0, 0, 0, 0};
        if (round + 1 < p.sched_rounds) dn = entry((unsigned)(round + 1) * G);
        const int h2 = __builtin_amdgcn_readfirstlane(dn.z) >> 24;
        const bool has_next = h2 != 0;

        auto body = [&](auto hc) {
            constexpr int H = decltype(hc)::value;
            f32x4_t acc[4][H];
            frag wf[4], xf[H];
            auto read_frags = [&](int slot_w, int slot_x, auto half) {
                constexpr int HF = decltype(half)::value;
                const unsigned ax = xrow0 + slot_x * SLOT_BYTES + (HF ? foff1 : foff0);
                if constexpr (TRW) {
                    const unsigned aw = tr_w0 + slot_w * SLOT_BYTES;
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = tr_read(aw, i, half);
                } else {
                    const unsigned aw = wrow0 + slot_w * SLOT_BYTES + (HF ? foff1 : foff0);
                    static_for<0, 4>([&](auto ic) {
                        wf[decltype(ic)::value] = lds_read(aw, std::integral_constant<int, decltype(ic)::value * 16 * ROW_BYTES>{});
                    });
                }
                if constexpr (TRX) {
                    const unsigned atx = tr_x0 + slot_x * SLOT_BYTES;
#pragma unroll
                    for (int j = 0; j < H; ++j) xf[j] = tr_read(atx, 2 * j, half);
                } else {
                    static_for<0, H>([&](auto jc) {
                        xf[decltype(jc)::value] = lds_read(ax, std::integral_constant<int, decltype(jc)::value * 32 * ROW_BYTES>{});
                    });
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            auto mfmas = [&] {
                __builtin_amdgcn_s_setprio(1);
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)  // 32 independent MFMAs between two on the same accumulator
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < H; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][c], xf[j][c], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < H; ++j) acc[i][j] = Mfma16<T>::run(wf[i], xf[j], acc[i][j]);
                }
                __builtin_amdgcn_s_setprio(0);
            };
            // MODE 0: kt + 2 < nk, every unit issued is this tile's own; 1: kt = nk - 2 (L1 issues the next tile's W(0));
            // 2: kt = nk - 1 (L0: the next tile's X(0), L1: its W(1)).
            auto kstep = [&](int kt, auto mode) {
                constexpr int MODE = decltype(mode)::value;
                int ax = a + 1, a3 = a + 3, a4 = a + 4;
                if (ax >= NSLOT) ax -= NSLOT;
                if (a3 >= NSLOT) a3 -= NSLOT;
                if (a4 >= NSLOT) a4 -= NSLOT;
                auto dma0 = [&] {
                    if constexpr (MODE == 2) {
                        if (has_next) issue_x(0, a3, 4 * h2);
                    } else {
                        issue_x(kt + 1, a3, std::integral_constant<int, 4 * H>{});
                    }
                };
                auto dma1 = [&] {
                    if constexpr (MODE == 0) issue_w(kt + 2, a4);
                    else if (has_next) issue_w(MODE == 1 ? 0 : 1, a4);
                };
                auto wait_all_but_newest = [&] {
                    if (MODE == 0 || has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                };
                dma0();
                read_frags(a, ax, std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                mfmas();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                dma1();
                read_frags(a, ax, std::integral_constant<int, 1>{});
                if (wm == 1) wait_all_but_newest();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                mfmas();
                if (wm == 0) wait_all_but_newest();
                __builtin_amdgcn_sched_barrier(0);
                // (group 1 does not meet group 0 again before the tile-end barrier: its last slot ends without one, which
                // also keeps the two groups' barrier counts equal)
                if (!(MODE == 2 && wm == 1)) __builtin_amdgcn_s_barrier();
                a += 2;
                if (a >= NSLOT) a -= NSLOT;
            };

            if (wm == 1) __builtin_amdgcn_s_barrier();  // G1 runs one slot behind G0
#ifdef BF_DEV
            init_acc<H>(acc, (p.bias && !(p.flags & 2)) ? p.bias + (long long)s * N : nullptr, n0, N, wn, lane);
#else
            init_acc<H>(acc, p.bias ? p.bias + (long long)s * N : nullptr, n0, N, wn, lane);
#endif

            for (int kt = 0; kt + 2 < nk; ++kt) kstep(kt, std::integral_constant<int, 0>{});
            if (has_next) setup_w(dn);  // every W unit of this tile is issued
            kstep(nk - 2, std::integral_constant<int, 1>{});
            if (has_next) setup_x(dn);  // ... and now every x unit
            kstep(nk - 1, std::integral_constant<int, 2>{});

            // epilogue scratch: 4 KiB per wave (two 16-row blocks of 16-bit outputs), all eight in the slot of X of the last
            // k-step.  Every fragment read of it is complete: group 0 passed its last barrier together with the end of
            // group 1's last LDS slot, group 1 comes from its last MFMA slot.  The next unit that lands there is W(2) of
            // the next tile, issued behind that tile's second barrier, which every wave reaches after its epilogue; the
            // slot of W of the last k-step takes the next tile's X(1) at once (no wave reads it any more).
            // (the one-slot epilogue scratch: 16-bit outputs through two 2 KiB slices per wave, fp32 outputs through one of 4 KiB)
            int sc = a + 4;
            if (sc >= NSLOT) sc -= NSLOT;
            YT* y = reinterpret_cast<YT*>(p.y) + (long long)s * M * N;
            YT* y2 = p.y2 ? reinterpret_cast<YT*>(p.y2) + (long long)s * M * N : nullptr;
            int m_end = min(M, m0 + h * UNIT);
            char* scratch = smem + sc * SLOT_BYTES + wid * 4096;
#ifdef BF_DEV
            if (p.flags & 16) m_end = 0;  // ablation: no global stores
            if (p.flags & 8) return;      // ablation: no epilogue
#endif
            const YT* gpre = AG ? reinterpret_cast<const YT*>(p.gpre) + (long long)s * M * N : nullptr;
            epilogue_wave<YT, H, sizeof(YT) == 2 ? 2 : 1>(scratch, acc, y, y2, m0, m_end, n0, N, wm, wn, lane, p.act, gpre);
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
        s = __builtin_amdgcn_readfirstlane(d.x);
        h = h2;
        m0 = __builtin_amdgcn_readfirstlane(d.w);
        n0 = (__builtin_amdgcn_readfirstlane(d.z) & 0xFFFFFF) * TN;
        // (no barrier between tiles: see `defer` in kstep)
    }
}

template <typename T, typename YT, bool TRW, bool SEG, bool TRX = false, bool AG = false>
int launch_r5(const GemmParams& p, hipStream_t stream, int grid) {
    hipLaunchKernelGGL((gemm256_ring5_kernel<T, YT, TRW, SEG, TRX, AG>), dim3(grid), dim3(512), 0, stream, p);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

bool bf_gemm256_r5_supported(const GemmParams& p, int w_dtype, int y_dtype) {
    if (p.K < 2 * TK) return false;
    if (y_dtype != w_dtype) return false;  // fp32 outputs stay on the burst kernel
    // an operand of one sample is addressed by 32-bit byte offsets
    // (rows of a partial last tile are addressed past M / N before the buffer's range check drops them: + one tile)
    if ((long long)(p.M + 256) * p.K >= (1ll << 30) || (long long)(p.N + 256) * p.K >= (1ll << 30)) return false;
    return true;
}

int bf_launch_gemm256_r5(const GemmParams& p, int w_dtype, hipStream_t stream, int grid) {
    if (p.segs > 1) BF_FAIL("bf_gemm256_r5: the forward form has no segments");
    if (w_dtype == BF_DT_BF16) return launch_r5<__bf16, __bf16, false, false>(p, stream, grid);
    return launch_r5<_Float16, _Float16, false, false>(p, stream, grid);
}

// fp32 operands and outputs (the reference's own precision, /root/reference/bayeformers/nn/parameters/base.py:32,
// layers/linear.py:104): same schedule, same ring, v_mfma_f32_16x16x4_f32.
bool bf_gemm256_f32_supported(int S, int M, int N, int K, const void* d_x, const void* d_w, const void* d_y,
                              const float* d_bias, int64_t x_sample_stride) {
    if (K % 32 != 0 || K < 64) return false;   // whole 128-byte k-steps, two of them at least
    if (N % 4 != 0) return false;              // 16-byte bias rows / output chunks
    if (((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_y | (uintptr_t)d_bias) & 15) return false;
    if (((size_t)x_sample_stride * 4) % 16 != 0) return false;
    // an operand of one sample is addressed by 32-bit byte offsets inside a buffer descriptor of < 2^31 bytes (+ one tile)
    if ((long long)(M + 256) * K >= (1ll << 28) || (long long)(N + 256) * K >= (1ll << 28) || (long long)M * N >= (1ll << 31)) return false;
    if (M >= (1 << 24) || (N + TN - 1) / TN >= (1 << 24)) return false;  // schedule entry packing
    const long long tiles = (long long)((M + UNIT * HMIN - 1) / (UNIT * HMIN)) * ((N + TN - 1) / TN) * S;
    return tiles <= 0x3FFFFFll;
}

int bf_launch_gemm256_f32(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    p.flags = 0;
    if (p.layers < 1) p.layers = 1;
    p.tiles_m = (p.M + TM - 1) / TM;
    p.tiles_n = (p.N + TN - 1) / TN;
    Gemm256Sched sc;
    if (bf_gemm256_get_schedule(p.S, p.layers, p.tiles_n, p.M, BF_SCHED_POLICY, stream, sc)) return 1;
    p.sched = sc.d_table;
    p.sched_rounds = sc.rounds;
    return launch_r5<float, float, false, false>(p, stream, sc.grid);
}

#ifdef BF_DEV
// TN form (both operands contraction-major, fp32 out): out[b][n][k] = sum_m a[b][m][n] * bmat[b][m][k] — the weight gradient.
// Developer builds only: measured against the unit ring of bf_gemm256.hip (profiles/r6h_tn_ring5_ab.txt), not faster.
bool bf_gemm256_r5_tn_supported(const GemmParams& p) {
    if (p.K < 2 * TK || p.M % 8 || p.N % 8) return false;
    // (a sample's operand is addressed by 32-bit byte offsets through a buffer descriptor)
    return (long long)(p.K + 64) * p.M < (1ll << 30) && (long long)(p.K + 64) * p.N < (1ll << 30);
}

int bf_launch_gemm256_r5_tn(const GemmParams& p, int dtype, hipStream_t stream, int grid) {
    if (dtype == BF_DT_BF16) return launch_r5<__bf16, float, true, false, true>(p, stream, grid);
    return launch_r5<_Float16, float, true, false, true>(p, stream, grid);
}
#endif

// NN form (x K-contiguous, W contraction-major as sampled, 16-bit out): y[s][m][k] = sum_n x[s][m][n] w[s][n][k]
int bf_launch_gemm256_r5_nn(const GemmParams& p, int dtype, hipStream_t stream, int grid) {
    const bool seg = p.segs > 1;
    if (p.gpre) {  // the activation-gradient epilogue (one layer, no segments: checked by the caller)
        if (dtype == BF_DT_BF16) return launch_r5<__bf16, __bf16, true, false, false, true>(p, stream, grid);
        return launch_r5<_Float16, _Float16, true, false, false, true>(p, stream, grid);
    }
    if (dtype == BF_DT_BF16)
        return seg ? launch_r5<__bf16, __bf16, true, true>(p, stream, grid) : launch_r5<__bf16, __bf16, true, false>(p, stream, grid);
    return seg ? launch_r5<_Float16, _Float16, true, true>(p, stream, grid) : launch_r5<_Float16, _Float16, true, false>(p, stream, grid);
}
