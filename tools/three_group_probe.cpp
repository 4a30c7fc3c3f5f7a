// Probe for the next GEMM structure: THREE wave groups per workgroup (12 waves, three per SIMD, 64 x 64
// outputs per wave, <= 168 registers) rotating through [MFMA slot | DMA-issue slot | fragment-read slot], one workgroup
// barrier per slot, against the ring kernel's TWO groups (8 waves, 128 x 64 per wave) in [fragment-read (+ DMA issue) | MFMA]
// ping-pong.  With two groups the matrix pipe of a SIMD is busy 2 M / (L + M) of the time (M = the MFMA slot, L = the
// partner's load slot); a third wave per SIMD would make that min(1, 3 M / (L + M)).  The probe keeps what decides that —
// the fragment reads, the MFMAs, the barriers and (optionally) LDS-DMA pieces from an L2-resident buffer — and drops the
// rest (no tiles, no epilogue, garbage operands).
//   make -C tools bin/three_group_probe && tools/bin/three_group_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int ROW = 128;           // bytes per LDS row (64 bf16 of k)
constexpr int SLOT = 32768;        // one operand unit: 256 rows
constexpr int NSLOT = 5;

template <int OFF>
__device__ __forceinline__ void rd(bf16x8_t& v, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
}

// GROUPS = 2: H = 8 row blocks per wave (32 MFMAs per slot, 12 fragment reads), 3: H = 4 (16 MFMAs, 8 reads).
// PIECES: LDS-DMA pieces (1 KiB each) a wave issues per k-half (2 groups: in its read slot, like the ring kernel; 3 groups:
// in its own issue slot).  A k-half of a 256 x 256 (192 x 256) tile is 32 (28) pieces per workgroup: 4 (2.33) per wave.
// MODE 0: as described; 1: no barriers (every wave free-running: reads, MFMAs); 2: MFMAs only (the matrix pipe's own rate on
// these operands — the chip is power-limited, the nominal 2.5 PFLOP/s is at 2.4 GHz)
template <int GROUPS, int PIECES, int MODE = 0>
__global__ __launch_bounds__(GROUPS * 256) void probe(float* out, const char* src, int halves, int stride, int stream) {
    constexpr int H = GROUPS == 2 ? 8 : 4;
    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wid >> 2, wn = wid & 3;
    for (int i = tid; i < NSLOT * SLOT / 4; i += GROUPS * 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
    __syncthreads();
    const int fsw = (lane >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned foff[2] = {(unsigned)((lane & 15) * ROW + (((lane >> 4) ^ fsw) << 4)),
                              (unsigned)((lane & 15) * ROW + (((4 + (lane >> 4)) ^ fsw) << 4))};
    const unsigned xrow0 = lds0 + g * 16 * ROW, wrow0 = lds0 + wn * 64 * ROW;
    f32x4_t acc[4][H];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < H; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t wf[4], xf[H];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 64 << 20, 0x00020000);
    // stride 0: every workgroup re-reads its own contiguous 64 KiB (1 KiB per piece).  stride > 0: the ring kernel's pattern — a
    // piece = 8 rows x 128 B of a row-major [4096][stride bytes] operand (x: rows of band blockIdx % 16, W: of band
    // blockIdx / 16 % 16, in the second half of the buffer), the k offset fixed (stream = 0: L2-hot) or moving 128 B per k-step
    unsigned goff = (blockIdx.x * 4096u + wid * 64u + lane) * 16u;
    unsigned goff_w = goff;
    if (stride) {
        const unsigned row = wid * 8u + (lane >> 3), ch = (lane & 7) * 16u;
        goff = ((blockIdx.x & 15u) * 256u + row) * (unsigned)stride + ch;
        goff_w = (32u << 20) + (((blockIdx.x >> 4) & 15u) * 256u + row) * (unsigned)stride + ch;
    }
    const unsigned rowblk = stride ? 64u * (unsigned)stride : 16384u;

    // (inline asm: the compiler's wait-count pass would drain every in-flight LDS-DMA piece before an LDS load it can see)
    auto reads = [&](int slot_w, int slot_x, int half) {
        const unsigned aw = wrow0 + slot_w * SLOT + foff[half], ax = xrow0 + slot_x * SLOT + foff[half];
        rd<0>(wf[0], aw); rd<16 * ROW>(wf[1], aw); rd<32 * ROW>(wf[2], aw); rd<48 * ROW>(wf[3], aw);
        constexpr int XS = GROUPS * 16 * ROW;
        rd<0>(xf[0], ax); rd<XS>(xf[1], ax); rd<2 * XS>(xf[2], ax); rd<3 * XS>(xf[3], ax);
        if constexpr (H == 8) { rd<4 * XS>(xf[4], ax); rd<5 * XS>(xf[5], ax); rd<6 * XS>(xf[6], ax); rd<7 * XS>(xf[7], ax); }
    };
    auto mfmas = [&] {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < H; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto dma = [&](int slot, int n, int j) {
        const int koff = (stride && stream) ? ((j >> 1) * 128) % stride : 0;
        const unsigned base = (j & 1) ? goff_w : goff;  // x unit in one half of a k-step, W unit in the other
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + slot * SLOT + wid * 1024 + i * 8192), 16,
                                                         (int)(base + i * rowblk), koff, 0, 0);
    };

    int a = 0;
    if (MODE) {
        reads(0, 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int j = 0; j < halves; ++j) {
            if (MODE == 1) {
                reads(a, (a + 1) % NSLOT, j & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (j & 1) a = (a + 2) % NSLOT;
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (GROUPS == 2) {
        // the ring kernel's slot sequence: group 0: L M L M ..., group 1 one slot behind; L = DMA issue + reads, one barrier per slot
        if (g == 1) __builtin_amdgcn_s_barrier();
        for (int j = 0; j < halves; ++j) {
            dma((a + 3) % NSLOT, PIECES, j);
            reads(a, (a + 1) % NSLOT, j & 1);
            if (PIECES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            mfmas();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (j & 1) a = (a + 2) % NSLOT;
        }
        if (g == 0) __builtin_amdgcn_s_barrier();
    } else {
        // three groups: in slot p the group with p % 3 == g runs its MFMAs, the one that ran them in p - 1 issues its DMA
        // pieces, the third reads the fragments of its next half.  Group g starts g slots late.
        for (int k = 0; k < g; ++k) __builtin_amdgcn_s_barrier();
        reads(a, (a + 1) % NSLOT, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j < halves; ++j) {
            mfmas();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            dma((a + 3) % NSLOT, PIECES, j);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (j & 1) a = (a + 2) % NSLOT;
            reads(a, (a + 1) % NSLOT, (j + 1) & 1);
            if (PIECES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        for (int k = g; k < 2; ++k) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < H; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * GROUPS * 256 + tid] = s;
}

template <int GROUPS, int PIECES, int MODE = 0>
static void run(float* out, const char* src, int halves, const char* what, int stride = 0, int stream = 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 256;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<GROUPS, PIECES, MODE>), dim3(grid), dim3(GROUPS * 256), 0, 0, out, src, halves, stride, stream);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<GROUPS, PIECES, MODE>), dim3(grid), dim3(GROUPS * 256), 0, 0, out, src, halves, stride, stream);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const int H = GROUPS == 2 ? 8 : 4;
    const double flop = (double)grid * GROUPS * 4 * halves * (4.0 * H) * (16.0 * 16 * 32 * 2);
    printf("%-52s %8.1f us  %7.0f TFLOP/s  (%.0f %% of 2500)\n", what, best * 1e3, flop / best / 1e9, flop / best / 1e9 / 25.0);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}

int main() {
    float* out;
    char* src;
    (void)hipMalloc(&out, 256 * 768 * 4);
    (void)hipMalloc(&src, 64 << 20);
    {  // the DMA source holds what the probe's LDS image holds (zero operands would run at a higher clock)
        float* h = (float*)malloc(64 << 20);
        for (size_t i = 0; i < (64u << 20) / 4; ++i) h[i] = 0.001f * (float)(i & 127);
        (void)hipMemcpy(src, h, 64 << 20, hipMemcpyHostToDevice);
        free(h);
    }
    const int halves = 2048;
    run<2, 0, 2>(out, src, halves, "2 groups: MFMAs only");
    run<3, 0, 2>(out, src, halves, "3 groups: MFMAs only");
    run<2, 0, 1>(out, src, halves, "2 groups: reads + MFMAs, no barriers");
    run<3, 0, 1>(out, src, halves, "3 groups: reads + MFMAs, no barriers");
    run<2, 0>(out, src, halves, "2 groups (8 waves, 128x64 per wave), no DMA");
    run<2, 4>(out, src, halves, "2 groups, 4 pieces per wave and k-half (L2-hot)");
    run<2, 4>(out, src, halves, "2 groups, 4 pieces, rows of 128 B at stride 6144 (L2-hot)", 6144, 0);
    run<2, 4>(out, src, halves, "2 groups, 4 pieces, rows at stride 6144, k streaming", 6144, 1);
    run<2, 4>(out, src, halves, "2 groups, 4 pieces, rows of 128 B at stride 1536 (L2-hot)", 1536, 0);
    run<2, 4>(out, src, halves, "2 groups, 4 pieces, rows at stride 1536, k streaming", 1536, 1);
    run<3, 0>(out, src, halves, "3 groups (12 waves, 64x64 per wave), no DMA");
    run<3, 2>(out, src, halves, "3 groups, 2 pieces per wave and k-half (L2-hot)");
    run<3, 3>(out, src, halves, "3 groups, 3 pieces per wave and k-half (L2-hot)");
    return 0;
}
