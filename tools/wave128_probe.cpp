// Probe for the next GEMM design: ONE wave per SIMD, 128x128 output per wave (256 fp32 accumulators per lane, the
// unified VGPR/AGPR file), fragments read from LDS, no global traffic in the loop.  What MFMA rate does the HIP
// compiler's schedule of (16 ds_read_b128 + 64 v_mfma_f32_16x16x32_bf16) per 32-deep k-half sustain?
// Compare with the current kernel's DMA-free ablation (8 waves, 128x64 per wave): 1800-1900 TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 tools/wave128_probe.cpp -o tools/wave128_probe && ./tools/wave128_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int ROW = 128;  // bytes per LDS row (64 bf16)

template <int DOUBLE_BUFFER>
__global__ __launch_bounds__(256) void probe(float* out, int ksteps) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 256 * ROW];  // A rows 0..255 | B rows 0..255 (64 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * 256 * ROW / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
    __syncthreads();
    const int wm = wid >> 1, wn = wid & 1;  // 2 x 2 waves, 128 x 128 each
    const int fsw = (lane >> 1) & 7;
    const char* abase = smem + (wm * 128 + (lane & 15)) * ROW;
    const char* bbase = smem + 256 * ROW + (wn * 128 + (lane & 15)) * ROW;
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ksteps; ++kt) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int off = (((h * 4 + (lane >> 4)) ^ fsw) << 4);
            bf16x8_t af[8], bfr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(abase + i * 16 * ROW + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(bbase + j * 16 * ROW + off);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (DOUBLE_BUFFER) __builtin_amdgcn_s_barrier();  // stands in for the stage swap of a real kernel
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

// Same work, software-pipelined by hand: the fragments of k-half s+1 are read while the 64 MFMAs of k-half s run, and
// sched_group_barrier pins the interleave at 4 MFMAs : 1 ds_read_b128.
__global__ __launch_bounds__(256) void probe_pipelined(float* out, int ksteps) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 256 * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * 256 * ROW / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
    __syncthreads();
    const int wm = wid >> 1, wn = wid & 1;
    const int fsw = (lane >> 1) & 7;
    const char* abase = smem + (wm * 128 + (lane & 15)) * ROW;
    const char* bbase = smem + 256 * ROW + (wn * 128 + (lane & 15)) * ROW;
    const int off0 = (((0 * 4 + (lane >> 4)) ^ fsw) << 4), off1 = (((1 * 4 + (lane >> 4)) ^ fsw) << 4);
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t af[2][8], bfr[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        af[0][i] = *reinterpret_cast<const bf16x8_t*>(abase + i * 16 * ROW + off0);
        bfr[0][i] = *reinterpret_cast<const bf16x8_t*>(bbase + i * 16 * ROW + off0);
    }
    for (int s = 0; s < 2 * ksteps; s += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nxt = h ^ 1, off = nxt ? off1 : off0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                af[nxt][i] = *reinterpret_cast<const bf16x8_t*>(abase + i * 16 * ROW + off);
                bfr[nxt][i] = *reinterpret_cast<const bf16x8_t*>(bbase + i * 16 * ROW + off);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[h][j], af[h][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = t + (float)af[0][0][0];
}

// The same loop fed by LDS DMA from an L2-resident operand slab (double-buffered 64 KiB stages, one barrier per
// k-step): 16 global_load_lds_dwordx4 per wave and k-step interleaved with the MFMAs.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ __launch_bounds__(256) void probe_dma(float* out, const __bf16* src, int ksteps, int kspan) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 512 * ROW];  // 2 stages x (256 A rows + 256 B rows)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int fsw = (lane >> 1) & 7;
    const int arow = (wm * 128 + (lane & 15)) * ROW, brow = 256 * ROW + (wn * 128 + (lane & 15)) * ROW;
    const int off0 = (((0 * 4 + (lane >> 4)) ^ fsw) << 4), off1 = (((1 * 4 + (lane >> 4)) ^ fsw) << 4);
    // DMA sources: piece i of wave w covers rows (i*4 + w)*8 .. +7 of the 512-row slab, chunks swizzled like the reads
    unsigned so[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i * 4 + wid) * 8 + (lane >> 3);
        so[i] = (unsigned)row * (unsigned)kspan + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
    auto stage = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((glb_void*)(src + so[i] + (kt % (kspan / 64)) * 64),
                                         (lds_void*)(smem + buf * 512 * ROW + (i * 4 + wid) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) stage(0, 0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t af[2][8], bfr[2][8];
    for (int kt = 0; kt < ksteps; ++kt) {
        const char* sb = smem + (kt & 1) * 512 * ROW;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[0][i] = *reinterpret_cast<const bf16x8_t*>(sb + arow + i * 16 * ROW + off0);
            bfr[0][i] = *reinterpret_cast<const bf16x8_t*>(sb + brow + i * 16 * ROW + off0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    af[1][i] = *reinterpret_cast<const bf16x8_t*>(sb + arow + i * 16 * ROW + off1);
                    bfr[1][i] = *reinterpret_cast<const bf16x8_t*>(sb + brow + i * 16 * ROW + off1);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) stage(kt + 1, (kt & 1) ^ 1, h * 8 + i);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[h][j], af[h][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // 8 MFMA
                if (h == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (the DMA)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = t;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int ksteps = 2048;
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (variant) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, ksteps);
            else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, ksteps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 2.0 * 256 * 256 * 64 * ksteps;
        printf("4 waves x 128x128, LDS fragment reads + MFMA only%s: %.0f TFLOP/s (%.1f %% of 2500)\n",
               variant ? ", one barrier per k-step" : "", flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe_pipelined, dim3(256), dim3(256), 0, 0, out, ksteps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 2.0 * 256 * 256 * 64 * ksteps;
    printf("same, hand-pipelined (4 MFMA : 1 ds_read via sched_group_barrier): %.0f TFLOP/s (%.1f %% of 2500)\n",
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
    {
        const int kspan = 512;  // 512 rows x 512 k bf16 = 512 KiB slab: L2-resident
        __bf16* src;
        hipMalloc(&src, (size_t)512 * kspan * 2);
        hipMemset(src, 0x3c, (size_t)512 * kspan * 2);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe_dma, dim3(256), dim3(256), 0, 0, out, (const __bf16*)src, ksteps, kspan);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        printf("fed by LDS DMA from an L2-hot slab, double-buffered, one barrier per k-step: %.0f TFLOP/s (%.1f %% of 2500)\n",
               flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
    }
    return 0;
}
