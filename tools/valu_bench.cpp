// valu_bench — cycles per wave64 instruction for the VALU ops the epsilon generator leans on (gfx950).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_bench.cpp -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; f[i] = 1.0f + 0.001f * (threadIdx.x + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }       // v_mad_u64_u32
            if (OP == 1) { a[i] = __umulhi(a[i], 0xD2511F53u); }                                                       // v_mul_hi_u32
            if (OP == 2) { a[i] = a[i] * 0xD2511F53u; }                                                                // v_mul_lo_u32
            if (OP == 3) { a[i] = __umul24(a[i], 0x511F53) ^ a[i]; }                                   // v_mul_u32_u24
            if (OP == 4) { a[i] = (a[i] ^ 0x9E3779B9u) + it; }                                                        // xor+add
            if (OP == 5) { f[i] = __builtin_amdgcn_exp2f(f[i]) * 0.5f; }
            if (OP == 6) { f[i] = __builtin_amdgcn_logf(f[i]) + 2.0f; }
            if (OP == 7) { f[i] = __builtin_amdgcn_sinf(f[i]) + 1.0f; }
            if (OP == 8) { f[i] = __builtin_amdgcn_sqrtf(f[i]) + 1.0f; }
            if (OP == 9) { f[i] = fmaf(f[i], 0.999f, 0.001f); }
            if (OP == 10) { f[i] = __builtin_amdgcn_rcpf(f[i]) + 1.0f; }
            if (OP == 11) { f[i] = __builtin_ldexpf(f[i], (int)(a[i] & 1)); }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
float run(uint32_t* d, int iters, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * waves_per_simd), block(256);   // 4 waves per block -> waves_per_simd per SIMD on 256 CUs
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 256 * 8 * 256 * 4 * 2));
    const char* names[] = {"v_mad_u64_u32 (+xor)", "v_mul_hi_u32", "v_mul_lo_u32", "v_mul_u32_u24 (+xor)", "xor+add (2 ops)", "v_exp_f32 (+mul)", "v_log_f32 (+add)", "v_sin_f32 (+add)", "v_sqrt_f32 (+add)", "v_fma_f32", "v_rcp_f32 (+add)", "v_ldexp_f32 (+and)"};
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        printf("waves/SIMD = %d: ns per wave-instruction-group (8 per iteration; subtract the companion op)\n", wps);
        float t[12];
        t[0] = run<0>(d, iters, wps); t[1] = run<1>(d, iters, wps); t[2] = run<2>(d, iters, wps); t[3] = run<3>(d, iters, wps);
        t[4] = run<4>(d, iters, wps); t[5] = run<5>(d, iters, wps); t[6] = run<6>(d, iters, wps); t[7] = run<7>(d, iters, wps);
        t[8] = run<8>(d, iters, wps); t[9] = run<9>(d, iters, wps); t[10] = run<10>(d, iters, wps); t[11] = run<11>(d, iters, wps);
        for (int i = 0; i < 12; ++i)
            printf("  %-24s %.3f ms  -> %.2f ns per op-group per wave, %.2f cycles@2.4GHz per SIMD per group\n", names[i], t[i],
                   t[i] * 1e6 / (iters * 8.0), t[i] * 1e6 / (iters * 8.0 * wps) * 2.4);
    }
    return 0;
}
