// How fast can a CU push 128 KiB tiles of finished output into memory?  (the GEMM epilogue's store pattern, alone)
//   hipcc -O3 --offload-arch=gfx950 tools/store_bench.cpp -o tools/store_bench && ./tools/store_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// each workgroup (512 threads) writes `tiles` tiles of 256 rows x 512 B; a wave instruction = 2 whole rows (1 KiB)
template <bool NT>
__global__ __launch_bounds__(512) void store_kernel(char* out, int tiles, long long row_stride, long long tile_stride) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    char* base = out + (long long)blockIdx.x * tiles * tile_stride;
    f32x4_t v = {(float)tid, 1.f, 2.f, 3.f};
    for (int t = 0; t < tiles; ++t) {
        char* tb = base + (long long)t * tile_stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i * 8 + wid) * 2 + (lane >> 5);
            f32x4_t* p = reinterpret_cast<f32x4_t*>(tb + row * row_stride + (lane & 31) * 16);
            if (NT) __builtin_nontemporal_store(v, p);
            else *p = v;
        }
        v[1] += 1.f;
    }
}

int main() {
    const int tiles = 64;
    const long long tile_bytes = 256 * 512;
    char* buf;
    hipMalloc(&buf, (size_t)256 * tiles * tile_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int grid : {8, 32, 64, 128, 256}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(store_kernel<true>, dim3(grid), dim3(512), 0, 0, buf, tiles, 512LL, tile_bytes);
                else hipLaunchKernelGGL(store_kernel<false>, dim3(grid), dim3(512), 0, 0, buf, tiles, 512LL, tile_bytes);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us_per_tile = ms * 1e3 / tiles;
            printf("%s stores, %3d workgroups: %.2f us per 128 KiB tile per CU, %.0f GB/s total\n", nt ? "nontemporal" : "plain      ",
                   grid, us_per_tile, (double)grid * tiles * tile_bytes / (ms * 1e-3) / 1e9);
        }
    return 0;
}
