// gemm_bench — developer micro-benchmark of bf_gemm_nt variants (BF_GEMM_VARIANT env knob) through the C-ABI.
//   hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -Iinclude -Lbayeformers_amd/lib -lbayeformers_amd \
//         -Wl,-rpath,'$ORIGIN/../bayeformers_amd/lib' -o tools/gemm_bench
//   tools/gemm_bench [S M N K] ...        (defaults: the BERT-base shapes at S=10, M=4096)
// For each shape: checks the 256x256x64 persistent kernel (variant 1) against variant 0 (the generic 128x128x32
// kernel), then times both.  BF_GEMM_ABLATE bits: 1 no DMA in the k-loop, 8 no epilogue, 16 no global stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bayeformers_amd.h"

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e));                            \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static double run(int variant, const void* x, const void* w, const float* b, void* y, int S, int M, int N, int K, int iters) {
    char v[8];
    snprintf(v, sizeof v, "%d", variant);
    setenv("BF_GEMM_VARIANT", v, 1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (bf_gemm_nt(x, BF_DT_BF16, (int64_t)M * K, w, BF_DT_BF16, b, y, BF_DT_BF16, S, M, N, K, nullptr)) {
            printf("bf_gemm_nt: %s\n", bf_last_error());
            exit(1);
        }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) bf_gemm_nt(x, BF_DT_BF16, (int64_t)M * K, w, BF_DT_BF16, b, y, BF_DT_BF16, S, M, N, K, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    std::vector<int> shapes;
    for (int i = 1; i < argc; ++i) shapes.push_back(atoi(argv[i]));
    if (shapes.empty()) shapes = {10, 4096, 768, 768, 10, 4096, 3072, 768, 10, 4096, 768, 3072, 10, 4096, 2304, 768,
                                  1, 4096, 4096, 4096};
    const char* vs = getenv("BF_BENCH_VARIANTS");
    std::vector<int> variants = {1};
    if (vs) {
        variants.clear();
        for (const char* p = vs; *p; ++p)
            if (*p >= '0' && *p <= '9') variants.push_back(*p - '0');
    }
    for (size_t q = 0; q + 3 < shapes.size(); q += 4) {
        const int S = shapes[q], M = shapes[q + 1], N = shapes[q + 2], K = shapes[q + 3];
        const size_t nx = (size_t)S * M * K, nw = (size_t)S * N * K, ny = (size_t)S * M * N;
        std::vector<uint16_t> hx(nx), hw(nw);
        std::vector<float> hb((size_t)S * N);
        uint32_t r = 12345;
        auto rnd = [&]() {
            r = r * 1664525u + 1013904223u;
            return ((r >> 8) & 0xFFFF) / 32768.0f - 1.0f;
        };
        for (auto& v : hx) v = f2bf(rnd());
        for (auto& v : hw) v = f2bf(rnd() * 0.05f);
        for (auto& v : hb) v = rnd();
        void *dx, *dw, *dy0, *dy1;
        float* db;
        CK(hipMalloc(&dx, nx * 2));
        CK(hipMalloc(&dw, nw * 2));
        CK(hipMalloc(&dy0, ny * 2));
        CK(hipMalloc(&dy1, ny * 2));
        CK(hipMalloc((void**)&db, hb.size() * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        const double flop = 2.0 * S * M * (double)N * K;
        const int iters = flop > 2e11 ? 10 : 30;
        double t0 = run(0, dx, dw, db, dy0, S, M, N, K, iters);
        printf("S=%d M=%d N=%d K=%d | v0 %.1f us %.0f TF", S, M, N, K, t0 * 1e3, flop / t0 / 1e9);
        std::vector<uint16_t> h0(ny), h1(ny);
        CK(hipMemcpy(h0.data(), dy0, ny * 2, hipMemcpyDeviceToHost));
        for (int v : variants) {
            CK(hipMemset(dy1, 0xFF, ny * 2));
            double t1 = run(v, dx, dw, db, dy1, S, M, N, K, iters);
            CK(hipMemcpy(h1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
            double maxd = 0;
            size_t bad = 0;
            for (size_t i = 0; i < ny; ++i) {
                double d = fabs((double)bf2f(h0[i]) - (double)bf2f(h1[i]));
                if (!(d <= 1e-2 * (1.0 + fabs((double)bf2f(h0[i]))))) ++bad;
                if (d > maxd || d != d) maxd = d;
            }
            printf(" | v%d %.1f us %.0f TF maxdiff %.3g bad %zu", v, t1 * 1e3, flop / t1 / 1e9, maxd, bad);
        }
        printf("\n");
        hipFree(dx); hipFree(dw); hipFree(dy0); hipFree(dy1); hipFree(db);
    }
    return 0;
}
