// gemm_bench — developer micro-benchmark of the sampled-weight GEMM kernels through the C-ABI (bf_gemm_nt_act).
// Needs the DEVELOPER build of the library (python -m bayeformers_amd.build --dev), whose environment switches select
// the kernel per call:
//   hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -Iinclude -Lbayeformers_amd/lib -lbayeformers_amd_dev \
//         -Wl,-rpath,'$ORIGIN/../bayeformers_amd/lib' -o tools/bin/gemm_bench
//   tools/bin/gemm_bench [S M N K act] ...      (defaults: the BERT-base launches at S=10, M=4096)
// BF_BENCH_CONFIGS="1 2:0 2:1:32" lists the configurations to compare as variant[:schedule policy[:ablation bits]] (variant 0 = generic
// 128x128 kernel, 1 = round-1 fixed-tile kernel, 2 = scheduled kernel).  Every configuration is checked against
// variant 0, then all of them are timed in interleaved rounds in this one process; median and best are reported.
// A fourth field selects the forward form (BF_GEMM_NT_FORM): "2:12:0:1" = scheduled kernel, policy 12, no ablation, five-slot ring.
// BF_BENCH_FLUSH=1|2|3|4: every timed launch is timed alone, after 512 MiB were written (nothing of the problem left in the
// Infinity Cache): 1 = all cold, 2 = W read once more after the flush (W cache-hot), 3 = x hot, 4 = W and x hot.
// BF_GEMM_ABLATE bits (dev build): 1 no DMA in the k-loop, 8 no epilogue, 16 no global stores, 64 L2-hot DMA.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
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

__global__ void touch_kernel(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i].x;
    if (acc == 0x12345678u) *sink = acc;
}

struct Config {
    int variant, policy;
    std::string name;
    int ablate = 0;
    int form = -1;  // BF_GEMM_NT_FORM (0 burst, 1 five-slot ring, 2 ring with buffer loads); -1 = the library's default
};

static void select(const Config& c) {
    char v[16];
    snprintf(v, sizeof v, "%d", c.variant);
    setenv("BF_GEMM_VARIANT", v, 1);
    snprintf(v, sizeof v, "%d", c.policy);
    setenv("BF_GEMM_SCHED", v, 1);
    snprintf(v, sizeof v, "%d", c.ablate);
    setenv("BF_GEMM_ABLATE", v, 1);
    if (c.form >= 0) {
        snprintf(v, sizeof v, "%d", c.form);
        setenv("BF_GEMM_NT_FORM", v, 1);
    } else {
        unsetenv("BF_GEMM_NT_FORM");
    }
}

int main(int argc, char** argv) {
    std::vector<int> shapes;
    for (int i = 1; i < argc; ++i) shapes.push_back(atoi(argv[i]));
    // S M N K act, with L = N / 768 stacked layers when N = 2304 (the one-launch Q/K/V shape is run as a plain N)
    if (shapes.empty())
        shapes = {10, 4096, 768, 768, 0, 10, 4096, 2304, 768, 0, 10, 4096, 3072, 768, 1, 10, 4096, 768, 3072, 0,
                  1, 4096, 4096, 4096, 0};
    std::vector<Config> configs;
    const char* cs = getenv("BF_BENCH_CONFIGS");
    std::string spec = cs ? cs : "1 2:0 2:1";
    for (size_t p = 0; p < spec.size();) {
        while (p < spec.size() && spec[p] == ' ') ++p;
        if (p >= spec.size()) break;
        size_t q = spec.find(' ', p);
        if (q == std::string::npos) q = spec.size();
        std::string tok = spec.substr(p, q - p);
        Config c;
        c.variant = atoi(tok.c_str());
        size_t colon = tok.find(':');
        c.policy = colon == std::string::npos ? 1 : atoi(tok.c_str() + colon + 1);
        size_t colon2 = colon == std::string::npos ? colon : tok.find(':', colon + 1);
        c.ablate = colon2 == std::string::npos ? (getenv("BF_GEMM_ABLATE") ? atoi(getenv("BF_GEMM_ABLATE")) : 0)
                                               : atoi(tok.c_str() + colon2 + 1);
        size_t colon3 = colon2 == std::string::npos ? colon2 : tok.find(':', colon2 + 1);
        if (colon3 != std::string::npos) c.form = atoi(tok.c_str() + colon3 + 1);
        c.name = "v" + tok;
        configs.push_back(c);
        p = q;
    }
    const int rounds = getenv("BF_BENCH_ROUNDS") ? atoi(getenv("BF_BENCH_ROUNDS")) : 7;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // BF_BENCH_LAYERS=L: run every shape as L stacked layers sharing x (bf_gemm_nt_layers), N per layer
    const int L = getenv("BF_BENCH_LAYERS") ? atoi(getenv("BF_BENCH_LAYERS")) : 1;
    for (size_t q = 0; q + 4 < shapes.size(); q += 5) {
        const int S = shapes[q], M = shapes[q + 1], N = shapes[q + 2], K = shapes[q + 3], act = shapes[q + 4];
        const size_t nx = (size_t)S * M * K, nw = (size_t)L * S * N * K, ny = (size_t)L * S * M * N;
        std::vector<uint16_t> hx(nx), hw(nw);
        std::vector<float> hb((size_t)L * S * N);
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
        const double flop = 2.0 * L * S * M * (double)N * K;
        const int iters = flop > 2e11 ? 6 : 20;
        auto call = [&](void* y) {
            const int rc = L > 1 ? bf_gemm_nt_layers(dx, BF_DT_BF16, (int64_t)M * K, dw, BF_DT_BF16, db, y, BF_DT_BF16, L, S, M,
                                                     N, K, act, nullptr)
                                 : bf_gemm_nt_act(dx, BF_DT_BF16, (int64_t)M * K, dw, BF_DT_BF16, db, y, BF_DT_BF16, S, M, N, K,
                                                  act, nullptr);
            if (rc) {
                printf("bf_gemm_nt: %s\n", bf_last_error());
                exit(1);
            }
        };
        // reference: the generic kernel
        select(Config{0, 0, "v0", 0});
        call(dy0);
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> h0(ny), h1(ny), hfirst;
        CK(hipMemcpy(h0.data(), dy0, ny * 2, hipMemcpyDeviceToHost));
        printf("L=%d S=%d M=%d N=%d K=%d act=%d |", L, S, M, N, K, act);
        for (const Config& c : configs) {
            select(c);
            CK(hipMemset(dy1, 0xFF, ny * 2));
            call(dy1);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
            double maxd = 0;
            size_t bad = 0;
            for (size_t i = 0; i < ny; ++i) {
                double d = fabs((double)bf2f(h0[i]) - (double)bf2f(h1[i]));
                if (!(d <= 1e-2 * (1.0 + fabs((double)bf2f(h0[i]))))) ++bad;
                if (d > maxd || d != d) maxd = d;
            }
            // the kernel forms accumulate in the same order: every configuration must agree with the first one bit for bit
            size_t bits = 0;
            if (hfirst.empty()) hfirst = h1;
            else
                for (size_t i = 0; i < ny; ++i) bits += hfirst[i] != h1[i];
            printf(" %s maxdiff %.3g bad %zu bitdiff %zu |", c.name.c_str(), maxd, bad, bits);
        }
        printf("\n");
        std::vector<std::vector<double>> t(configs.size());
        const int flush = getenv("BF_BENCH_FLUSH") ? atoi(getenv("BF_BENCH_FLUSH")) : 0;
        if (flush) {
            static void* big = nullptr;
            static unsigned* sink = nullptr;
            if (!big) {
                CK(hipMalloc(&big, (size_t)512 << 20));
                CK(hipMalloc((void**)&sink, 4));
            }
            for (int rd = 0; rd < 3 * rounds + 1; ++rd)
                for (size_t ci = 0; ci < configs.size(); ++ci) {
                    select(configs[ci]);
                    CK(hipMemsetAsync(big, rd & 0xFF, (size_t)512 << 20, nullptr));
                    if (flush == 2 || flush == 4) hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, nullptr, (const uint4*)dw, nw * 2 / 16, sink);
                    if (flush == 3 || flush == 4) hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, nullptr, (const uint4*)dx, nx * 2 / 16, sink);
                    CK(hipEventRecord(e0, nullptr));
                    call(dy1);
                    CK(hipEventRecord(e1, nullptr));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rd) t[ci].push_back(ms);
                }
        } else
        for (int rd = 0; rd < rounds + 1; ++rd)
            for (size_t ci = 0; ci < configs.size(); ++ci) {
                select(configs[ci]);
                call(dy1);  // warm
                CK(hipEventRecord(e0, nullptr));
                for (int i = 0; i < iters; ++i) call(dy1);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rd) t[ci].push_back(ms / iters);
            }
        for (size_t ci = 0; ci < configs.size(); ++ci) {
            std::sort(t[ci].begin(), t[ci].end());
            const double med = t[ci][t[ci].size() / 2], best = t[ci][0];
            printf("    %-6s median %.1f us %.0f TF | best %.1f us %.0f TF\n", configs[ci].name.c_str(), med * 1e3,
                   flop / med / 1e9, best * 1e3, flop / best / 1e9);
        }
        hipFree(dx); hipFree(dw); hipFree(dy0); hipFree(dy1); hipFree(db);
    }
    return 0;
}
