// bf_api.hip — the extern "C" surface declared in include/bayeformers_amd.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bf_common.h"
#ifdef BF_DEV
#include "bf_dev_api.h"
#endif
#include "bf_gemm_params.h"
#include "bf_philox.h"

static thread_local char g_err[512] = "";
// device-resident sample counters, one slot per HIP device: a pointer is only meaningful on the device it was
// allocated on, and launches look it up by the calling thread's current device
static constexpr int kMaxDevices = 64;
static const uint32_t* g_sample_counter[kMaxDevices] = {};

static int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    return dev;
}

const uint32_t* bf_sample_counter() { return g_sample_counter[current_device_slot()]; }

void bf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------- profiling
#include <mutex>
#include <vector>
namespace {
struct ProfEntry {
    int kind;
    double work;
    hipEvent_t start, stop;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfEntry> g_prof;            // recorded launches
std::vector<hipEvent_t> g_prof_free;      // event pool

hipEvent_t prof_event() {
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    bool on;
    ProfEntry e;
    hipStream_t stream;
    ProfScope(int kind, double work, hipStream_t s) : on(g_prof_on), stream(s) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        e.kind = kind;
        e.work = work;
        e.start = prof_event();
        e.stop = prof_event();
        (void)hipEventRecord(e.start, stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(e.stop, stream);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(e);
    }
};

double sample_work_bytes(const bf_tensor_t* t, int n, int S) {
    double b = 0;
    for (int i = 0; i < n; ++i) {
        const double per = (t[i].prior.kind == BF_PRIOR_GAUSSIAN ? 16.0 : 8.0) +
                           (t[i].d_sample_out ? (double)S * (double)bf_dtype_size(t[i].out_dtype) : 0.0);
        b += per * (double)t[i].n;
    }
    return b;
}
}  // namespace

__global__ __launch_bounds__(256) void bf_probe_read_kernel(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane
    for (; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n; i += stride) acc ^= p[i].x;
    if (acc == 0x9E3779B9u) *sink = acc;  // never true for the buffers bench.py fills; keeps the loads alive
}

// the stale-prior counter: one pinned, device-mapped word for the process (bf_stale_counter)
static uint32_t* g_stale_host = nullptr;
static uint32_t* g_stale_dev = nullptr;
static std::once_flag g_stale_once;
uint32_t* bf_stale_counter_dev() {
    std::call_once(g_stale_once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return; }
        *reinterpret_cast<uint32_t*>(h) = 0;
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_stale_host = reinterpret_cast<uint32_t*>(h);
        g_stale_dev = reinterpret_cast<uint32_t*>(d);
    });
    return g_stale_dev;
}

extern "C" {

int bf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

int bf_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        g_prof_free.push_back(e.start);
        g_prof_free.push_back(e.stop);
    }
    g_prof.clear();
    return 0;
}

int bf_profile_read(int kind, uint64_t* launches, double* total_ms, double* total_work) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    uint64_t n = 0;
    double ms = 0, work = 0;
    for (auto& e : g_prof) {
        if (e.kind != kind) continue;
        BF_HIP_CHECK(hipEventSynchronize(e.stop));
        float t = 0.f;
        BF_HIP_CHECK(hipEventElapsedTime(&t, e.start, e.stop));
        ms += t;
        work += e.work;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    return 0;
}

size_t bf_profile_read_launches(int kind, float* ms, double* work, size_t cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    size_t n = 0;
    for (auto& e : g_prof) {
        if (e.kind != kind) continue;
        if (n < cap && (ms || work)) {
            float t = 0.f;
            if (hipEventSynchronize(e.stop) != hipSuccess || hipEventElapsedTime(&t, e.start, e.stop) != hipSuccess) t = -1.f;
            if (ms) ms[n] = t;
            if (work) work[n] = e.work;
        }
        ++n;
    }
    return n;
}

int bf_version(void) { return BF_VERSION_MAJOR * 1000 + BF_VERSION_MINOR; }

int bf_stale_counter(const uint32_t** h_counter) {
    if (!h_counter) BF_FAIL("bf_stale_counter: h_counter is NULL");
    if (!bf_stale_counter_dev()) BF_FAIL("bf_stale_counter: the pinned counter could not be allocated");
    *h_counter = g_stale_host;
    return 0;
}

// Measurement utility (bench.py's traffic leg): one streaming pass of 16-byte loads over `bytes` of device memory, the
// access shape of the GEMM's LDS-DMA pieces and of the sampling kernel's parameter reads.  Run under a PMC pass it
// calibrates the L2's fabric-side counters on a known byte count and on data of a known home: a buffer far larger than
// the 256 MiB Infinity Cache comes from HBM, a 96 MiB buffer read again comes from the cache.
int bf_probe_stream_read(const void* d_buf, size_t bytes, void* d_sink, void* stream) {
    if (!d_buf || !d_sink || bytes < 16) BF_FAIL("bf_probe_stream_read: NULL buffer / sink or fewer than 16 bytes");
    if ((uintptr_t)d_buf & 15) BF_FAIL("bf_probe_stream_read: the buffer must be 16-byte aligned");
    hipLaunchKernelGGL(bf_probe_read_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_buf, bytes / 16, (unsigned*)d_sink);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_set_sample_counter(const uint32_t* d_counter) {
    g_sample_counter[current_device_slot()] = d_counter;
    return 0;
}

const uint32_t* bf_get_sample_counter(void) { return g_sample_counter[current_device_slot()]; }

const char* bf_last_error(void) { return g_err; }

int bf_device_info(char* name, size_t name_len, int* n_cu, int* wave_size) {
    int dev = 0;
    BF_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    BF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    return 0;
}

int bf_philox_normal_host(float* out, uint64_t n, uint64_t seed, uint32_t sample, uint32_t stream_id,
                          uint64_t offset) {
    if (!out && n) BF_FAIL("bf_philox_normal_host: out is NULL");
    uint64_t i = 0;
    while (i < n) {
        const uint64_t e = offset + i;
        float z[4];
        bf_normal4_host(e >> 2, sample, stream_id, seed, z);
        for (uint64_t j = e & 3; j < 4 && i < n; ++j, ++i) out[i] = z[j];
    }
    return 0;
}

int bf_philox_normal(float* d_out, uint64_t n, int S, uint64_t seed, uint32_t sample_base, uint32_t stream_id,
                     void* stream) {
    if (!d_out && n) BF_FAIL("bf_philox_normal: d_out is NULL");
    return bf_launch_philox_normal(d_out, n, S, seed, sample_base, stream_id, (hipStream_t)stream);
}

size_t bf_sample_logprob_workspace_bytes(const bf_tensor_t* tensors, int n_tensors, int S) {
    if (!tensors || n_tensors < 1 || S < 1) return 0;
    return bf_sample_partials_bytes(tensors, n_tensors, S);
}

int bf_sample_logprob(const bf_tensor_t* tensors, int n_tensors, int S, uint64_t seed, uint32_t sample_base,
                      double* d_logprob_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!tensors) BF_FAIL("bf_sample_logprob: tensors is NULL");
    ProfScope prof(BF_PROF_SAMPLE, n_tensors >= 1 && n_tensors <= 2 ? sample_work_bytes(tensors, n_tensors, S) : 0.0,
                   (hipStream_t)stream);
    return bf_launch_sample_logprob(tensors, n_tensors, S, seed, sample_base, d_logprob_out, d_workspace,
                                    workspace_bytes, (hipStream_t)stream);
}

size_t bf_sample_table_bytes(const bf_tensor_t* tensors, int n_tensors, uint32_t* total_blocks) {
    if (!tensors || n_tensors < 1) return 0;
    return bf_table_blob_bytes(tensors, n_tensors, total_blocks);
}

int bf_sample_table_build(const bf_tensor_t* tensors, int n_tensors, void* h_blob, size_t blob_bytes,
                          uint32_t* h_block_begin, int32_t* h_kinds) {
    if (!tensors || n_tensors < 1) BF_FAIL("bf_sample_table_build: no tensors");
    return bf_table_build(tensors, n_tensors, h_blob, blob_bytes, h_block_begin, h_kinds);
}

int bf_sample_logprob_table(const void* d_blob, int n_tensors, uint32_t block_begin, uint32_t block_end, int S,
                            uint64_t seed, uint32_t sample_base, double* d_partials, int prior_kinds, void* stream) {
    ProfScope prof(BF_PROF_SAMPLE, 0.0, (hipStream_t)stream);
    return bf_launch_sample_table(d_blob, n_tensors, block_begin, block_end, S, seed, sample_base, d_partials,
                                  (hipStream_t)stream, prior_kinds);
}

int bf_reduce_logprob(const double* d_partials, const uint32_t* d_rows, int n_groups, int S, double* d_out,
                      void* stream) {
    return bf_launch_reduce_groups(d_partials, d_rows, n_groups, S, d_out, (hipStream_t)stream);
}

int bf_gemm_nt(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
               const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, void* stream) {
    ProfScope prof(BF_PROF_GEMM, 2.0 * S * M * (double)N * K, (hipStream_t)stream);
    return bf_launch_gemm_nt(d_x, x_dtype, x_sample_stride, d_w, w_dtype, d_bias, d_y, y_dtype, S, M, N, K,
                             (hipStream_t)stream);
}

int bf_gemm_nt_act(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                   const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, int act, void* stream) {
    if (act != BF_ACT_NONE && act != BF_ACT_GELU) BF_FAIL("bf_gemm_nt_act: unknown activation %d", act);
    ProfScope prof(BF_PROF_GEMM, 2.0 * S * M * (double)N * K, (hipStream_t)stream);
    return bf_launch_gemm_nt(d_x, x_dtype, x_sample_stride, d_w, w_dtype, d_bias, d_y, y_dtype, S, M, N, K,
                             (hipStream_t)stream, act);
}

int bf_gemm_nt_act_pre(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                       const float* d_bias, void* d_y, void* d_pre, int y_dtype, int S, int M, int N, int K, int act,
                       void* stream) {
    if (act != BF_ACT_NONE && act != BF_ACT_GELU) BF_FAIL("bf_gemm_nt_act_pre: unknown activation %d", act);
    if (!d_pre) BF_FAIL("bf_gemm_nt_act_pre: d_pre is NULL");
    ProfScope prof(BF_PROF_GEMM, 2.0 * S * M * (double)N * K, (hipStream_t)stream);
    return bf_launch_gemm_nt(d_x, x_dtype, x_sample_stride, d_w, w_dtype, d_bias, d_y, y_dtype, S, M, N, K,
                             (hipStream_t)stream, act, 1, d_pre);
}

int bf_gemm_nt_layers(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                      const float* d_bias, void* d_y, int y_dtype, int L, int S, int M, int N, int K, int act,
                      void* stream) {
    if (act != BF_ACT_NONE && act != BF_ACT_GELU) BF_FAIL("bf_gemm_nt_layers: unknown activation %d", act);
    if (L < 1) BF_FAIL("bf_gemm_nt_layers: L must be >= 1 (got %d)", L);
    ProfScope prof(BF_PROF_GEMM, 2.0 * L * S * M * (double)N * K, (hipStream_t)stream);
    return bf_launch_gemm_nt(d_x, x_dtype, x_sample_stride, d_w, w_dtype, d_bias, d_y, y_dtype, S, M, N, K,
                             (hipStream_t)stream, act, L);
}

// workspace layout of bf_linear_fwd: [W_s : S*N*K compute_dtype][b_s : S*N fp32][log-prob partials]
static void linear_ws_layout(int S, int N, int K, int has_bias, int compute_dtype, size_t* off_w, size_t* off_b,
                             size_t* off_p, size_t* total) {
    const size_t wbytes = bf_align_up((size_t)S * N * K * bf_dtype_size(compute_dtype), 256);
    const size_t bbytes = has_bias ? bf_align_up((size_t)S * N * sizeof(float), 256) : 0;
    bf_tensor_t t[2];
    memset(t, 0, sizeof(t));
    t[0].n = (uint64_t)N * K;
    t[1].n = (uint64_t)N;
    size_t pbytes = bf_sample_partials_bytes(t, has_bias ? 2 : 1, S);
    const size_t fbytes = bf_align_up(bf_fused_small_partial_rows(N) * (size_t)S * 2 * sizeof(double), 256);
    if (fbytes > pbytes) pbytes = fbytes;  // the single-kernel small-M path keeps one partial row per 16 features
    *off_w = 0;
    *off_b = wbytes;
    *off_p = wbytes + bbytes;
    *total = wbytes + bbytes + pbytes;
}

size_t bf_linear_fwd_workspace_bytes(int S, int M, int N, int K, int has_bias, int compute_dtype, int x_dtype) {
    (void)M;
    (void)x_dtype;
    if (S < 1 || N < 1 || K < 1) return 0;
    size_t ow, ob, op, total;
    linear_ws_layout(S, N, K, has_bias, compute_dtype, &ow, &ob, &op, &total);
    return total;
}

int bf_linear_fwd(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                  const bf_tensor_t* bias, void* d_y, int y_dtype, int compute_dtype, int S, int M, int N, int K,
                  uint64_t seed, uint32_t sample_base, double* d_logprob_out, void* d_workspace,
                  size_t workspace_bytes, void* stream) {
    if (!weight) BF_FAIL("bf_linear_fwd: weight is NULL");
    if (S < 1 || M < 1 || N < 1 || K < 1) BF_FAIL("bf_linear_fwd: bad shape S=%d M=%d N=%d K=%d", S, M, N, K);
    if (weight->n != (uint64_t)N * (uint64_t)K)
        BF_FAIL("bf_linear_fwd: weight.n=%llu != N*K=%llu", (unsigned long long)weight->n,
                (unsigned long long)N * (unsigned long long)K);
    if (bias && bias->n != (uint64_t)N) BF_FAIL("bf_linear_fwd: bias.n=%llu != N=%d", (unsigned long long)bias->n, N);
    if (compute_dtype < BF_DT_F32 || compute_dtype > BF_DT_F16) BF_FAIL("bf_linear_fwd: bad compute dtype %d", compute_dtype);
    size_t ow, ob, op, total;
    linear_ws_layout(S, N, K, bias != nullptr, compute_dtype, &ow, &ob, &op, &total);
    if (!d_workspace || workspace_bytes < total)
        BF_FAIL("bf_linear_fwd: workspace too small (%zu < %zu bytes)", workspace_bytes, total);
    char* ws = reinterpret_cast<char*>(d_workspace);

    // small M: one fused kernel (epsilon in registers -> MFMA operand), no sampled weights in memory
#ifdef BF_DEV
    static const bool fused_off = getenv("BF_NO_FUSED_SMALL") != nullptr;  // developer A/B of the small-M path
#else
    constexpr bool fused_off = false;
#endif
    if (!fused_off && bf_fused_small_supported(x_dtype, y_dtype, compute_dtype, x_sample_stride, d_x, weight, bias, S, M, N, K)) {
        for (int i = 0; i < (bias ? 2 : 1); ++i)
            if ((i ? bias : weight)->prior.kind == BF_PRIOR_GAUSSIAN &&
                (!(i ? bias : weight)->prior.d_mu || !(i ? bias : weight)->prior.d_rho))
                BF_FAIL("bf_linear_fwd: gaussian prior needs d_mu/d_rho");
        ProfScope prof(BF_PROF_FUSED_SMALL, 2.0 * S * M * (double)N * K, (hipStream_t)stream);
        int rc = bf_launch_fused_small(d_x, x_dtype, x_sample_stride, weight, bias, d_y, compute_dtype, S, M, N, K, seed,
                                       sample_base, reinterpret_cast<double*>(ws + op), (hipStream_t)stream);
        if (rc) return rc;
        return bf_launch_reduce_partials(reinterpret_cast<const double*>(ws + op),
                                         (uint32_t)bf_fused_small_partial_rows(N), S, d_logprob_out, (hipStream_t)stream);
    }

    bf_tensor_t t[2];
    t[0] = *weight;
    t[0].d_sample_out = ws + ow;
    t[0].out_dtype = compute_dtype;
    int nt = 1;
    if (bias) {
        t[1] = *bias;
        t[1].d_sample_out = ws + ob;
        t[1].out_dtype = BF_DT_F32;
        nt = 2;
    }
    int rc;
    {
        ProfScope prof(BF_PROF_SAMPLE, sample_work_bytes(t, nt, S), (hipStream_t)stream);
        rc = bf_launch_sample_logprob(t, nt, S, seed, sample_base, d_logprob_out, ws + op, total - op,
                                      (hipStream_t)stream);
    }
    if (rc) return rc;
    ProfScope prof(BF_PROF_GEMM, 2.0 * S * M * (double)N * K, (hipStream_t)stream);
    return bf_launch_gemm_nt(d_x, x_dtype, x_sample_stride, ws + ow, compute_dtype,
                             bias ? reinterpret_cast<const float*>(ws + ob) : nullptr, d_y, y_dtype, S, M, N, K,
                             (hipStream_t)stream);
}

#ifdef BF_DEV  // csrc/bf_dev_api.h
size_t bf_linear_fwd_ws_workspace_bytes(int S, int N) {
    if (S < 1 || N < 1) return 0;
    return bf_align_up(bf_fused_ws_partial_rows(N) * (size_t)S * 2 * sizeof(double), 256);
}

int bf_linear_fwd_ws(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                     const bf_tensor_t* bias, void* d_y, int y_dtype, int compute_dtype, int S, int M, int N, int K,
                     uint64_t seed, uint32_t sample_base, int row_shares, double* d_logprob_out, void* d_workspace,
                     size_t workspace_bytes, void* stream) {
    if (!weight || !d_x || !d_y || !d_logprob_out) BF_FAIL("bf_linear_fwd_ws: NULL argument");
    if (weight->n != (uint64_t)N * (uint64_t)K) BF_FAIL("bf_linear_fwd_ws: weight.n != N*K");
    if (bias && bias->n != (uint64_t)N) BF_FAIL("bf_linear_fwd_ws: bias.n != N");
    if (!bf_fused_ws_supported(x_dtype, y_dtype, compute_dtype, x_sample_stride, d_x, weight, bias, S, M, N, K))
        BF_FAIL("bf_linear_fwd_ws: unsupported problem (needs 16-bit x/y of the compute dtype, N %% 64 == 0, "
                "K %% 64 == 0, K <= 768): S=%d M=%d N=%d K=%d", S, M, N, K);
    if (!d_workspace || workspace_bytes < bf_linear_fwd_ws_workspace_bytes(S, N))
        BF_FAIL("bf_linear_fwd_ws: workspace too small");
    ProfScope prof(BF_PROF_FUSED_WS, 2.0 * S * M * (double)N * K, (hipStream_t)stream);
    int rc = bf_launch_fused_ws(d_x, x_sample_stride, weight, bias, d_y, compute_dtype, S, M, N, K, seed, sample_base,
                                row_shares, reinterpret_cast<double*>(d_workspace), (hipStream_t)stream);
    if (rc) return rc;
    return bf_launch_reduce_partials(reinterpret_cast<const double*>(d_workspace),
                                     (uint32_t)bf_fused_ws_partial_rows(N), S, d_logprob_out, (hipStream_t)stream);
}
#endif  // BF_DEV

int bf_kl_grad(const bf_tensor_t* tensor, int S, uint64_t seed, uint32_t sample_base, const double* d_g,
               float* d_dmu, float* d_drho, void* stream) {
    return bf_launch_kl_grad(tensor, S, seed, sample_base, d_g, d_dmu, d_drho, (hipStream_t)stream);
}

int bf_embedding_fwd(const int64_t* d_ids, const float* d_mu, const float* d_rho, void* d_out, int out_dtype,
                     int64_t n_tokens, int64_t tokens_per_sample, int64_t V, int D, uint64_t seed, uint32_t sample_base,
                     uint32_t stream_id, void* stream) {
    return bf_launch_embedding_fwd((const long long*)d_ids, d_mu, d_rho, d_out, out_dtype, n_tokens, tokens_per_sample, V,
                                   D, seed, sample_base, stream_id, (hipStream_t)stream);
}

int bf_embedding_bwd(const int64_t* d_ids, const void* d_grad, int grad_dtype, const float* d_rho, float* d_dmu,
                     float* d_drho, int64_t n_tokens, int64_t tokens_per_sample, int64_t V, int D, uint64_t seed,
                     uint32_t sample_base, uint32_t stream_id, void* stream) {
    return bf_launch_embedding_bwd((const long long*)d_ids, d_grad, grad_dtype, d_rho, d_dmu, d_drho, n_tokens,
                                   tokens_per_sample, V, D, seed, sample_base, stream_id, (hipStream_t)stream);
}

int bf_add_layernorm(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, int param_dtype,
                     void* d_out, int dtype, int64_t rows, int N, float eps, void* stream) {
    return bf_launch_add_layernorm(d_x, d_residual, d_gamma, d_beta, param_dtype, d_out, dtype, rows, N, eps,
                                   (hipStream_t)stream);
}

int bf_embed_layernorm(const int64_t* d_ids, const int64_t* d_type_ids, const int64_t* d_pos_ids, const void* d_word,
                       const void* d_type, const void* d_pos, const void* d_gamma, const void* d_beta, int param_dtype,
                       void* d_out, int dtype, int64_t rows, int N, int seq_len, int64_t pos_rows, int64_t word_rows,
                       int64_t type_rows, int64_t pos_table_rows, float eps, void* stream) {
    return bf_launch_embed_layernorm((const long long*)d_ids, (const long long*)d_type_ids, (const long long*)d_pos_ids,
                                     d_word, d_type, d_pos, d_gamma, d_beta, param_dtype, d_out, dtype, rows, N, seq_len,
                                     pos_rows, word_rows, type_rows, pos_table_rows, eps, (hipStream_t)stream);
}

int bf_attention_fwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                     void* d_out, float* d_lse, int dtype, int B, int T, int H, int head_dim, int64_t token_stride,
                     float scaling, void* stream) {
    return bf_launch_attention_fwd(d_q, d_k, d_v, d_mask, d_mask_off, d_out, d_lse, dtype, B, T, H, head_dim,
                                   token_stride, scaling, (hipStream_t)stream);
}

int bf_attention_bwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                     const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dq, void* d_dk,
                     void* d_dv, int dtype, int B, int T, int H, int head_dim, int64_t token_stride, float scaling,
                     void* stream) {
    return bf_launch_attention_bwd(d_q, d_k, d_v, d_mask, d_mask_off, d_out, d_dout, d_lse, d_delta, d_dq, d_dk, d_dv,
                                   dtype, B, T, H, head_dim, token_stride, scaling, (hipStream_t)stream);
}

static bf_dropout_t make_dropout(float p_drop, uint64_t seed, uint32_t call, uint32_t site, uint64_t first_group = 0,
                                 const uint32_t* d_call = nullptr) {
    bf_dropout_t d;
    d.d_call = d_call;
    d.k0 = (uint32_t)seed;
    d.k1 = (uint32_t)(seed >> 32);
    d.call = call;
    d.site = site;
    d.g0_lo = (uint32_t)first_group;
    d.g0_hi = (uint32_t)(first_group >> 32);
    d.thresh = bf_dropout_thresh(p_drop);
    d.inv_keep = 1.0f / (1.0f - (float)d.thresh / 65536.0f);
    return d;
}

int bf_dropout_keep_host(uint8_t* out, uint64_t first_group, uint64_t n_groups, float p_drop, uint64_t seed, uint32_t call,
                         uint32_t site) {
    if (!out && n_groups) BF_FAIL("bf_dropout_keep_host: out is NULL");
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_dropout_keep_host: p must be in [0, 1) (got %g)", p_drop);
    const uint32_t thresh = bf_dropout_thresh(p_drop);
    for (uint64_t i = 0; i < n_groups; ++i) {
        const uint64_t g = first_group + i;
        const uint32_t keep = bf_dropout_keep8((uint32_t)g, (uint32_t)(g >> 32), call, site, (uint32_t)seed, (uint32_t)(seed >> 32), thresh);
        for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)((keep >> j) & 1u);
    }
    return 0;
}

int bf_attention_fwd_dropout(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                             void* d_out, float* d_lse, int dtype, int B, int T, int H, int head_dim, int64_t token_stride,
                             float scaling, float p_drop, uint64_t seed, uint32_t call, uint32_t site, uint64_t first_group, uint32_t* d_keep_bits,
                             const uint32_t* d_call, void* stream) {
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_attention_fwd_dropout: p must be in [0, 1) (got %g)", p_drop);
    const bf_dropout_t d = make_dropout(p_drop, seed, call, site, first_group, d_call);
    return bf_launch_attention_fwd(d_q, d_k, d_v, d_mask, d_mask_off, d_out, d_lse, dtype, B, T, H, head_dim, token_stride,
                                   scaling, (hipStream_t)stream, &d, d_keep_bits);
}

int bf_attention_bwd_dropout(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                             const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dq, void* d_dk,
                             void* d_dv, int dtype, int B, int T, int H, int head_dim, int64_t token_stride, float scaling,
                             float p_drop, const uint32_t* d_keep_bits, void* stream) {
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_attention_bwd_dropout: p must be in [0, 1) (got %g)", p_drop);
    const bf_dropout_t d = make_dropout(p_drop, 0, 0, 0);
    if (d.thresh && !d_keep_bits) BF_FAIL("bf_attention_bwd_dropout: the forward's keep bits are needed");
    return bf_launch_attention_bwd(d_q, d_k, d_v, d_mask, d_mask_off, d_out, d_dout, d_lse, d_delta, d_dq, d_dk, d_dv, dtype,
                                   B, T, H, head_dim, token_stride, scaling, (hipStream_t)stream,
                                   d.thresh ? d_keep_bits : nullptr, d.inv_keep);
}

int bf_attention_bwd_colsum(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                            const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dq, void* d_dk,
                            void* d_dv, int dtype, int B, int T, int H, int head_dim, int64_t token_stride, float scaling,
                            float p_drop, const uint32_t* d_keep_bits, int samples, float* d_partial, float* d_colsum,
                            void* stream) {
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_attention_bwd_colsum: p must be in [0, 1) (got %g)", p_drop);
    if (!d_partial || !d_colsum) BF_FAIL("bf_attention_bwd_colsum: needs d_partial ([B][H][3][64] fp32) and d_colsum ([3][samples][H*64] fp32)");
    const bf_dropout_t d = make_dropout(p_drop, 0, 0, 0);
    if (d.thresh && !d_keep_bits) BF_FAIL("bf_attention_bwd_colsum: the forward's keep bits are needed");
    return bf_launch_attention_bwd(d_q, d_k, d_v, d_mask, d_mask_off, d_out, d_dout, d_lse, d_delta, d_dq, d_dk, d_dv, dtype,
                                   B, T, H, head_dim, token_stride, scaling, (hipStream_t)stream,
                                   d.thresh ? d_keep_bits : nullptr, d.inv_keep, samples, d_partial, d_colsum);
}

int bf_add_layernorm_dropout(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, int param_dtype,
                             void* d_out, int dtype, int64_t rows, int N, float eps, float p_drop, uint64_t seed, uint32_t call,
                             uint32_t site, uint64_t first_group, const uint32_t* d_call, void* stream) {
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_add_layernorm_dropout: p must be in [0, 1) (got %g)", p_drop);
    const bf_dropout_t d = make_dropout(p_drop, seed, call, site, first_group, d_call);
    return bf_launch_add_layernorm(d_x, d_residual, d_gamma, d_beta, param_dtype, d_out, dtype, rows, N, eps,
                                   (hipStream_t)stream, &d);
}

int bf_add_layernorm_dropout_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype,
                                 const void* d_dy, void* d_dz, void* d_dx, float* d_dgamma, float* d_dbeta, void* d_workspace,
                                 size_t workspace_bytes, int dtype, int64_t rows, int N, float eps, float p_drop, uint64_t seed,
                                 uint32_t call, uint32_t site, uint64_t first_group, const uint32_t* d_call, void* stream) {
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_add_layernorm_dropout_bwd: p must be in [0, 1) (got %g)", p_drop);
    const bf_dropout_t d = make_dropout(p_drop, seed, call, site, first_group, d_call);
    return bf_launch_add_layernorm_bwd(d_x, d_residual, d_gamma, param_dtype, d_dy, d_dz, d_dgamma, d_dbeta, d_workspace,
                                       workspace_bytes, dtype, rows, N, eps, (hipStream_t)stream, &d, d_dx);
}

int bf_add_layernorm_bwd_sum(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype, const void* d_dy,
                             const void* d_dy2, void* d_dz, void* d_dx, float* d_dgamma, float* d_dbeta, void* d_workspace,
                             size_t workspace_bytes, int dtype, int64_t rows, int N, float eps, float p_drop, uint64_t seed,
                             uint32_t call, uint32_t site, uint64_t first_group, const uint32_t* d_call, void* stream) {
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) BF_FAIL("bf_add_layernorm_bwd_sum: p must be in [0, 1) (got %g)", p_drop);
    const bf_dropout_t d = make_dropout(p_drop, seed, call, site, first_group, d_call);
    return bf_launch_add_layernorm_bwd(d_x, d_residual, d_gamma, param_dtype, d_dy, d_dz, d_dgamma, d_dbeta, d_workspace,
                                       workspace_bytes, dtype, rows, N, eps, (hipStream_t)stream, d.thresh ? &d : nullptr, d_dx,
                                       d_dy2);
}

size_t bf_add_layernorm_bwd_workspace_bytes(int64_t rows, int N) { return bf_add_layernorm_bwd_ws_bytes(rows, N); }

int bf_add_layernorm_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype, const void* d_dy,
                         void* d_dz, float* d_dgamma, float* d_dbeta, void* d_workspace, size_t workspace_bytes, int dtype,
                         int64_t rows, int N, float eps, void* stream) {
    return bf_launch_add_layernorm_bwd(d_x, d_residual, d_gamma, param_dtype, d_dy, d_dz, d_dgamma, d_dbeta, d_workspace,
                                       workspace_bytes, dtype, rows, N, eps, (hipStream_t)stream);
}

int bf_gemm_tn(const void* d_a, const void* d_bm, float* d_out, int dtype, int batch, int Mc, int N, int K, void* stream) {
    if (!d_a || !d_bm || !d_out) BF_FAIL("bf_gemm_tn: null pointer");
    if (!bf_gemm256_tn_supported(dtype, batch, Mc, N, K, d_a, d_bm, d_out))
        BF_FAIL("bf_gemm_tn: needs a 16-bit dtype, Mc %% 64 == 0, N %% 8 == 0, K %% 8 == 0 and 16-byte aligned pointers");
    return bf_launch_gemm256_tn(d_a, d_bm, d_out, dtype, batch, Mc, N, K, (hipStream_t)stream);
}

int bf_gemm_nn(const void* d_x, const void* d_w, void* d_y, int dtype, int S, int M, int N, int K, void* stream) {
    if (!d_x || !d_w || !d_y) BF_FAIL("bf_gemm_nn: null pointer");
    if (!bf_gemm256_nn_supported(dtype, S, M, N, K, d_x, d_w, d_y))
        BF_FAIL("bf_gemm_nn: needs a 16-bit dtype, N %% 64 == 0, K %% 8 == 0, M * K >= 16384 and 16-byte aligned pointers");
    return bf_launch_gemm256_nn(d_x, d_w, d_y, dtype, S, M, N, K, (hipStream_t)stream);
}

int bf_gemm_nn_actgrad_supported(const void* d_x, const void* d_w, const void* d_y, const void* d_pre, int dtype, int S, int M,
                                 int N, int K) {
    return bf_gemm256_nn_actgrad_supported(dtype, S, M, N, K, d_x, d_w, d_y, d_pre) ? 1 : 0;
}

int bf_gemm_nn_actgrad(const void* d_x, const void* d_w, void* d_y, const void* d_pre, int dtype, int S, int M, int N, int K,
                       int act, void* stream) {
    if (!d_x || !d_w || !d_y || !d_pre) BF_FAIL("bf_gemm_nn_actgrad: null pointer");
    if (act != BF_ACT_GELU) BF_FAIL("bf_gemm_nn_actgrad: unknown activation %d", act);
    return bf_launch_gemm256_nn(d_x, d_w, d_y, dtype, S, M, N, K, (hipStream_t)stream, 1, d_pre, act);
}

int bf_gemm_nn_layers(const void* d_x, const void* d_w, void* d_y, int dtype, int L, int S, int M, int N, int K,
                      void* stream) {
    if (!d_x || !d_w || !d_y) BF_FAIL("bf_gemm_nn_layers: null pointer");
    if (L < 1 || L > 4) BF_FAIL("bf_gemm_nn_layers: L must be 1..4 (got %d)", L);
    if (!bf_gemm256_nn_supported(dtype, S, M, N, K, d_x, d_w, d_y) || (long long)L * S * M * N >= (1ll << 40))
        BF_FAIL("bf_gemm_nn_layers: needs a 16-bit dtype, N %% 64 == 0, K %% 8 == 0, M * K >= 16384 and 16-byte aligned pointers");
    return bf_launch_gemm256_nn(d_x, d_w, d_y, dtype, S, M, N, K, (hipStream_t)stream, L);
}

// workspace layout of bf_linear_bwd
struct BwdLayout {
    size_t w, wt, dyt, xt, dw, db, dbp, dpre, lp, part, total;
    int splits;
};

// Split-K factor of the weight-gradient GEMM dW_s = dy[s]^T x[s] (reduction over the M rows of the batch): with
// S * ceil(N/256) * ceil(K/256) output tiles a layer like 768x768 fills 90 of the 256 CUs, so the M axis is cut into
// `splits` chunks that run as extra batch entries and leave fp32 partial products for param_grad to add up.  Picks
// the factor minimising rounds * (k-steps per tile + fixed tile cost) + the partials' extra HBM traffic.
static int bwd_splits(int S, int M, int N, int K, int dtype) {
    if (dtype == BF_DT_F32) return 1;
    const double tiles = (double)S * ((N + 255) / 256) * ((K + 255) / 256);
    int best = 1;
    double best_cost = 1e30;
    for (int sp = 1; sp <= 16; sp *= 2) {
        if (M % (64 * sp) || M / sp < 256) break;
        const double rounds = ceil(tiles * sp / 256.0);
        const double ksteps = (double)M / sp / 64.0;
        const double traffic_us = sp > 1 ? (double)sp * S * N * K * 8.0 / 3.0e6 : 0.0;  // write + read at ~3 TB/s
        const double cost = rounds * (ksteps + 6.0) * 2.0 + traffic_us;                  // ~2 us per 256x256x64 k-step
        if (cost < best_cost) best_cost = cost, best = sp;
    }
    return best;
}

static BwdLayout bwd_layout(int S, int M, int N, int K, int has_bias, int dtype, int act = BF_ACT_NONE) {
    const size_t es = bf_dtype_size(dtype);
    BwdLayout L;
    L.splits = bwd_splits(S, M, N, K, dtype);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += bf_align_up(bytes, 256);
        return o;
    };
    L.w = take((size_t)S * N * K * es);
    L.wt = take((size_t)S * N * K * es);
    L.dyt = take((size_t)S * N * M * es);
    L.xt = take((size_t)S * K * M * es);
    L.dw = take((size_t)L.splits * S * N * K * sizeof(float));
    // a fused activation's backward writes dpre = dy * act'(pre) here and its column sums into db (bias or not)
    const bool sums = has_bias || act != BF_ACT_NONE;
    L.db = take(sums ? (size_t)S * N * sizeof(float) : 0);
    L.dbp = take(sums ? bf_colsum_workspace_bytes(S, M, N) : 0);
    L.dpre = take(act != BF_ACT_NONE ? (size_t)S * M * N * es : 0);
    L.lp = take((size_t)S * 2 * sizeof(double));
    bf_tensor_t t;
    memset(&t, 0, sizeof(t));
    t.n = (uint64_t)N * K;
    L.part = take(bf_sample_partials_bytes(&t, 1, S));
    L.total = off;
    return L;
}

size_t bf_linear_bwd_workspace_bytes(int S, int M, int N, int K, int has_bias, int dtype, int act) {
    if (S < 1 || M < 1 || N < 1 || K < 1) return 0;
    return bwd_layout(S, M, N, K, has_bias, dtype, act).total;
}

int bf_linear_bwd_splits(int S, int M, int N, int K, int dtype) {
    if (S < 1 || M < 1 || N < 1 || K < 1) return 0;
    return bwd_splits(S, M, N, K, dtype);
}

size_t bf_param_grad_table_bytes(const bf_pgrad_t* entries, int n, uint32_t* total_blocks) {
    if (!entries || n < 1) return 0;
    return bf_pgrad_table_bytes(entries, n, total_blocks);
}

int bf_param_grad_table_build(const bf_pgrad_t* entries, int n, void* h_blob, size_t blob_bytes) {
    if (!entries || n < 1 || !h_blob) BF_FAIL("bf_param_grad_table_build: no entries or no blob");
    return bf_pgrad_table_build(entries, n, h_blob, blob_bytes);
}

int bf_param_grad_table(const void* d_blob, int n, uint32_t total_blocks, int S, uint64_t seed, uint32_t sample_base,
                        void* stream) {
    return bf_launch_pgrad_table(d_blob, n, total_blocks, S, seed, sample_base, (hipStream_t)stream);
}

int bf_linear_bwd(const void* d_x, int64_t x_sample_stride, const void* d_dy, int dtype, const bf_tensor_t* weight,
                  const bf_tensor_t* bias, void* d_dx, float* d_dmu_w, float* d_drho_w, float* d_dmu_b,
                  float* d_drho_b, int S, int M, int N, int K, uint64_t seed, uint32_t sample_base, int act,
                  const void* d_act_pre, const float* d_dy_colsum, float* d_dw_keep, float* d_db_keep, void* d_workspace,
                  size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_x || !d_dy || !weight || !d_drho_w) BF_FAIL("bf_linear_bwd: NULL argument");
    if (act != BF_ACT_NONE && act != BF_ACT_GELU) BF_FAIL("bf_linear_bwd: unknown activation %d", act);
    if (act != BF_ACT_NONE && !d_act_pre) BF_FAIL("bf_linear_bwd: a fused activation needs the forward's pre-activation");
    if (S < 1 || M < 1 || N < 1 || K < 1) BF_FAIL("bf_linear_bwd: bad shape S=%d M=%d N=%d K=%d", S, M, N, K);
    if (weight->n != (uint64_t)N * (uint64_t)K) BF_FAIL("bf_linear_bwd: weight.n != N*K");
    if (bias && (bias->n != (uint64_t)N || !d_drho_b)) BF_FAIL("bf_linear_bwd: bad bias arguments");
    if (dtype < BF_DT_F32 || dtype > BF_DT_F16) BF_FAIL("bf_linear_bwd: bad dtype %d", dtype);
    if (x_sample_stride != 0 && x_sample_stride != (int64_t)M * K) BF_FAIL("bf_linear_bwd: x_sample_stride must be 0 or M*K");
    const BwdLayout L = bwd_layout(S, M, N, K, bias != nullptr, dtype, act);
    if (!d_workspace || workspace_bytes < L.total)
        BF_FAIL("bf_linear_bwd: workspace too small (%zu < %zu bytes)", workspace_bytes, L.total);
    char* ws = reinterpret_cast<char*>(d_workspace);
    const int es = (int)bf_dtype_size(dtype);
    if (d_db_keep && !bias) BF_FAIL("bf_linear_bwd: d_db_keep without a bias");
    // the bias gradient's per-sample column sums of dy, [S][N] fp32: in the workspace, or where the caller keeps them
    float* db_dst = d_db_keep ? d_db_keep : reinterpret_cast<float*>(ws + L.db);

    int rc = 0;
    // 0. the forward applied act() in its GEMM epilogue: d_dy is the gradient of act(y); dy = d_dy * act'(y_pre), and the
    //    bias gradient's column sums of dy come out of the same pass
    bool fused_colsum = false;
    if (act != BF_ACT_NONE) {
        if ((rc = bf_launch_gelu_bwd_colsum(d_dy, d_act_pre, ws + L.dpre, dtype, S, M, N, reinterpret_cast<float*>(ws + L.dbp),
                                            db_dst, stream)))
            return rc;
        d_dy = ws + L.dpre;
        fused_colsum = true;
    }
    // 1. W_s: the forward's own samples if the caller still holds them (weight->d_sample_out, same dtype), else
    //    regenerated from the same counters; no prior term needed
    const char* w_s = ws + L.w;
    if (weight->d_sample_out && weight->out_dtype == dtype) {
        w_s = reinterpret_cast<const char*>(weight->d_sample_out);
    } else if (d_dx) {
        bf_tensor_t t = *weight;
        t.prior.kind = BF_PRIOR_NONE;
        t.d_sample_out = ws + L.w;
        t.out_dtype = dtype;
        rc = bf_launch_sample_logprob(&t, 1, S, seed, sample_base, reinterpret_cast<double*>(ws + L.lp), ws + L.part,
                                      L.total - L.part, stream);
        if (rc) return rc;
    }
    if (d_dx) {
        // 2. dx[s] = dy[s] W_s: the NN form of the 256-wide kernel reads W_s [N][K] as it was sampled (contraction-major,
        //    fragments through the LDS transpose read); other shapes: a transposed copy W_s^T [K][N] and the NT kernel
        if (bf_gemm256_nn_supported(dtype, S, M, N, K, d_dy, w_s, d_dx)) {
            if ((rc = bf_launch_gemm256_nn(d_dy, w_s, d_dx, dtype, S, M, N, K, stream))) return rc;
        } else {
            if ((rc = bf_launch_transpose(w_s, ws + L.wt, es, S, N, K, stream))) return rc;
            if ((rc = bf_launch_gemm_nt(d_dy, dtype, (int64_t)M * N, ws + L.wt, dtype, nullptr, d_dx, dtype, S, M, K, N, stream)))
                return rc;
        }
    }
    // 3. dW_s = dy[s]^T x[s], fp32 out.  With split-K the M axis is cut into `sp` chunks that are multiplied as sp * S
    //    batch entries of M / sp rows each.  The TN form of the 256-wide kernel reads dy and x as they are
    //    (contraction-major, fragments through the LDS transpose read); shapes it does not take go through transposed
    //    copies and the NT kernel.
    const int sp = x_sample_stride == 0 ? 1 : L.splits;
    const int Mc = M / sp;
    if (d_dw_keep && (x_sample_stride == 0 || ((uintptr_t)d_dw_keep & 15)))
        BF_FAIL("bf_linear_bwd: d_dw_keep needs per-sample activations (x_sample_stride = M*K) and a 16-byte aligned buffer");
    char* dw_dst = d_dw_keep ? reinterpret_cast<char*>(d_dw_keep) : ws + L.dw;  // [S][sp][N][K] fp32
    if (x_sample_stride != 0 && bf_gemm256_tn_supported(dtype, S * sp, Mc, N, K, d_dy, d_x, dw_dst)) {
        if ((rc = bf_launch_gemm256_tn(d_dy, d_x, reinterpret_cast<float*>(dw_dst), dtype, S * sp, Mc, N, K, stream)))
            return rc;
    } else {
        const bool with_t = !fused_colsum && bias && bf_transpose_colsum_supported(dtype, S * sp, Mc, N, d_dy, ws + L.dyt);
        if (with_t) {  // the bias gradient's column sums ride along with the transpose of dy
            fused_colsum = true;
            if ((rc = bf_launch_transpose_colsum(d_dy, ws + L.dyt, dtype, S * sp, Mc, N, sp, reinterpret_cast<float*>(ws + L.dbp),
                                                 db_dst, stream)))
                return rc;
        } else if ((rc = bf_launch_transpose(d_dy, ws + L.dyt, es, S * sp, Mc, N, stream))) {
            return rc;
        }
        const int xs_batch = x_sample_stride == 0 ? 1 : S * sp;
        if ((rc = bf_launch_transpose(d_x, ws + L.xt, es, xs_batch, Mc, K, stream))) return rc;
        // operands: "x" = dy^T [S][N][M] (stride N*M), "w" = x^T [S][K][M]; a shared x is broadcast by passing it S times
        if (x_sample_stride == 0) {
            for (int s = 0; s < S; ++s)
                if ((rc = bf_launch_gemm_nt(ws + L.dyt + (size_t)s * N * M * es, dtype, 0, ws + L.xt, dtype, nullptr,
                                            dw_dst + (size_t)s * N * K * sizeof(float), BF_DT_F32, 1, N, K, M, stream)))
                    return rc;
        } else if ((rc = bf_launch_gemm_nt(ws + L.dyt, dtype, (int64_t)N * Mc, ws + L.xt, dtype, nullptr, dw_dst, BF_DT_F32,
                                           S * sp, N, K, Mc, stream))) {
            return rc;
        }
    }
    // 4. reduce over samples (and split-K partials) with eps regenerated — unless the caller keeps dW_s and reduces the
    //    weights of all its layers in one launch later (d_dw_keep, bf_param_grad_table)
    if (!d_dw_keep && (rc = bf_launch_param_grad(reinterpret_cast<const float*>(dw_dst), weight->d_rho, weight->n, S, sp, seed,
                                                 sample_base, weight->stream_id, d_dmu_w, d_drho_w, stream)))
        return rc;
    if (bias) {
        // the column sums of dy: from the pass that formed dy (fused activation / transpose), from the kernel that produced
        // d_dy (d_dy_colsum), or a pass of their own
        // (d_db_keep: the sums stay in the caller's buffer — or in d_dy_colsum, which the caller then copies itself — and the
        // reduction over the samples is left to its bf_param_grad_table launch)
        const float* colsum = db_dst;
        if (!fused_colsum && d_dy_colsum && act == BF_ACT_NONE) colsum = d_dy_colsum;
        else if (!fused_colsum && (rc = bf_launch_colsum(d_dy, dtype, db_dst, S, M, N, reinterpret_cast<float*>(ws + L.dbp), stream)))
            return rc;
        if (!d_db_keep && (rc = bf_launch_param_grad(colsum, bias->d_rho, bias->n, S, 1, seed,
                                                     sample_base, bias->stream_id, d_dmu_b, d_drho_b, stream)))
            return rc;
    }
    return 0;
}

}  // extern "C"
