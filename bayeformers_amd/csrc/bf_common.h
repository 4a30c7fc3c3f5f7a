// bf_common.h — internal helpers shared by the HIP translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bayeformers_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// thread-local error string behind bf_last_error()
void bf_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define BF_FAIL(...)               \
    do {                           \
        bf_set_error(__VA_ARGS__); \
        return 1;                  \
    } while (0)

#define BF_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) BF_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

static inline size_t bf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline size_t bf_dtype_size(int dt) { return dt == BF_DT_F32 ? 4 : 2; }

// optional device-resident Monte-Carlo sample counter (bf_set_sample_counter): kernels add *counter to sample_base
const uint32_t* bf_sample_counter();

// device-visible address of the stale-prior counter (bf_stale_counter; pinned host memory), NULL if it could not be set up
uint32_t* bf_stale_counter_dev();

// A prior's baked constants against the device scalars they were read from (bf_prior_t.d_pi / d_sigma1 / d_sigma2): true =
// they differ (the module's tensors were edited in place behind the host's back).
struct bf_prior_check_t {
    const float* p[3];
    float v[3];
};
static inline bf_prior_check_t bf_prior_check_of(const bf_prior_t& pr) {
    bf_prior_check_t c{};
    if (pr.kind == BF_PRIOR_MIXTURE && pr.d_pi && pr.d_sigma1 && pr.d_sigma2) {
        c.p[0] = pr.d_pi; c.p[1] = pr.d_sigma1; c.p[2] = pr.d_sigma2;
        c.v[0] = pr.pi; c.v[1] = pr.sigma1; c.v[2] = pr.sigma2;
    }
    return c;
}
__device__ __forceinline__ bool bf_prior_check_failed(const bf_prior_check_t& c) {
    if (!c.p[0]) return false;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) bad |= __float_as_uint(*c.p[i]) != __float_as_uint(c.v[i]);
    return bad;
}
__device__ __forceinline__ void bf_stale_bump(uint32_t* counter) {
    if (counter) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// internal launchers (defined in bf_sample.hip / bf_gemm.hip), all asynchronous on `stream`
int bf_launch_philox_normal(float* d_out, uint64_t n, int S, uint64_t seed, uint32_t sample_base, uint32_t stream_id,
                            hipStream_t stream);
size_t bf_sample_partials_bytes(const bf_tensor_t* tensors, int n_tensors, int S);
int bf_launch_sample_logprob(const bf_tensor_t* tensors, int n_tensors, int S, uint64_t seed, uint32_t sample_base,
                             double* d_logprob_out, void* d_workspace, size_t workspace_bytes, hipStream_t stream);
size_t bf_table_blob_bytes(const bf_tensor_t* tensors, int n_tensors, uint32_t* total_blocks);
int bf_table_build(const bf_tensor_t* tensors, int n_tensors, void* h_blob, size_t blob_bytes, uint32_t* h_block_begin, int32_t* h_kinds = nullptr);
int bf_launch_sample_table(const void* d_blob, int n_tensors, uint32_t block_begin, uint32_t block_end, int S,
                           uint64_t seed, uint32_t sample_base, double* d_partials, hipStream_t stream, int prior_kinds = 0);
int bf_launch_reduce_groups(const double* d_partials, const uint32_t* d_rows, int G, int S, double* d_out,
                            hipStream_t stream);
int bf_launch_gemm_nt(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                      const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, hipStream_t stream,
                      int act = BF_ACT_NONE, int layers = 1, void* d_pre = nullptr);
// out = gelu(in) elementwise (16-bit or fp32 tensors of n elements)
int bf_launch_gelu(const void* d_in, void* d_out, int dtype, uint64_t n, hipStream_t stream);
// dpre = dy * gelu'(pre) elementwise on [S][M][N] 16-bit tensors, with the column sums of dpre per sample
// (d_colsum [S][N] fp32; d_partial: bf_colsum_workspace_bytes of scratch)
bool bf_gelu_bwd_colsum_supported(int dtype, int S, int M, int N, const void* d_dy, const void* d_pre, const void* d_out);
int bf_launch_gelu_bwd_colsum(const void* d_dy, const void* d_pre, void* d_dpre, int dtype, int S, int M, int N,
                              float* d_partial, float* d_colsum, hipStream_t stream);

// erf-GELU x/2 (1 + erf(x / sqrt 2)), the activation of HF BERT's intermediate layer, on the fp32 accumulators.
// With q = 1/2 erfc(|x| / sqrt 2) = Phi(-|x|):  gelu(x) = x (1 - q) for x >= 0 and x q for x < 0, i.e. in one
// expression gelu(x) = x/2 + |x| (1/2 - q)  (the same cancellation for very negative x as the textbook
// x/2 (1 + erf(x / sqrt 2)) has).
// erfc by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-exact for a bf16/fp16 or fp32 epilogue):
// erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + 0.3275911 z), on the hardware rcp / exp2
// units.  Written on pairs so that everything but |x|, rcp and exp2 runs on the packed fp32 pipe
// (v_pk_fma_f32 / v_pk_mul_f32): ~8.5 issue slots per value, 2 of them transcendental, instead of ~17 for a scalar form.
__device__ __forceinline__ f32x2_t bf_pk_mul(f32x2_t a, f32x2_t b) {
    f32x2_t r;  // the compiler scalarises a product whose consumers are per-element transcendentals
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2_t bf_gelu2(f32x2_t x) {
    const f32x2_t ax = __builtin_elementwise_abs(x);
    const f32x2_t d = __builtin_elementwise_fma(ax, (f32x2_t)(0.3275911f * 0.70710678118654752f), (f32x2_t)(1.0f));
    const f32x2_t hx = x * (f32x2_t)(0.5f);
    const f32x2_t xx = bf_pk_mul(bf_pk_mul(x, (f32x2_t)(-0.72134752044448170f)), x);  // -x^2/2 * log2(e)
    f32x2_t t, e;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        t[j] = __builtin_amdgcn_rcpf(d[j]);
        e[j] = __builtin_amdgcn_exp2f(xx[j]);
    }
    // p = -1/2 (a1 + t (a2 + ...)): the sign and the 1/2 live in the coefficients
    f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t)(-0.5f * 1.061405429f), (f32x2_t)(-0.5f * -1.453152027f));
    p = __builtin_elementwise_fma(t, p, (f32x2_t)(-0.5f * 1.421413741f));
    p = __builtin_elementwise_fma(t, p, (f32x2_t)(-0.5f * -0.284496736f));
    p = __builtin_elementwise_fma(t, p, (f32x2_t)(-0.5f * 0.254829592f));
    const f32x2_t r = __builtin_elementwise_fma(p * t, e, (f32x2_t)(0.5f));  // 1/2 - q
    return __builtin_elementwise_fma(ax, r, hx);
}
// d/dx of the erf-GELU: Phi(x) + x phi(x), from the same erfc approximation (r = 1/2 - q = Phi(|x|) - 1/2):
// Phi(x) = 1/2 + sign(x) r,  phi(x) = exp(-x^2/2) / sqrt(2 pi).
__device__ __forceinline__ float bf_gelu_grad(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * -0.72134752044448170f * x);
    float p = fmaf(t, -0.5f * 1.061405429f, -0.5f * -1.453152027f);
    p = fmaf(t, p, -0.5f * 1.421413741f);
    p = fmaf(t, p, -0.5f * -0.284496736f);
    p = fmaf(t, p, -0.5f * 0.254829592f);
    const float r = fmaf(p * t, e, 0.5f);
    return fmaf(x * 0.39894228040143268f, e, 0.5f + copysignf(r, x));
}
// The same form with A&S 7.1.25 (three terms, |error| <= 2.5e-5 on erf, i.e. <= 1.1e-5 |x| on gelu): for results that are
// rounded to bf16 / fp16 anyway (relative precision 3.9e-3 / 4.9e-4) — two fewer FMAs per value in the VALU-bound epilogue
// of the FFN-up GEMM.
__device__ __forceinline__ f32x2_t bf_gelu2_16(f32x2_t x) {
    const f32x2_t ax = __builtin_elementwise_abs(x);
    const f32x2_t d = __builtin_elementwise_fma(ax, (f32x2_t)(0.47047f * 0.70710678118654752f), (f32x2_t)(1.0f));
    const f32x2_t hx = x * (f32x2_t)(0.5f);
    const f32x2_t xx = bf_pk_mul(bf_pk_mul(x, (f32x2_t)(-0.72134752044448170f)), x);  // -x^2/2 * log2(e)
    f32x2_t t, e;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        t[j] = __builtin_amdgcn_rcpf(d[j]);
        e[j] = __builtin_amdgcn_exp2f(xx[j]);
    }
    f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t)(-0.5f * 0.7478556f), (f32x2_t)(-0.5f * -0.0958798f));
    p = __builtin_elementwise_fma(t, p, (f32x2_t)(-0.5f * 0.3480242f));
    const f32x2_t r = __builtin_elementwise_fma(p * t, e, (f32x2_t)(0.5f));  // 1/2 - q
    return __builtin_elementwise_fma(ax, r, hx);
}
__device__ __forceinline__ float bf_gelu(float x) {
    const f32x2_t v = {x, x};
    return bf_gelu2(v)[0];
}
// OUT16: the result is stored as bf16 / fp16
template <bool OUT16 = false>
__device__ __forceinline__ f32x4_t bf_apply_act(f32x4_t v, int act) {
    if (act == BF_ACT_GELU) {
        const f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
        const f32x2_t lo = OUT16 ? bf_gelu2_16(a) : bf_gelu2(a), hi = OUT16 ? bf_gelu2_16(b) : bf_gelu2(b);
        v = f32x4_t{lo[0], lo[1], hi[0], hi[1]};
    }
    return v;
}
int bf_launch_transpose(const void* d_in, void* d_out, int elem_size, int batch, int rows, int cols, hipStream_t stream);
size_t bf_colsum_workspace_bytes(int S, int M, int N);
bool bf_transpose_colsum_supported(int dtype, int batch, int rows, int cols, const void* d_in, const void* d_out);
int bf_launch_transpose_colsum(const void* d_in, void* d_out, int dtype, int batch, int rows, int cols,
                               int batch_per_group, float* d_partial, float* d_out_sums, hipStream_t stream);
int bf_launch_colsum(const void* d_dy, int dtype, float* d_out, int S, int M, int N, float* d_partial,
                     hipStream_t stream);
int bf_launch_param_grad(const float* d_dw, const float* d_rho, uint64_t n, int S, int splits, uint64_t seed,
                         uint32_t sample_base, uint32_t stream_id, float* d_dmu, float* d_drho, hipStream_t stream);
size_t bf_pgrad_table_bytes(const bf_pgrad_t* t, int n, uint32_t* total_blocks);
int bf_pgrad_table_build(const bf_pgrad_t* t, int n, void* h_blob, size_t blob_bytes);
int bf_launch_pgrad_table(const void* d_blob, int n, uint32_t total_blocks, int S, uint64_t seed, uint32_t sample_base,
                          hipStream_t stream);
int bf_launch_reduce_partials(const double* d_partials, uint32_t nrows, int S, double* d_out, hipStream_t stream);
bool bf_fused_small_supported(int x_dtype, int y_dtype, int compute_dtype, int64_t x_sample_stride, const void* d_x,
                              const bf_tensor_t* weight, const bf_tensor_t* bias, int S, int M, int N, int K);
size_t bf_fused_small_partial_rows(int N);
int bf_launch_fused_small(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                          const bf_tensor_t* bias, void* d_y, int compute_dtype, int S, int M, int N, int K, uint64_t seed,
                          uint32_t sample_base, double* d_partials, hipStream_t stream);
bool bf_fused_ws_supported(int x_dtype, int y_dtype, int compute_dtype, int64_t x_sample_stride, const void* d_x,
                           const bf_tensor_t* weight, const bf_tensor_t* bias, int S, int M, int N, int K);
size_t bf_fused_ws_partial_rows(int N);
int bf_launch_fused_ws(const void* d_x, int64_t x_sample_stride, const bf_tensor_t* weight, const bf_tensor_t* bias,
                       void* d_y, int compute_dtype, int S, int M, int N, int K, uint64_t seed, uint32_t sample_base,
                       int msplit, double* d_partials, hipStream_t stream);
int bf_launch_kl_grad(const bf_tensor_t* t, int S, uint64_t seed, uint32_t sample_base, const double* d_g,
                      float* d_dmu, float* d_drho, hipStream_t stream);
int bf_launch_embedding_fwd(const long long* d_ids, const float* d_mu, const float* d_rho, void* d_out, int out_dtype,
                            long long n_tokens, long long tokens_per_sample, long long V, int D, uint64_t seed,
                            uint32_t sample_base, uint32_t stream_id, hipStream_t stream);
int bf_launch_embedding_bwd(const long long* d_ids, const void* d_grad, int grad_dtype, const float* d_rho, float* d_dmu,
                            float* d_drho, long long n_tokens, long long tokens_per_sample, long long V, int D,
                            uint64_t seed, uint32_t sample_base, uint32_t stream_id, hipStream_t stream);
struct bf_dropout_t;  // bf_philox.h
int bf_launch_add_layernorm(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta,
                            int param_dtype, void* d_out, int dtype, long long rows, int N, float eps,
                            hipStream_t stream, const bf_dropout_t* drop = nullptr);
int bf_launch_embed_layernorm(const long long* d_ids, const long long* d_type_ids, const long long* d_pos_ids,
                              const void* d_word, const void* d_type, const void* d_pos, const void* d_gamma,
                              const void* d_beta, int param_dtype, void* d_out, int dtype, long long rows, int N,
                              int seq_len, long long pos_rows, long long word_rows, long long type_rows,
                              long long pos_table_rows, float eps, hipStream_t stream);
int bf_launch_attention_fwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask,
                            const unsigned char* d_mask_off, void* d_out, float* d_lse, int dtype, int B, int T, int H,
                            int head_dim, long long token_stride, float scaling, hipStream_t stream,
                            const bf_dropout_t* drop = nullptr, uint32_t* d_keep_bits = nullptr);
int bf_launch_attention_bwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask,
                            const unsigned char* d_mask_off, const void* d_out, const void* d_dout, const float* d_lse,
                            float* d_delta, void* d_dq, void* d_dk, void* d_dv, int dtype, int B, int T, int H,
                            int head_dim, long long token_stride, float scaling, hipStream_t stream,
                            const uint32_t* d_keep_bits = nullptr, float inv_keep = 1.0f, int samples = 0,
                            float* d_cs_partial = nullptr, float* d_colsum = nullptr);
size_t bf_add_layernorm_bwd_ws_bytes(long long rows, int N);
int bf_launch_add_layernorm_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype,
                                const void* d_dy, void* d_dz, float* d_dgamma, float* d_dbeta, void* d_workspace,
                                size_t workspace_bytes, int dtype, long long rows, int N, float eps, hipStream_t stream,
                                const bf_dropout_t* drop = nullptr, void* d_dx = nullptr, const void* d_dy2 = nullptr);
