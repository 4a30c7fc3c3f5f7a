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
void bf_set_error(const char* fmt, ...);

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

// internal launchers (defined in bf_sample.hip / bf_gemm.hip), all asynchronous on `stream`
int bf_launch_philox_normal(float* d_out, uint64_t n, int S, uint64_t seed, uint32_t sample_base, uint32_t stream_id,
                            hipStream_t stream);
size_t bf_sample_partials_bytes(const bf_tensor_t* tensors, int n_tensors, int S);
int bf_launch_sample_logprob(const bf_tensor_t* tensors, int n_tensors, int S, uint64_t seed, uint32_t sample_base,
                             double* d_logprob_out, void* d_workspace, size_t workspace_bytes, hipStream_t stream);
size_t bf_table_blob_bytes(const bf_tensor_t* tensors, int n_tensors, uint32_t* total_blocks);
int bf_table_build(const bf_tensor_t* tensors, int n_tensors, void* h_blob, size_t blob_bytes, uint32_t* h_block_begin);
int bf_launch_sample_table(const void* d_blob, int n_tensors, uint32_t block_begin, uint32_t block_end, int S,
                           uint64_t seed, uint32_t sample_base, double* d_partials, hipStream_t stream);
int bf_launch_reduce_groups(const double* d_partials, const uint32_t* d_rows, int G, int S, double* d_out,
                            hipStream_t stream);
int bf_launch_gemm_nt(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                      const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, hipStream_t stream,
                      int act = BF_ACT_NONE, int layers = 1);

// erf-GELU x/2 (1 + erf(x / sqrt 2)), the activation of HF BERT's intermediate layer, on the fp32 accumulators.
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-exact for a bf16/fp16 or fp32 epilogue) on the
// hardware rcp/exp2 units: ~14 VALU ops per value instead of libm erff's ~45.
__device__ __forceinline__ float bf_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = fmaf(-p * t, e, 1.0f);
    return 0.5f * x * (1.0f + __builtin_copysignf(erf_abs, x));
}
__device__ __forceinline__ f32x4_t bf_apply_act(f32x4_t v, int act) {
    if (act == BF_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = bf_gelu(v[j]);
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
int bf_launch_reduce_partials(const double* d_partials, uint32_t nrows, int S, double* d_out, hipStream_t stream);
bool bf_fused_small_supported(int x_dtype, int y_dtype, int compute_dtype, int64_t x_sample_stride, const void* d_x,
                              const bf_tensor_t* weight, const bf_tensor_t* bias, int S, int M, int N, int K);
size_t bf_fused_small_partial_rows(int N);
int bf_launch_fused_small(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                          const bf_tensor_t* bias, void* d_y, int compute_dtype, int S, int M, int N, int K, uint64_t seed,
                          uint32_t sample_base, double* d_partials, hipStream_t stream);
int bf_launch_kl_grad(const bf_tensor_t* t, int S, uint64_t seed, uint32_t sample_base, const double* d_g,
                      float* d_dmu, float* d_drho, hipStream_t stream);
int bf_launch_embedding_fwd(const long long* d_ids, const float* d_mu, const float* d_rho, void* d_out, int out_dtype,
                            long long n_tokens, long long tokens_per_sample, long long V, int D, uint64_t seed,
                            uint32_t sample_base, uint32_t stream_id, hipStream_t stream);
int bf_launch_embedding_bwd(const long long* d_ids, const void* d_grad, int grad_dtype, const float* d_rho, float* d_dmu,
                            float* d_drho, long long n_tokens, long long tokens_per_sample, long long V, int D,
                            uint64_t seed, uint32_t sample_base, uint32_t stream_id, hipStream_t stream);
int bf_launch_add_layernorm(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta,
                            int param_dtype, void* d_out, int dtype, long long rows, int N, float eps,
                            hipStream_t stream);
int bf_launch_attention_fwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, void* d_out, int dtype,
                            int B, int T, int H, int head_dim, long long token_stride, float scaling,
                            hipStream_t stream);
size_t bf_add_layernorm_bwd_ws_bytes(long long rows, int N);
int bf_launch_add_layernorm_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype,
                                const void* d_dy, void* d_dz, float* d_dgamma, float* d_dbeta, void* d_workspace,
                                size_t workspace_bytes, int dtype, long long rows, int N, float eps, hipStream_t stream);
