// bf_backward.hip — backward of the sampled-weight linear layer.
//
// Reference behaviour being reproduced: autograd through `F.linear(input, mu + eps * softplus(rho), ...)`
// (/root/reference/bayeformers/nn/layers/linear.py:97,104, parameters/gaussian.py:100-101) with eps a constant of
// the graph (drawn under no_grad) and the two log-prob scalars DETACHED (`.data =`, linear.py:99-102), i.e. the KL
// terms give no gradient in the reference; only the likelihood path does:
//     dX[s]   = dY[s] W_s                      dW_s = dY[s]^T X[s]
//     dmu_w   = sum_s dW_s                     drho_w = (sum_s dW_s * eps_s) * softplus'(rho_w)
//     dmu_b   = sum_s sum_m dY[s][m][:]        drho_b = (sum_s db_s * eps_b,s) * softplus'(rho_b)
// eps is REGENERATED from the Philox counter (same seed / sample index / stream as the forward): nothing but x is
// kept from the forward.
//
// Kernels here are the glue around the two MFMA GEMMs (bf_gemm_nt): a batched 16-bit transpose (the NT kernel wants
// both operands contiguous along the reduction axis), the column sum for the bias, and the eps-weighted reduction
// over samples.  All HBM-bound streaming kernels.
#include "bf_common.h"
#include "bf_philox.h"

namespace {

// out[b][c][r] = in[b][r][c] for 2- or 4-byte elements; 64x64 tile through LDS (padded), 256 threads.
template <typename E>
__global__ __launch_bounds__(256) void transpose_kernel(const E* __restrict__ in, E* __restrict__ out, int rows,
                                                        int cols) {
    __shared__ E tile[64][64 + 4 / sizeof(E)];
    const long long b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row-groups
    in += b * (long long)rows * cols;
    out += b * (long long)rows * cols;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < rows && c < cols) ? in[(long long)r * cols + c] : (E)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(long long)c * rows + r] = tile[tx][ty * 16 + i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ out, int M, int N) {
    // grid = (ceil(N/64), S); block = 64 columns x 4 row-lanes
    __shared__ float sh[4][64];
    const int s = blockIdx.y, n = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;
    const T* p = dy + (long long)s * M * N;
    float acc = 0.f;
    if (n < N)
        for (int m = lane_r; m < M; m += 4) acc += (float)p[(long long)m * N + n];
    sh[lane_r][threadIdx.x & 63] = acc;
    __syncthreads();
    if (lane_r == 0 && n < N) out[(long long)s * N + n] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// dmu[e] = sum_s dw[s][e];  drho[e] = (sum_s dw[s][e] * eps(s, e)) * softplus'(rho[e]).  thread = 4 scalars.
__global__ __launch_bounds__(256) void param_grad_kernel(const float* __restrict__ dw, const float* __restrict__ rho,
                                                         unsigned long long n, int S, uint32_t k0, uint32_t k1,
                                                         uint32_t sample_base, const uint32_t* __restrict__ counter,
                                                         uint32_t stream, float* __restrict__ dmu,
                                                         float* __restrict__ drho) {
    sample_base += counter ? *counter : 0u;
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long e0 = g * 4;
    if (e0 >= n) return;
    const int nv = n - e0 >= 4 ? 4 : (int)(n - e0);
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, se[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        float z[4];
        bf_normal4_dev((uint32_t)g, (uint32_t)(g >> 32), sample_base + (uint32_t)s, stream, k0, k1, z);
        const float* p = dw + (unsigned long long)s * n + e0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = i < nv ? p[i] : 0.f;
            sm[i] += d;
            se[i] = fmaf(d, z[i], se[i]);
        }
    }
    for (int i = 0; i < nv; ++i) {
        if (dmu) dmu[e0 + i] = sm[i];
        const float r = rho[e0 + i];
        // torch softplus backward (beta = 1, threshold = 20): grad * (r > 20 ? 1 : e^r / (e^r + 1))
        const float zr = expf(r);
        drho[e0 + i] = se[i] * (r > 20.0f ? 1.0f : zr / (zr + 1.0f));
    }
}

}  // namespace

int bf_launch_transpose(const void* d_in, void* d_out, int elem_size, int batch, int rows, int cols, hipStream_t stream) {
    if (!d_in || !d_out) BF_FAIL("bf_transpose: NULL argument");
    if (batch < 1 || rows < 1 || cols < 1 || batch > 65535) BF_FAIL("bf_transpose: bad shape %d x %d x %d", batch, rows, cols);
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
    if (elem_size == 2)
        hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t*)d_in, (uint16_t*)d_out, rows, cols);
    else if (elem_size == 4)
        hipLaunchKernelGGL(transpose_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t*)d_in, (uint32_t*)d_out, rows, cols);
    else
        BF_FAIL("bf_transpose: element size %d", elem_size);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_colsum(const void* d_dy, int dtype, float* d_out, int S, int M, int N, hipStream_t stream) {
    if (!d_dy || !d_out) BF_FAIL("bf_colsum: NULL argument");
    dim3 grid((N + 63) / 64, S);
    if (dtype == BF_DT_BF16)
        hipLaunchKernelGGL(colsum_kernel<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)d_dy, d_out, M, N);
    else if (dtype == BF_DT_F16)
        hipLaunchKernelGGL(colsum_kernel<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)d_dy, d_out, M, N);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)d_dy, d_out, M, N);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_launch_param_grad(const float* d_dw, const float* d_rho, uint64_t n, int S, uint64_t seed, uint32_t sample_base,
                         uint32_t stream_id, float* d_dmu, float* d_drho, hipStream_t stream) {
    if (!d_dw || !d_rho || !d_drho) BF_FAIL("bf_param_grad: NULL argument");
    if (n == 0 || S < 1) BF_FAIL("bf_param_grad: empty");
    const uint64_t groups = (n + 3) / 4;
    hipLaunchKernelGGL(param_grad_kernel, dim3((uint32_t)((groups + 255) / 256)), dim3(256), 0, stream, d_dw, d_rho,
                       (unsigned long long)n, S, (uint32_t)seed, (uint32_t)(seed >> 32), sample_base, bf_sample_counter(), stream_id, d_dmu,
                       d_drho);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
