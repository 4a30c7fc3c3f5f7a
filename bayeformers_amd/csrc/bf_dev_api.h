/* bf_dev_api.h — entry points of the DEVELOPER build only (python -m bayeformers_amd.build --dev ->
 * libbayeformers_amd_dev.so, -DBF_DEV): measured alternatives that nothing dispatches to.  Same conventions as
 * include/bayeformers_amd.h. */
#pragma once
#include "../../include/bayeformers_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same operation as bf_linear_fwd in ONE launch for LARGE M, weight-stationary: a workgroup owns (sample, strip
 * of 64 output features, a share of the rows), draws the strip's 64 x K sampled weights once into LDS — they never
 * exist in HBM — accumulates the log-probs exactly once (row share 0), and streams its rows of x against the resident
 * strip (csrc/bf_fused_ws.hip).  Needs 16-bit x / y of the compute dtype, N % 64 == 0, K % 64 == 0, K <= 768.
 * row_shares: workgroups per (sample, strip) along M (each regenerates the strip); 0 = enough to fill the chip.
 * This is the measured alternative to sampling launch + 256-wide GEMM (LABBOOK.md §4.3): it is NOT what
 * bnn.Linear dispatches to, because it is slower — four times the LDS-DMA bytes per flop of the 256 x 256 tile. */
size_t bf_linear_fwd_ws_workspace_bytes(int S, int N);
int bf_linear_fwd_ws(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                     const bf_tensor_t* bias, void* d_y, int y_dtype, int compute_dtype, int S, int M, int N, int K,
                     uint64_t seed, uint32_t sample_base, int row_shares, double* d_logprob_out, void* d_workspace,
                     size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
