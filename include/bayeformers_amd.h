/* bayeformers_amd.h — C-ABI of the MI355X (gfx950) Monte-Carlo variational forward path.
 *
 * The reference (yliess86/BayeFormers) has no FFI: its boundary is the Python class protocol.  Every entry
 * point below replaces a specific stretch of reference Python on the hot path; the Python host layer in
 * bayeformers_amd/ binds these with ctypes and keeps the reference's class/attribute surface.
 *
 * Conventions
 *   - every pointer named d_* (or documented "device") is a device pointer valid on the current HIP device;
 *   - nothing is allocated inside: outputs and workspaces are caller-provided (torch tensors on the host side);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the legacy default stream) and
 *     the call returns without synchronising;
 *   - return value 0 = ok, non-zero = error; bf_last_error() returns a thread-local description.  No C++
 *     exception crosses the boundary;
 *   - no hidden RNG state: epsilon is a pure function of (seed, sample index, stream id, element index) —
 *     see bayeformers_amd/csrc/bf_philox.h for the contract (Philox4x32-7 + Box-Muller).
 */
#ifndef BAYEFORMERS_AMD_H
#define BAYEFORMERS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_VERSION_MAJOR 0
#define BF_VERSION_MINOR 6  /* 2: bf_embed_layernorm takes the table row counts; 3: bf_sample_table_build reports the
                               tensors' effective prior kinds, bf_sample_logprob_table takes the launch's set of them;
                               4: bf_add_layernorm_bwd_sum; 5: bf_prior_t carries the device addresses its constants were
                               read from (re-checked by the kernels), bf_stale_counter; bf_linear_bwd takes d_dy_colsum,
                               bf_attention_bwd_colsum; the dropout entries take first_group; 6: the dropout entries take d_call
                               (device-resident part of the call number), bf_gemm_schedule_policy / _fetch_rows,
                               bf_profile_read_launches */

/* element types of activations / sampled weights */
enum { BF_DT_F32 = 0, BF_DT_BF16 = 1, BF_DT_F16 = 2 };

/* prior kinds.
 *   BF_PRIOR_MIXTURE  = ScaledGaussianMixture.log_prob   /root/reference/bayeformers/nn/parameters/gaussian.py:160-171
 *   BF_PRIOR_GAUSSIAN = Gaussian.log_prob used as MOPED prior  .../gaussian.py:103-116, layers/linear.py:147-150
 *   BF_PRIOR_NONE     = NoneParameter.log_prob -> 0.0     /root/reference/bayeformers/nn/parameters/base.py:68-69 */
enum { BF_PRIOR_MIXTURE = 0, BF_PRIOR_GAUSSIAN = 1, BF_PRIOR_NONE = 2 };

typedef struct bf_prior {
    int32_t kind;        /* BF_PRIOR_* */
    float pi;            /* mixture: weight of the first component */
    float sigma1;        /* mixture: std of the first component */
    float sigma2;        /* mixture: std of the second component */
    const float* d_mu;   /* gaussian: device, prior mean, same shape as the tensor */
    const float* d_rho;  /* gaussian: device, prior rho (sigma_p = softplus(rho)), same shape */
    /* gaussian, optional assertion by the caller: pi == 1 says that d_mu[i] equals the tensor's d_mu[i] and
     * softplus(d_rho[i]) equals sigma1 for every i — the MOPED prior of a frozen mean
     * (/root/reference/bayeformers/nn/layers/linear.py:147-150: prior.mu is the pretrained tensor the posterior mean shares,
     * prior.rho = ones).  bf_sample_logprob / bf_sample_logprob_table then read neither (8 instead of 16 bytes per
     * scalar); every other entry point ignores the assertion and reads d_mu / d_rho, which stay valid.  pi == 0: no claim.
     * With pi == 1, sigma2 is the constant VALUE of d_rho (softplus(sigma2) == sigma1). */
    const float* d_pi;      /* mixture, optional: the device scalars pi / sigma1 / sigma2 were read from (ScaledGaussianMixture's */
    const float* d_sigma1;  /* parameters, gaussian.py:139-141).  All three non-NULL: every kernel that evaluates the prior   */
    const float* d_sigma2;  /* re-reads them and compares with the values above, bit for bit — see bf_stale_counter.          */
    /* What the struct bakes in is a HOST copy of device state: an in-place edit of the module's tensors through `.data` (the
     * reference's own idiom, layers/linear.py:140-150) changes the device state with no host-visible trace.  So the kernels
     * re-check it: the mixture constants exactly (three scalar loads per workgroup), an asserted alias (pi == 1) on one
     * element per wave (d_mu[i] against the tensor's mean, d_rho[i] against sigma2) — a wholesale edit (fill_, copy_, mul_)
     * cannot pass, a single edited element can.  A failed check makes the log-prior of that tensor NaN for the launch and
     * increments the counter of bf_stale_counter; there is no host synchronisation. */
} bf_prior_t;

/* One Gaussian variational parameter (a weight or a bias): Gaussian(mu, rho) of
 * /root/reference/bayeformers/nn/parameters/gaussian.py:22-116. */
typedef struct bf_tensor {
    const float* d_mu;   /* device, n floats */
    const float* d_rho;  /* device, n floats */
    uint64_t n;          /* number of scalars */
    bf_prior_t prior;
    uint32_t stream_id;  /* Philox stream: 2*layer_id + (0 weight | 1 bias) */
    int32_t out_dtype;   /* BF_DT_* of d_sample_out */
    void* d_sample_out;  /* device, [S][n] of out_dtype, or NULL when only the log-probs are wanted */
} bf_tensor_t;

int bf_version(void);
const char* bf_last_error(void);

/* *h_counter = address of a host-resident (pinned, device-visible) uint32, valid for the life of the process, that the
 * kernels increment whenever a bf_prior_t's baked constants fail their device-side re-check (see bf_prior_t).  The caller
 * polls it without synchronising — a change means: drop every cached copy of prior state and describe the priors again. */
int bf_stale_counter(const uint32_t** h_counter);

/* Device-resident Monte-Carlo sample counter (optional; one per HIP device, set for / read from the calling
 * thread's current device; NULL = off, the default).  While set, every kernel launched on that device adds the
 * uint32 at d_counter to its `sample_base` argument.  This is what makes a whole step capturable in a hipGraph: the
 * graph bakes the host-side sample_base, and a captured one-element `counter += S` moves every replay on to fresh
 * epsilon.  bf_get_sample_counter returns the pointer currently set (so that a caller that points the kernels at a
 * saved counter for a backward pass can restore exactly what was there, from any thread). */
int bf_set_sample_counter(const uint32_t* d_counter);
const uint32_t* bf_get_sample_counter(void);

/* Number of compute units / name of the current device (diagnostics for bench.py). */
int bf_device_info(char* name, size_t name_len, int* n_cu, int* wave_size);

/* Host twin of the device epsilon generator: out[i] = eps(seed, sample, stream_id, element offset+i).
 * Replaces the reference's only RNG touch-point, Normal(0,1).sample(size) at gaussian.py:100. */
int bf_philox_normal_host(float* out, uint64_t n, uint64_t seed, uint32_t sample, uint32_t stream_id, uint64_t offset);

/* Device epsilon only (test hook for the RNG contract): d_out[s][i] for s in [0,S). */
int bf_philox_normal(float* d_out, uint64_t n, int S, uint64_t seed, uint32_t sample_base, uint32_t stream_id,
                     void* stream);

/* Fused reparameterise + log-prob kernel.  For each tensor t and each sample s in [0,S):
 *     eps  = philox_normal(seed, sample_base + s, t.stream_id, e)
 *     W    = mu + softplus(rho) * eps                     Gaussian.sample      gaussian.py:90-101, :81-88
 *     lq_s += sum_e log N(W; mu, softplus(rho))            Gaussian.log_prob    gaussian.py:103-116
 *     lp_s += sum_e log prior(W)                           gaussian.py:160-171 | :103-116 | base.py:68-69
 * and, when t.d_sample_out != NULL, W is written as out_dtype to d_sample_out[s][e].
 * d_logprob_out is [S][2] doubles: {log_prior, log_variational_posterior} summed over all `tensors`
 * — the two scalars Linear.forward leaves behind (layers/linear.py:99-102).  Deterministic (fixed-order) sums.
 * Workspace: bf_sample_logprob_workspace_bytes(). */
size_t bf_sample_logprob_workspace_bytes(const bf_tensor_t* tensors, int n_tensors, int S);
int bf_sample_logprob(const bf_tensor_t* tensors, int n_tensors, int S, uint64_t seed, uint32_t sample_base,
                      double* d_logprob_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* Cross-layer batching of the same kernel: ONE launch over the blocks of many tensors (e.g. the six linears of a
 * transformer layer, or a whole model), described by a device-resident table.
 *   bf_sample_table_bytes  size of the table blob for `tensors` (and the total number of 1024-scalar blocks);
 *   bf_sample_table_build  fills a HOST blob (the caller uploads it to the device once; it embeds the tensors'
 *                          mu/rho/prior/d_sample_out pointers, so rebuild when any of them moves) and
 *                          h_block_begin[n_tensors+1], the first block of each tensor;
 *   bf_sample_logprob_table  samples + log-probs of blocks [block_begin, block_end) (whole tensors) and writes one
 *                          [S][2] row of fp64 partial sums per block into d_partials[block] (total_blocks rows);
 *                          h_kinds (nullable, [n_tensors]): each tensor's EFFECTIVE prior kind — BF_PRIOR_*, or 3 for a
 *                          Gaussian prior the caller asserted to be the alias of the posterior's mean (bf_prior_t.pi == 1).
 *                          prior_kinds: bit k set = kind k occurs among the launched tensors, 0 = not known.  A launch
 *                          whose tensors all share one kind runs a kernel compiled for that kind alone (fewer registers,
 *                          more waves per SIMD); any other value selects the kernel that takes every kind;
 *   bf_reduce_logprob      d_out[g][s][{log_prior, log_q}] = fixed-order sum of partial rows [d_rows[g], d_rows[g+1]). */
size_t bf_sample_table_bytes(const bf_tensor_t* tensors, int n_tensors, uint32_t* total_blocks);
int bf_sample_table_build(const bf_tensor_t* tensors, int n_tensors, void* h_blob, size_t blob_bytes,
                          uint32_t* h_block_begin, int32_t* h_kinds);
int bf_sample_logprob_table(const void* d_blob, int n_tensors, uint32_t block_begin, uint32_t block_end, int S,
                            uint64_t seed, uint32_t sample_base, double* d_partials, int prior_kinds, void* stream);
int bf_reduce_logprob(const double* d_partials, const uint32_t* d_rows, int n_groups, int S, double* d_out,
                      void* stream);

/* Batched NT GEMM on the matrix cores:  y[s] = x[s] * w[s]^T + bias[s]   (F.linear, layers/linear.py:104)
 *   x: [S or 1][M][K] of x_dtype, sample stride x_sample_stride elements (0 = one x shared by all samples)
 *   w: [S][N][K] of w_dtype (what bf_sample_logprob wrote);  bias: [S][N] fp32 or NULL
 *   y: [S][M][N] of y_dtype.
 * compute: w_dtype BF16/F16 -> v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate (x must be w_dtype or F32);
 *          w_dtype F32 -> v_mfma_f32_16x16x4_f32 (exact fp32; x must be F32). */
int bf_gemm_nt(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
               const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, void* stream);

/* Same GEMM with an activation fused into the epilogue (applied to the fp32 accumulators after the bias):
 * BF_ACT_GELU = x/2 (1 + erf(x/sqrt 2)), the `intermediate_act_fn` that follows the dense layer of HF BERT's
 * BertIntermediate — saves one full read+write of the [S*B*L, 3072] activation per transformer layer. */
enum { BF_ACT_NONE = 0, BF_ACT_GELU = 1 };
int bf_gemm_nt_act(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                   const float* d_bias, void* d_y, int y_dtype, int S, int M, int N, int K, int act, void* stream);

/* bf_gemm_nt_act with a second output for training: d_y = act(y) and d_pre = y (the pre-activation, same dtype and
 * shape [S][M][N]), which the backward of the fused activation needs (bf_linear_bwd's d_act_pre).  One launch with two
 * stores when the 256-wide kernel takes the shape, else the GEMM followed by the elementwise activation. */
int bf_gemm_nt_act_pre(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                       const float* d_bias, void* d_y, void* d_pre, int y_dtype, int S, int M, int N, int K, int act,
                       void* stream);

/* L layers that consume the SAME activations (the query / key / value projections of an attention block,
 * HF BertSelfAttention around bnn.Linear.forward, bayeformers/nn/layers/linear.py:83-104) in ONE launch:
 *   y[l][s] = act(x[s] W_{l,s}^T + b_{l,s}),  d_w [L][S][N][K], d_bias [L][S][N] (nullable), d_y [L][S][M][N].
 * An m-panel of x is then shared by the n-tiles of all L layers while it sits in L2.  Falls back to one launch per
 * layer for shapes the 256x256 kernel does not take. */
int bf_gemm_nt_layers(const void* d_x, int x_dtype, int64_t x_sample_stride, const void* d_w, int w_dtype,
                      const float* d_bias, void* d_y, int y_dtype, int L, int S, int M, int N, int K, int act,
                      void* stream);

/* Weight-gradient GEMM of the backward pass (autograd of F.linear, layers/linear.py:104), per batch entry b:
 *   out[b][n][k] = sum_m a[b][m][n] * bm[b][m][k]        (dW = dy^T x; a = dy [Mc][N], bm = x [Mc][K])
 * Both operands are read as they lie (contraction-major), products accumulate in fp32, out is fp32 [batch][N][K].
 * dtype BF_DT_BF16 | BF_DT_F16; needs Mc % 64 == 0, N % 8 == 0, K % 8 == 0 and 16-byte aligned pointers (fails with a
 * message otherwise: bf_linear_bwd takes its transposed-copy route for such shapes). */
int bf_gemm_tn(const void* d_a, const void* d_bm, float* d_out, int dtype, int batch, int Mc, int N, int K, void* stream);

/* Input-gradient GEMM of the backward pass, per sample s:
 *   y[s][m][k] = sum_n x[s][m][n] * w[s][n][k]          (dx = dy W_s; x = dy [M][N], w = W_s [N][K] as sampled)
 * 16-bit operands and output (dtype BF_DT_BF16 | BF_DT_F16), fp32 accumulation; w is read contraction-major, so no
 * transposed copy of the sampled weights is made.  Needs N % 64 == 0, K % 8 == 0, M * K >= 128 * 128 and 16-byte
 * aligned pointers (fails with a message otherwise: bf_linear_bwd transposes W_s for such shapes). */
int bf_gemm_nn(const void* d_x, const void* d_w, void* d_y, int dtype, int S, int M, int N, int K, void* stream);
/* The same for L (1..4) layers that read the SAME activations (query / key / value): their input gradients add up,
 *   y[s][m][k] = sum_l sum_n x[l][s][m][n] * w[l][s][n][k]      (x [L][S][M][N], w [L][S][N][K] as bf_gemm_nt_layers lays them out),
 * as ONE contraction of length L*N: one launch and one output instead of L launches and L - 1 additions. */
int bf_gemm_nn_layers(const void* d_x, const void* d_w, void* d_y, int dtype, int L, int S, int M, int N, int K,
                      void* stream);
/* bf_gemm_nn with the derivative of an activation in its epilogue (ABI 6): y[s][m][k] = (sum_n x[s][m][n] w[s][n][k]) * act'(pre[s][m][k])
 * — the gradient that reaches the PRE-activation of the layer whose act(pre) was this layer's input (HF BertIntermediate ->
 * BertOutput: dpre = (dy W_down) o gelu'(pre); /root/reference/examples/bert_glue.py:239 runs it as two autograd nodes), without a
 * pass of its own over the [S][M][K] gradient.  act = BF_ACT_GELU; d_pre: [S][M][K] of `dtype`, 16-byte aligned; needs the ring
 * form of the NN kernel (bf_gemm_nn_actgrad_supported = 1: 16-bit dtype, K % 8 == 0, N >= 128 ...). */
int bf_gemm_nn_actgrad_supported(const void* d_x, const void* d_w, const void* d_y, const void* d_pre, int dtype, int S, int M,
                                 int N, int K);
int bf_gemm_nn_actgrad(const void* d_x, const void* d_w, void* d_y, const void* d_pre, int dtype, int S, int M, int N, int K,
                       int act, void* stream);

/* Build (once per device and shape) the tile schedule the 256-wide GEMM kernels run a [S][M][N] problem of L stacked
 * layers on — a small device table, allocated here.  The first bf_gemm_nt* / bf_linear_* call of a shape does the same
 * implicitly, which is an error inside a stream capture: call this for the shapes of a step before capturing it (or
 * simply run the step once).  Returns 0, or 1 with bf_last_error() set. */
int bf_gemm_prepare(int S, int L, int M, int N, void* stream);

/* The host-built tile schedule the 256-wide persistent GEMM kernel runs for a problem of S samples x L layers x
 * [M, N] outputs on n_cu compute units (introspection: the library builds and caches the same table on the first
 * launch of a shape).  The output is cut into tiles of 32 h rows (h = 1..8) x 256 columns; every workgroup b of the
 * `grid` persistent workgroups runs entries out[(j * grid + b) * 4 ..] for j = 0 .. rounds - 1 until an entry of
 * height 0.  Entry = {pair = l * S + s, s, n_tile | h << 24, first row}.  Returns the number of int32 values the
 * table holds (4 * rounds * grid); fills `out` only when cap_values is large enough. */
size_t bf_gemm_schedule(int S, int L, int M, int N, int n_cu, int32_t* out, size_t cap_values, int* rounds, int* grid);
/* The same under another schedule policy (policy < 0: the library's default; bits in csrc/bf_gemm256_dev.h) — for
 * tools/sched_l2_sim.py and the A/B of policies; and the model the library chooses a policy's column grouping by:
 * bf_gemm_schedule_fetch_rows = rows of K elements the eight per-XCD L2s fetch over the fabric for a table (workgroup b on
 * XCD b % 8, the workgroups of an XCD in k-lockstep on their j-th tiles, nothing kept between rounds), validated against
 * TCC_EA0_RDREQ in profiles/r6b_sched_l2_model.md.  Multiply by K * element size for bytes.  -1 on bad arguments. */
size_t bf_gemm_schedule_policy(int S, int L, int M, int N, int n_cu, int policy, int32_t* out, size_t cap_values, int* rounds, int* grid);
int64_t bf_gemm_schedule_fetch_rows(const int32_t* table, int rounds, int grid);

/* The whole of Linear.forward (layers/linear.py:83-104) for S Monte-Carlo samples in one call:
 * sample W_s and b_s, accumulate both log-probs, y[s] = x[s] W_s^T + b_s.
 *   weight.n must be N*K (row-major [N][K], as nn.Linear), bias may be NULL (NoneParameter, base.py:55-69);
 *   weight->d_sample_out / bias->d_sample_out / out_dtype are ignored (scratch lives in the workspace);
 *   compute_dtype: BF_DT_BF16 | BF_DT_F16 | BF_DT_F32.
 * d_logprob_out: [S][2] doubles {log_prior, log_variational_posterior}. */
size_t bf_linear_fwd_workspace_bytes(int S, int M, int N, int K, int has_bias, int compute_dtype, int x_dtype);
int bf_linear_fwd(const void* d_x, int x_dtype, int64_t x_sample_stride, const bf_tensor_t* weight,
                  const bf_tensor_t* bias, void* d_y, int y_dtype, int compute_dtype, int S, int M, int N, int K,
                  uint64_t seed, uint32_t sample_base, double* d_logprob_out, void* d_workspace,
                  size_t workspace_bytes, void* stream);

/* (The weight-stationary single-launch variant for LARGE M, bf_linear_fwd_ws, is a DEVELOPER-build entry point since
 * round 4 — csrc/bf_dev_api.h, library libbayeformers_amd_dev.so: measured 2.1-3.3x slower than sampling launch + tiled
 * GEMM at M = 4096 and slower than both alternatives at every M from 32 to 2048 (profiles/r4b_mid_m_crossover.txt), it
 * is dispatched nowhere, so the product library does not carry it.) */

/* Backward of bf_linear_fwd, reproducing the reference's autograd graph: gradients flow through
 * F.linear(input, mu + eps*softplus(rho), ...) (layers/linear.py:97,104, gaussian.py:100-101) with eps a constant and
 * the two log-prob scalars detached (linear.py:99-102: the KL terms carry no gradient in the reference).
 *   dx[s] = dy[s] W_s;  dW_s = dy[s]^T x[s];  dmu = sum_s dW_s;  drho = (sum_s dW_s*eps_s) * softplus'(rho); same for b.
 * eps is regenerated from (seed, sample_base + s, stream) — the same values the forward used; W_s is regenerated
 * too unless weight->d_sample_out still holds the forward's samples ([S][N][K] of `dtype`, weight->out_dtype ==
 * dtype), in which case they are read from there.  x, dy and dx share
 * one dtype, which is also the MFMA operand type (BF16 | F16 | F32).  Any of d_dx, d_dmu_w, d_dmu_b may be NULL
 * (not needed); d_drho_b/d_dmu_b are ignored when bias is NULL.  Gradients are written, not accumulated.
 * act / d_act_pre: when the forward fused an activation into its GEMM (bf_gemm_nt_act_pre), d_dy is the gradient of
 * act(y) and d_act_pre the forward's pre-activation y ([S][M][N] of `dtype`, 16-bit, N % 8 == 0): dy = d_dy * act'(y)
 * is formed first, in one pass that also yields the bias gradient's column sums.  act = BF_ACT_NONE: d_act_pre unused.
 * d_dy_colsum (nullable): [S][N] fp32 column sums of d_dy per sample, when the kernel that PRODUCED d_dy left them
 * (bf_attention_bwd_colsum): the bias gradient then needs no pass over d_dy of its own.
 * d_dw_keep (nullable, ABI 6): the per-sample weight gradients dW_s — [S][splits][N][K] fp32 with
 * splits = bf_linear_bwd_splits(S, M, N, K, dtype) split-K partial products per sample — are left THERE instead of in the
 * workspace and the weight's reduction over the samples is NOT done (d_dmu_w / d_drho_w untouched): the caller reduces the
 * weights of many layers with one bf_param_grad_table launch after the backward pass.
 * d_db_keep (nullable; needs a bias): likewise for the bias — its per-sample gradients, the column sums of dy ([S][N] fp32),
 * are written THERE by whichever kernel forms them and not reduced (d_dmu_b / d_drho_b untouched); when d_dy_colsum is
 * given they are already in that buffer and nothing is written (the caller copies them where its table reads). */
size_t bf_linear_bwd_workspace_bytes(int S, int M, int N, int K, int has_bias, int dtype, int act);
int bf_linear_bwd_splits(int S, int M, int N, int K, int dtype);
int bf_linear_bwd(const void* d_x, int64_t x_sample_stride, const void* d_dy, int dtype, const bf_tensor_t* weight,
                  const bf_tensor_t* bias, void* d_dx, float* d_dmu_w, float* d_drho_w, float* d_dmu_b,
                  float* d_drho_b, int S, int M, int N, int K, uint64_t seed, uint32_t sample_base, int act,
                  const void* d_act_pre, const float* d_dy_colsum, float* d_dw_keep, float* d_db_keep, void* d_workspace,
                  size_t workspace_bytes, void* stream);

/* dmu = sum_s dW_s, drho = (sum_s dW_s o eps_s) o softplus'(rho) — step 4 of bf_linear_bwd — for MANY tensors in one launch
 * (/root/reference/examples/bert_glue.py:239: loss.backward() reaches every layer's mu and rho): the per-sample gradients
 * that bf_linear_bwd calls left in their d_dw_keep buffers are reduced after the backward pass by one kernel over a device
 * table of entries (the cross-layer sampling launch's pattern, bf_sample_logprob_table).
 *   bf_param_grad_table_bytes / _build: size and contents of the HOST blob for n entries (copy it to the device once; it holds
 *   the entries' pointers); *total_blocks = grid of the launch.  bf_param_grad_table: the launch — S, seed, sample_base as the
 *   backward calls of the step had them (the device-resident sample counter, when set, is added as everywhere). */
typedef struct bf_pgrad {
    const float* d_dw;   /* device, [S][splits][n] fp32 */
    const float* d_rho;  /* device, n */
    float* d_dmu;        /* device, n, or NULL (frozen mean) */
    float* d_drho;       /* device, n */
    uint64_t n;
    uint32_t stream_id;  /* Philox stream of the tensor: 2*layer_id + (0 weight | 1 bias) */
    int32_t splits;
} bf_pgrad_t;
size_t bf_param_grad_table_bytes(const bf_pgrad_t* entries, int n, uint32_t* total_blocks);
int bf_param_grad_table_build(const bf_pgrad_t* entries, int n, void* h_blob, size_t blob_bytes);
int bf_param_grad_table(const void* d_blob, int n, uint32_t total_blocks, int S, uint64_t seed, uint32_t sample_base,
                        void* stream);

/* Opt-in Bayes-by-Backprop gradient of the KL terms.  The reference detaches its log-probs (layers/linear.py:99-102
 * store them with `.data =`), so `loss = (lvp - log_prior)/n_batches + nll` (bert_glue.py:235) trains the likelihood
 * only; this entry gives the gradient the formula implies.  For one tensor t and S samples, with
 *   L = sum_s d_g[s][0] * log_prior_s + d_g[s][1] * log_q_s,   W_s = mu + softplus(rho) * eps_s:
 *   d_dmu[e]  = sum_s g_p[s] * score(W_s[e]),                        score = d log prior / dw
 *   d_drho[e] = softplus'(rho[e]) * sum_s (g_p[s] * score(W_s[e]) * eps_s[e] - g_q[s] / sigma[e]).
 * eps is regenerated from the Philox counter.  d_dmu may be NULL.  Gradients are written, not accumulated. */
int bf_kl_grad(const bf_tensor_t* tensor, int S, uint64_t seed, uint32_t sample_base, const double* d_g,
               float* d_dmu, float* d_drho, void* stream);

/* bnn.Embedding — an EXTENSION: the north star names it, the reference has no such layer (TORCH2BAYE holds only
 * nn.Linear, bayeformers/nn/__init__.py:25), so its semantics are defined here by analogy with Linear.forward and
 * parity is unpinned.  One table draw per Monte-Carlo sample, of which only the gathered rows are materialised:
 *   d_out[t][d] = mu[id_t][d] + softplus(rho[id_t][d]) * eps(seed, sample_base + t / tokens_per_sample, stream, id_t*D + d)
 * (ids are int64, clamped to [0, V)).  The log-probs are those of the WHOLE table: bf_sample_logprob on (mu, rho) with
 * d_sample_out = NULL and the same stream id.  bf_embedding_bwd scatter-adds (atomics; d_dmu/d_drho must be zeroed by
 * the caller) dmu[id] += g, drho[id] += g * eps * softplus'(rho). */
int bf_embedding_fwd(const int64_t* d_ids, const float* d_mu, const float* d_rho, void* d_out, int out_dtype,
                     int64_t n_tokens, int64_t tokens_per_sample, int64_t V, int D, uint64_t seed, uint32_t sample_base,
                     uint32_t stream_id, void* stream);
int bf_embedding_bwd(const int64_t* d_ids, const void* d_grad, int grad_dtype, const float* d_rho, float* d_dmu,
                     float* d_drho, int64_t n_tokens, int64_t tokens_per_sample, int64_t V, int D, uint64_t seed,
                     uint32_t sample_base, uint32_t stream_id, void* stream);

/* out = LayerNorm(x + residual) * gamma + beta over the last axis (biased variance, eps inside the square root), one
 * pass over HBM: the consumer of a Bayesian dense layer's output in the transformer blocks the reference converts
 * (HF BertSelfOutput/BertOutput around bnn.Linear.forward, bayeformers/nn/layers/linear.py:83-104).  x, residual
 * (nullable: plain LayerNorm) and out are [rows, N] of `dtype`; gamma/beta are [N] of `param_dtype` (BF_DT_F32 or
 * `dtype`).  The sum and the statistics are fp32.  N % 8 == 0, N <= 8192, 16-byte aligned pointers. */
int bf_add_layernorm(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, int param_dtype,
                     void* d_out, int dtype, int64_t rows, int N, float eps, void* stream);

/* The embedding block that feeds the first Bayesian layers of a converted transformer (HF BertEmbeddings, called
 * ahead of bnn.Linear.forward, bayeformers/nn/layers/linear.py:83-104), in one pass:
 *   out[r] = LayerNorm(word[ids[r]] + type[type_ids ? type_ids[r] : 0] + pos[pos_ids ? pos_ids[r % pos_rows] : r % seq_len])
 * tables [*, N] and out [rows, N] of `dtype`; gamma/beta fp32 or `dtype`; ids int64 [rows].  Sum and statistics in fp32.
 * word_rows / type_rows / pos_table_rows = the number of rows of the three tables: an id outside its table reads nothing
 * and makes its output row NaN (torch.nn.functional.embedding asserts on the device instead).  seq_len must not exceed
 * pos_table_rows when d_pos_ids is NULL. */
int bf_embed_layernorm(const int64_t* d_ids, const int64_t* d_type_ids, const int64_t* d_pos_ids, const void* d_word,
                       const void* d_type, const void* d_pos, const void* d_gamma, const void* d_beta, int param_dtype,
                       void* d_out, int dtype, int64_t rows, int N, int seq_len, int64_t pos_rows, int64_t word_rows,
                       int64_t type_rows, int64_t pos_table_rows, float eps, void* stream);

/* Backward of bf_add_layernorm.  z = x + residual and its row statistics are recomputed from the forward's inputs;
 * d_dz [rows, N] of `dtype` is the gradient of BOTH x and residual; d_dgamma / d_dbeta are fp32 [N], written (not
 * accumulated) in a fixed summation order.  N % 8 == 0, N <= 4096. */
size_t bf_add_layernorm_bwd_workspace_bytes(int64_t rows, int N);
int bf_add_layernorm_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype, const void* d_dy,
                         void* d_dz, float* d_dgamma, float* d_dbeta, void* d_workspace, size_t workspace_bytes, int dtype,
                         int64_t rows, int N, float eps, void* stream);

/* out[b][t][h][:] = softmax_keys(q[b][t][h] . k[b][:][h] * scaling + mask[b][:]) v[b][:][h] — the attention that sits
 * between the Bayesian query/key/value projections and the Bayesian output projection of the transformers the
 * reference converts (HF BertSelfAttention around bnn.Linear.forward, bayeformers/nn/layers/linear.py:83-104; the
 * reference runs whatever the wrapped model runs there).  Inference-time forward.  q, k, v: element (b, t, h, d) at
 * ((b*T + t) * token_stride + h*head_dim + d) of `dtype` (BF16 | F16) — i.e. the [B*T, H*head_dim] outputs of the
 * projections as they are; d_mask: additive fp32 [B][T] over the keys (-inf = masked), nullable; d_out: [B][T][H][head_dim]
 * contiguous.  d_lse (nullable): [B][H][T] fp32, receives log2(sum_keys 2^(score log2 e)) per query for
 * bf_attention_bwd.  d_mask_off (nullable): one device byte; non-zero means "the mask hides nothing" and the kernel skips it —
 * lets a caller that builds the additive mask from a padding mask on the device avoid a host round trip to find out.
 * head_dim == 64, T a multiple of 128, 16-byte aligned pointers; anything else is refused (status 1). */
int bf_attention_fwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                     void* d_out, float* d_lse, int dtype, int B, int T, int H, int head_dim, int64_t token_stride,
                     float scaling, void* stream);

/* Backward of bf_attention_fwd (autograd through the attention block in the training loop,
 * /root/reference/examples/bert_glue.py:239): given the forward's inputs, its output, the gradient of the output
 * ([B][T][H][head_dim] contiguous) and the log-sum-exp rows the forward wrote (d_lse, [B][H][T] fp32; pass a buffer as
 * d_lse to bf_attention_fwd), computes dq, dk, dv ([B][T][H][head_dim] contiguous each).  Two kernels, both
 * recomputing the probabilities from q, k and d_lse: one workgroup per 128 queries walks the key tiles for dq (and
 * leaves delta[b][h][t] = sum_d dout * out in d_delta, [B][H][T] fp32 scratch), one per 128 keys walks the query
 * tiles for dk and dv.  No atomics: deterministic.  Same shape limits as the forward. */
int bf_attention_bwd(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                     const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dq, void* d_dk,
                     void* d_dv, int dtype, int B, int T, int H, int head_dim, int64_t token_stride, float scaling,
                     void* stream);

/* ---- training mode: HuggingFace dropout inside the fused kernels ------------------------------------------------------
 * The reference trains with the wrapped model in .train() (/root/reference/examples/bert_glue.py:221,227-241): HF's
 * dropout (p = 0.1) acts on the attention probabilities and on every dense output in front of a residual + LayerNorm.
 * Here a keep / drop decision is a pure function of (seed, call, site, element) — the dropout contract of
 * csrc/bf_philox.h: groups of 8 elements, one Philox4x32-7 block per group under stream 0x80000000 | site, 16-bit fields
 * compared with round(p * 65536) — so nothing but a bit per attention probability is kept for the backward pass.
 *   call = one number per forward (the Python side reserves it with the forward's sample indices, so the recomputation
 *          of a checkpointed block finds the same masks), site = the module the dropout belongs to.
 * bf_dropout_keep_host: the host twin — keep flags (0 / 1), 8 per group, of groups first_group .. first_group + n_groups.
 * first_group (the five device entries): the index of the tensor's first group in the step's GLOBAL numbering — 0 on one
 *   process; an S-sharded rank passes (first global sample of its shard) x (groups per sample), so that sample s draws the
 *   same masks whichever rank runs it.
 *
 * bf_attention_fwd_dropout: bf_attention_fwd with the probabilities dropped after the softmax normalisation.  Group of
 *   probability (b, h, q, key): g = (((b*H + h)*T + q) * (T/32) + (key/128)*4 + c) * 4 + lg, field e*4 + j, where
 *   key % 128 = (2c + e)*16 + 4 lg + j.  d_keep_bits (nullable; needed by the backward): [B][H][T][T/32] words, word
 *   (b, h, q, key/128, lg), bit c*8 + e*4 + j.
 * bf_attention_bwd_dropout: bf_attention_bwd for that forward (any supported T; d_keep_bits as the forward wrote them).
 * bf_add_layernorm_dropout: LayerNorm(dropout(x) + residual); group of element (row, n): row * (N/8) + n/8, field n % 8.
 * bf_add_layernorm_dropout_bwd: its backward; d_dz = gradient of the residual, d_dx = d_dz o keep / (1 - p).
 * d_call (nullable, ABI 6): a device-resident uint32 the kernels ADD to `call` — the device-resident part of the forward's
 *   number.  A training step captured in a hipGraph bakes the host-side `call`; the captured step copies a one-element device
 *   counter (the copy is what d_call points at, forward and backward alike) and increments it, so replay k draws the masks
 *   the k-th eager step would have drawn (the scheme of bf_set_sample_counter for epsilon). */
int bf_dropout_keep_host(uint8_t* out, uint64_t first_group, uint64_t n_groups, float p_drop, uint64_t seed, uint32_t call,
                         uint32_t site);
int bf_attention_fwd_dropout(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                             void* d_out, float* d_lse, int dtype, int B, int T, int H, int head_dim, int64_t token_stride,
                             float scaling, float p_drop, uint64_t seed, uint32_t call, uint32_t site, uint64_t first_group, uint32_t* d_keep_bits,
                             const uint32_t* d_call, void* stream);
int bf_attention_bwd_dropout(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                             const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dq, void* d_dk,
                             void* d_dv, int dtype, int B, int T, int H, int head_dim, int64_t token_stride, float scaling,
                             float p_drop, const uint32_t* d_keep_bits, void* stream);

/* bf_attention_bwd_dropout (p_drop = 0: no dropout) for sequences of ONE 128-token tile that also leaves the per-sample
 * COLUMN SUMS of dq, dk and dv as stored: d_colsum [3][samples][H * head_dim] fp32 (the B sequences being `samples` equal
 * groups), through d_partial ([B][H][3][head_dim] fp32 of scratch).  dq / dk / dv are the output gradients of the Bayesian
 * query / key / value layers: their column sums per Monte-Carlo sample are those layers' bias gradients
 * (/root/reference/examples/bert_glue.py:239), handed to bf_linear_bwd as d_dy_colsum instead of three more passes over the
 * activation-sized gradients. */
int bf_attention_bwd_colsum(const void* d_q, const void* d_k, const void* d_v, const float* d_mask, const uint8_t* d_mask_off,
                            const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dq, void* d_dk,
                            void* d_dv, int dtype, int B, int T, int H, int head_dim, int64_t token_stride, float scaling,
                            float p_drop, const uint32_t* d_keep_bits, int samples, float* d_partial, float* d_colsum,
                            void* stream);
int bf_add_layernorm_dropout(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, int param_dtype,
                             void* d_out, int dtype, int64_t rows, int N, float eps, float p_drop, uint64_t seed, uint32_t call,
                             uint32_t site, uint64_t first_group, const uint32_t* d_call, void* stream);
int bf_add_layernorm_dropout_bwd(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype,
                                 const void* d_dy, void* d_dz, void* d_dx, float* d_dgamma, float* d_dbeta, void* d_workspace,
                                 size_t workspace_bytes, int dtype, int64_t rows, int N, float eps, float p_drop, uint64_t seed,
                                 uint32_t call, uint32_t site, uint64_t first_group, const uint32_t* d_call, void* stream);

/* bf_add_layernorm_bwd / bf_add_layernorm_dropout_bwd (p_drop = 0: no dropout, d_dx unused) for an output that had TWO
 * consumers — in a transformer layer the normalised rows feed the next dense layer AND the next residual connection
 * (HF BertLayer: attention_output -> intermediate(...) and output(..., attention_output)).  Autograd would add the two
 * gradients with a pass of its own (read 2, write 1 activation-sized tensors) before this backward reads the sum; here
 * d_dy2 (nullable) is added to d_dy on load, in fp32. */
int bf_add_layernorm_bwd_sum(const void* d_x, const void* d_residual, const void* d_gamma, int param_dtype, const void* d_dy,
                             const void* d_dy2, void* d_dz, void* d_dx, float* d_dgamma, float* d_dbeta, void* d_workspace,
                             size_t workspace_bytes, int dtype, int64_t rows, int N, float eps, float p_drop, uint64_t seed,
                             uint32_t call, uint32_t site, uint64_t first_group, const uint32_t* d_call, void* stream);


/* Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * While enabled, every sampling launch (kind BF_PROF_SAMPLE) and every GEMM launch (BF_PROF_GEMM) made through
 * this library is bracketed by two events; bf_profile_read() synchronises them and returns, per kind, the number
 * of launches, the summed kernel time and the summed ALGORITHMIC work:
 *   BF_PROF_GEMM   work = 2*S*M*N*K flop (the tiled MFMA GEMM kernels);
 *   BF_PROF_FUSED_SMALL  work = 2*S*M*N*K flop (the single fused sampling+MFMA kernel of bf_linear_fwd, M <= 64);
 *   BF_PROF_SAMPLE work = bytes: (8 | 16 with a Gaussian prior) per scalar read + S * sizeof(out) per scalar written
 *                  (0 for bf_sample_logprob_table launches: the caller knows the table's totals). */
enum { BF_PROF_SAMPLE = 0, BF_PROF_GEMM = 1, BF_PROF_FUSED_SMALL = 2, BF_PROF_FUSED_WS = 3 };
int bf_profile_enable(int on);
int bf_profile_reset(void);
int bf_profile_read(int kind, uint64_t* launches, double* total_ms, double* total_work);
/* The same launch by launch, in launch order: ms[i] / work[i] of the i-th recorded launch of `kind` (up to cap entries; a
 * launch whose events could not be read has ms = -1).  Returns the number of recorded launches of that kind.  bench.py
 * derives the per-position fractions of a BERT step from it (launch i of a step is position i % 4 of layer i / 4). */
size_t bf_profile_read_launches(int kind, float* ms, double* work, size_t cap);

/* Rows per sample (M) up to which bf_linear_fwd runs an N x K layer as its single fused kernel (sampling + log-probs +
 * MFMA in one launch, no sampled weights in memory) instead of sampling launch + tiled GEMM: bf_fused_small_rows_for(N, K)
 * is the measured crossover (64 rows; 128 for layers of at most 512 x 512 weights), capped by bf_fused_small_max_rows(),
 * which the setter (0 .. 128, 0 = never) moves — a tuning knob for tools/crossover_bench.py.  Python's bnn.Linear asks
 * bf_fused_small_rows_for, so that the layers it keeps out of the cross-layer sampling plan are exactly those the
 * kernel will take. */
int bf_fused_small_max_rows(void);
int bf_set_fused_small_max_rows(int rows);
int bf_fused_small_rows_for(int N, int K);

/* Measurement utility, not part of the path: one streaming pass of 16-byte loads over `bytes` (a multiple of 16) of device
 * memory (kernel bf_probe_read_kernel).  bench.py runs it inside its rocprofv3 --pmc child passes to calibrate the L2's
 * fabric-side read counters on a known byte count and on data whose home is known (a 2 GiB buffer: HBM; a 96 MiB buffer
 * read again: the 256 MiB Infinity Cache), which is what lets roofline.traffic_detail split the GEMM's fabric fetch into
 * HBM reads and Infinity-Cache hits (rocprofv3 exposes no memory-side counter on gfx950).  d_sink: 4 writable bytes. */
int bf_probe_stream_read(const void* d_buf, size_t bytes, void* d_sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BAYEFORMERS_AMD_H */
