"""Parity of the HIP path (through the C-ABI) with the reference's golden outputs and the oracle."""
import numpy as np
import pytest
import torch

import bayeformers_amd as bf
import bayeformers_amd.nn as bnn
from bayeformers_amd import ops
from oracle import bayes_oracle as bo
from util import SEED, layer_from_case, load_case, oracle_layer, run_layer

pytestmark = pytest.mark.gpu

LOGPROB_RTOL = 2e-6      # relative to sum|terms| (== |sum| unless the signed terms cancel)
Y_FP32_RTOL = 2e-5       # exact-fp32 MFMA path vs the reference's fp32 F.linear (accumulation order differs)
CASES = ["mix_bias", "mix_nobias_oddK", "mix_custom_prior", "moped", "edge", "edge_inf"]
NO_REF_LOGPROB = {"edge", "edge_inf"}   # reference gives -inf / cancels there (documented deviations)


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(f"{golden_dir}/linear_cases.npz")


def bf16_tol(x, w_mu, w_sigma):
    """|y_bf16 - y_fp32| bound: each product carries ~2 * 2^-9 relative rounding error, summed over K."""
    scale = float(np.sqrt((x.astype(np.float64) ** 2).sum(1).max() * ((np.abs(w_mu) + 4 * w_sigma) ** 2).sum(1).max()))
    return 2.0 ** -7 * scale


@pytest.mark.parametrize("name", CASES)
def test_fp32_path_matches_reference(cases, name):
    c = load_case(cases, name)
    S, base = int(c["S"]), int(c["base"])
    bf.set_compute_dtype("fp32")
    try:
        layer = layer_from_case(c)
        y, lp = run_layer(layer, torch.from_numpy(c["x"]).cuda(), S, base)
    finally:
        bf.set_compute_dtype("bf16")
    y, lp = y.cpu().numpy(), lp.cpu().numpy()
    for s in range(S):
        scale = np.abs(c["y"][s]).max()
        np.testing.assert_allclose(y[s], c["y"][s], rtol=Y_FP32_RTOL, atol=Y_FP32_RTOL * scale)
        _, _, _, lp64, lq64, (mag_p, mag_q) = oracle_layer(c, base + s)
        assert lp[s, 0] == pytest.approx(lp64, abs=LOGPROB_RTOL * mag_p)
        assert lp[s, 1] == pytest.approx(lq64, abs=LOGPROB_RTOL * mag_q)
        if name not in NO_REF_LOGPROB:
            assert lp[s, 0] == pytest.approx(c["log_prior"][s], abs=LOGPROB_RTOL * mag_p)
            assert lp[s, 1] == pytest.approx(c["lvp"][s], abs=LOGPROB_RTOL * mag_q)
    # reference-style attribute access: 0-d fp32 means over the samples
    assert layer.log_prior.shape == () and layer.log_prior.dtype == torch.float32
    assert float(layer.log_variational_posterior) == pytest.approx(lp[:, 1].mean(), rel=1e-6)


@pytest.mark.parametrize("name", ["mix_bias", "mix_nobias_oddK", "moped"])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_low_precision_path_within_stated_tolerance(cases, name, dtype):
    c = load_case(cases, name)
    S, base = int(c["S"]), int(c["base"])
    bf.set_compute_dtype(dtype)
    try:
        layer = layer_from_case(c)
        y, lp = run_layer(layer, torch.from_numpy(c["x"]).cuda(), S, base)
    finally:
        bf.set_compute_dtype("bf16")
    sigma = np.log1p(np.exp(c["w_rho"].astype(np.float64)))
    tol = bf16_tol(c["x"], c["w_mu"], sigma) * (1.0 if dtype == "bf16" else 0.125)
    for s in range(S):
        assert np.abs(y[s].cpu().numpy() - c["y"][s]).max() < tol
        # log-probs do not depend on the MFMA precision
        assert float(lp[s, 0]) == pytest.approx(c["log_prior"][s], rel=LOGPROB_RTOL)
        assert float(lp[s, 1]) == pytest.approx(c["lvp"][s], rel=LOGPROB_RTOL)


def test_sampled_weights_match_oracle(cases):
    c = load_case(cases, "moped")
    layer = layer_from_case(c)
    outs, lp = ops.sample_logprob([layer.weight, layer.bias], [layer.weight_prior, layer.bias_prior], [0, 1], 3, SEED,
                                  100, out_dtype=torch.float32)
    for s in range(3):
        eps = bo.eps_tensor(c["w_mu"].shape, SEED, 100 + s, 0, 0)
        W = bo.gaussian_sample(torch.from_numpy(c["w_mu"]), torch.from_numpy(c["w_rho"]), eps)
        sig = bo.sigma(torch.from_numpy(c["w_rho"]))
        err = (outs[0][s].cpu() - W).abs()
        assert bool((err <= 4e-6 * sig + 1e-7 * W.abs()).all())
    bf16 = ops.sample_logprob([layer.weight], [layer.weight_prior], [0], 3, SEED, 100, out_dtype=torch.bfloat16)[0][0]
    assert torch.equal(bf16, outs[0].to(torch.bfloat16)) or \
        (bf16.float() - outs[0]).abs().max() <= outs[0].abs().max() * 2.0 ** -8


def test_batched_equals_serial_and_is_deterministic(cases):
    """Sample s of an S-batched call == the single-sample call at the same global index, bit for bit."""
    c = load_case(cases, "mix_bias")
    x = torch.from_numpy(c["x"]).cuda()
    layer = layer_from_case(c)
    y, lp = run_layer(layer, x, 4, 20)
    y2, lp2 = run_layer(layer, x, 4, 20)
    assert torch.equal(y, y2) and torch.equal(lp, lp2)
    for s in range(4):
        ys, lps = run_layer(layer, x, 1, 20 + s)
        assert torch.equal(ys[0], y[s]) and torch.equal(lps[0], lp[s])
    # bare layer (no bnn.Model): reference-style call, scalar attributes
    bf.manual_seed(SEED, next_sample=21)
    with torch.no_grad():
        yb = layer(x)
    assert torch.equal(yb, y[1]) and float(layer.log_prior) == pytest.approx(float(lp[1, 0]), rel=1e-6)


GEMM_SHAPES = [(1, 1, 1, 8), (2, 5, 3, 8), (1, 128, 128, 32), (3, 130, 70, 72), (2, 33, 129, 40), (1, 257, 2, 768),
               (2, 64, 10, 512), (1, 17, 9, 33), (2, 31, 65, 100), (1, 300, 256, 264),
               # K % 64 == 0 and M*N >= 128*128: the 256x256x64 LDS-DMA kernel, with ragged M and N edges
               (1, 256, 256, 64), (2, 300, 200, 128), (3, 513, 259, 192), (1, 1000, 130, 768), (2, 129, 1030, 64),
               # one k-step per tile and two tiles per workgroup: the tile-to-tile hand-over with nothing in between
               (12, 4096, 512, 64)]


@pytest.mark.parametrize("S,M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("wdt", [torch.bfloat16, torch.float16, torch.float32])
def test_gemm_nt_against_torch(S, M, N, K, wdt):
    g = torch.Generator(device="cuda").manual_seed(S * 1000003 + M * 1009 + N * 31 + K)
    w = torch.randn(S, N, K, device="cuda", generator=g)
    x = torch.randn(S, M, K, device="cuda", generator=g)
    bias = torch.randn(S, N, device="cuda", generator=g)
    wq = w.to(wdt)
    for xdt in ({wdt, torch.float32}):
        xq = x.to(xdt)
        ref = torch.einsum("smk,snk->smn", xq.to(wdt).double(), wq.double()) + bias[:, None, :].double()
        for ydt in ({wdt, torch.float32}):
            y = ops.gemm_nt(xq, wq, bias, S, M, N, K, M * K, ydt)
            tol = {torch.float32: 1e-5, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[ydt]
            err = (y.double() - ref).abs().max().item()
            assert err <= tol * ref.abs().max().item() + 1e-5 * np.sqrt(K), (xdt, ydt, err)
    # shared x (sample stride 0) and no bias
    y = ops.gemm_nt(x[0].to(wdt).contiguous(), wq, None, S, M, N, K, 0, torch.float32)
    ref = torch.einsum("mk,snk->smn", x[0].to(wdt).double(), wq.double())
    assert (y.double() - ref).abs().max().item() <= 1e-5 * (ref.abs().max().item() + np.sqrt(K))


@pytest.mark.parametrize("S,L,M,N,K,act", [(10, 3, 4096, 768, 768, 0), (10, 1, 4096, 3072, 768, 1), (12, 1, 4096, 512, 64, 0),
                                            (3, 1, 1000, 520, 192, 1)])
def test_gemm_forms_are_repeatable(S, L, M, N, K, act):
    """The persistent kernel hands LDS buffers from one tile to the next without a workgroup barrier in between and its
    epilogue runs per wave: a missing dependency would show as run-to-run differences.  Every form, many launches,
    bit for bit."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(S, M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(L, S, N, K, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(L, S, N, device="cuda", generator=g)
    ref = ops.gemm_nt_layers(x, w, b, L, S, M, N, K, M * K, torch.bfloat16, act).clone()
    for _ in range(40):
        assert torch.equal(ops.gemm_nt_layers(x, w, b, L, S, M, N, K, M * K, torch.bfloat16, act), ref)
    a = torch.randn(S, M, N, device="cuda", generator=g).bfloat16()
    if M % 64 == 0 and N % 8 == 0 and K % 8 == 0:
        r = ops.gemm_tn(a, x).clone()
        for _ in range(20):
            assert torch.equal(ops.gemm_tn(a, x), r)
    if N % 64 == 0 and K % 8 == 0:
        w1 = w[0].contiguous()
        r = ops.gemm_nn(a, w1).clone()
        for _ in range(20):
            assert torch.equal(ops.gemm_nn(a, w1), r)


# The launches of the benchmarked steps (BASELINE configs[2] and configs[4]), EVERY output row against an fp64 einsum of the
# same 16-bit operands: (S, L, M, N, K, act, dtype).  These shapes have more tiles than workgroups (480 ... 1920 tiles on 256
# persistent workgroups), so the ring's tile-to-tile hand-over (the next tile's W(0) / X(0) / W(1) issued in a tile's last
# two k-steps, csrc/bf_gemm256_r5.hip) is exercised on every row — F.linear of /root/reference/bayeformers/nn/layers/linear.py:104.
BENCH_NT_SHAPES = [
    (10, 1, 4096, 768, 768, 0, torch.bfloat16),     # BERT-base attention-out
    (10, 1, 4096, 768, 3072, 0, torch.bfloat16),    # BERT-base FFN-down
    (10, 1, 4096, 3072, 768, 1, torch.bfloat16),    # BERT-base FFN-up + GELU
    (10, 3, 4096, 768, 768, 0, torch.bfloat16),     # BERT-base query / key / value, one stacked launch
    (10, 1, 6144, 1024, 1024, 0, torch.float16),    # BERT-large attention-out (configs[4], fp16)
    (10, 1, 6144, 4096, 1024, 1, torch.float16),    # BERT-large FFN-up + GELU
    (10, 1, 6144, 1024, 4096, 0, torch.float16),    # BERT-large FFN-down
    (10, 3, 6144, 1024, 1024, 0, torch.float16),    # BERT-large query / key / value
    # the reference's own precision (--dtype fp32): the same ring on v_mfma_f32_16x16x4_f32, fp32 operands and outputs
    (10, 1, 4096, 768, 768, 0, torch.float32), (10, 1, 4096, 768, 3072, 0, torch.float32),
    (10, 1, 4096, 3072, 768, 1, torch.float32), (10, 3, 4096, 768, 768, 0, torch.float32),
    (3, 1, 1000, 520, 96, 1, torch.float32),        # ragged M and N, three k-steps
]


@pytest.mark.parametrize("S,L,M,N,K,act,dt", BENCH_NT_SHAPES)
def test_gemm_nt_benchmarked_shapes_all_rows_against_fp64(S, L, M, N, K, act, dt):
    g = torch.Generator(device="cuda").manual_seed(S * 7919 + L * 131 + M * 31 + N * 7 + K)
    x = torch.randn(S, M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(L, S, N, K, device="cuda", generator=g) * 0.1).to(dt)
    b = torch.randn(L, S, N, device="cuda", generator=g)
    y = ops.gemm_nt_layers(x, w, b, L, S, M, N, K, M * K, dt, act)
    assert y.shape == (L, S, M, N) and y.dtype == dt
    tol = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 1e-5}[dt]
    xd = x.double()
    worst = 0.0
    for l in range(L):  # one layer at a time: the fp64 reference of a whole launch is 1-3 GB
        ref = torch.einsum("smk,snk->smn", xd, w[l].double()) + b[l][:, None, :].double()
        if act:
            ref = torch.nn.functional.gelu(ref)
        bound = tol * ref.abs().max().item() + 1e-5 * np.sqrt(K)
        err = (y[l].double() - ref).abs().max().item()
        worst = max(worst, err / bound)
        assert err <= bound, (l, err, bound)
        del ref
    if L == 1:  # the single-layer entry point runs the same schedule: same bits
        assert torch.equal(ops.gemm_nt(x, w[0], b[0], S, M, N, K, M * K, dt, act), y[0])
    print(f"[gemm_nt all rows] S={S} L={L} M={M} N={N} K={K} act={act} {str(dt)[6:]}: max err / bound = {worst:.3f}")


def test_gemm_fuzz_all_forms_against_fp64():
    """tools/gemm_fuzz.py as a test: 65 random ragged shapes (M 130..3000, N = 8..1112, K = 64..512, bf16 / fp16, with
    and without GELU) through every form of the 256-wide GEMM — NT, NT with the pre-activation output, TN, NN — against
    fp64 einsums over all outputs."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gemm_fuzz.py")
    spec = importlib.util.spec_from_file_location("gemm_fuzz", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run(65, 0) == 0


def test_gemm_detects_transposes():
    """Asymmetric operands: a swapped row/col mapping in the MFMA epilogue cannot pass."""
    M, N, K = 48, 80, 64
    x = torch.zeros(1, M, K, device="cuda")
    w = torch.zeros(1, N, K, device="cuda")
    x[0, :, 0] = torch.arange(M, device="cuda").float() + 1       # y[m, n] = (m + 1) * (2n + 1)
    w[0, :, 0] = 2 * torch.arange(N, device="cuda").float() + 1
    y = ops.gemm_nt(x.bfloat16(), w.bfloat16(), None, 1, M, N, K, M * K, torch.float32)[0]
    ref = torch.outer(torch.arange(M).float() + 1, 2 * torch.arange(N).float() + 1).cuda()
    assert torch.equal(y, ref)


@pytest.mark.parametrize("batch,Mc,N,K", [(1, 64, 8, 8), (2, 128, 256, 256), (3, 192, 200, 136), (1, 1024, 776, 264),
                                          (2, 64, 1032, 40), (20, 2048, 768, 768), (1, 320, 24, 520), (40, 64, 768, 768),
                                          # more tiles than workgroups: the unit ring carries on across tile boundaries
                                          # (2 / 3 k-steps: no steady-state step; ragged edges; the benchmarked shape)
                                          (40, 128, 768, 768), (36, 192, 768, 1024), (30, 256, 776, 520),
                                          (10, 4096, 3072, 768)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_tn_against_torch(batch, Mc, N, K, dt):
    """bf_gemm_tn (dW = dy^T x, operands read contraction-major through the LDS transpose read) against an fp64 einsum
    of the same 16-bit operands: exact products, fp32 accumulation."""
    g = torch.Generator(device="cuda").manual_seed(batch * 7919 + Mc * 31 + N * 7 + K)
    a = torch.randn(batch, Mc, N, device="cuda", generator=g).to(dt)
    b = torch.randn(batch, Mc, K, device="cuda", generator=g).to(dt)
    out = ops.gemm_tn(a, b)
    ref = torch.einsum("bmn,bmk->bnk", a.double(), b.double())
    assert out.shape == (batch, N, K) and out.dtype == torch.float32
    assert (out.double() - ref).abs().max().item() <= 2e-6 * np.sqrt(Mc) * ref.abs().max().item()


@pytest.mark.parametrize("S,M,N,K", [(1, 128, 64, 128), (2, 300, 128, 200), (3, 513, 192, 264), (10, 4096, 768, 768),
                                      (1, 1000, 3072, 776), (2, 129, 64, 1032), (12, 4096, 64, 512)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_nn_against_torch(S, M, N, K, dt):
    """bf_gemm_nn (dx = dy W_s, W_s read contraction-major through the LDS transpose read, dy K-contiguous) against an
    fp64 einsum of the same 16-bit operands."""
    g = torch.Generator(device="cuda").manual_seed(S * 7919 + M * 31 + N * 7 + K)
    x = torch.randn(S, M, N, device="cuda", generator=g).to(dt)
    w = (torch.randn(S, N, K, device="cuda", generator=g) * 0.1).to(dt)
    y = ops.gemm_nn(x, w)
    ref = torch.einsum("smn,snk->smk", x.double(), w.double())
    tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert y.shape == (S, M, K) and y.dtype == dt
    assert (y.double() - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-5 * np.sqrt(N)


@pytest.mark.parametrize("L,S,M,N,K", [(3, 2, 300, 128, 200), (2, 1, 513, 64, 264), (3, 10, 4096, 768, 768), (4, 2, 1000, 192, 72)])
def test_gemm_nn_layers_against_torch(L, S, M, N, K):
    """bf_gemm_nn_layers: the contraction over L stacked layers (dx = sum_l dy_l W_l) against an fp64 einsum."""
    g = torch.Generator(device="cuda").manual_seed(L * 7919 + M * 31 + N * 7 + K)
    x = torch.randn(L, S, M, N, device="cuda", generator=g).bfloat16()
    w = (torch.randn(L, S, N, K, device="cuda", generator=g) * 0.1).bfloat16()
    y = ops.gemm_nn_layers(x, w)
    ref = torch.einsum("lsmn,lsnk->smk", x.double(), w.double())
    assert (y.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-5 * np.sqrt(L * N)
    one = ops.gemm_nn_layers(x[:1].contiguous(), w[:1].contiguous())
    assert torch.equal(one, ops.gemm_nn(x[0], w[0]))


def test_gemm_nn_detects_transposes():
    S, M, N, K = 1, 128, 64, 256
    x = torch.zeros(S, M, N, device="cuda")
    w = torch.zeros(S, N, K, device="cuda")
    x[0, :, 0] = torch.arange(M, device="cuda").float() % 16 + 1          # y[m][k] = (m % 16 + 1) * (k % 8 + 1)
    w[0, 0, :] = torch.arange(K, device="cuda").float() % 8 + 1
    x[0, :, 37] = 1.0                                                     # + w[37][k] = k % 3
    w[0, 37, :] = torch.arange(K, device="cuda").float() % 3
    y = ops.gemm_nn(x.bfloat16(), w.bfloat16())[0].float()
    ref = torch.outer(torch.arange(M).float() % 16 + 1, torch.arange(K).float() % 8 + 1) + (torch.arange(K).float() % 3)[None, :]
    assert torch.equal(y, ref.cuda())


def test_gemm_tn_detects_transposes_and_rejects_shapes():
    """Asymmetric operands (a swapped n/k mapping or a permuted contraction index on ONE side cannot pass), and the
    shapes the kernel does not take fail with a message instead of computing something else."""
    Mc, N, K = 128, 48, 80
    a = torch.zeros(1, Mc, N, device="cuda")
    b = torch.zeros(1, Mc, K, device="cuda")
    m = torch.arange(Mc, device="cuda").float()
    a[0] = (m[:, None] == torch.arange(N, device="cuda")[None, :] * 2).float()        # a[m][n] = [m == 2n]
    b[0] = m[:, None] + torch.arange(K, device="cuda").float()[None, :]               # out[n][k] = 2 n + k (bf16-exact)
    out = ops.gemm_tn(a.bfloat16(), b.bfloat16())[0]
    ref = (2 * torch.arange(N).float()[:, None] + torch.arange(K).float()[None, :]).cuda()
    assert torch.equal(out, ref)
    for shape in [(1, 96, 8, 8), (1, 64, 12, 8), (1, 64, 8, 20)]:
        with pytest.raises(bf._C.BayeFormersAMDError, match="bf_gemm_tn"):
            ops.gemm_tn(torch.zeros(shape[0], shape[1], shape[2], device="cuda").bfloat16(),
                        torch.zeros(shape[0], shape[1], shape[3], device="cuda").bfloat16())


@pytest.mark.parametrize("kind", ["mixture", "moped"])
@pytest.mark.parametrize("M", [32, 4096])
def test_full_size_layer_768(golden_dir, kind, M):
    """BASELINE config 2 at full size against the REAL reference's outputs (tests/golden/linear768_c2.npz):
    bnn.Linear(768, 768), S = 10, default init + mixture prior and the MOPED variant (delta = 0.05, freeze=True: the
    aliased-prior fast path of the sampling kernel), x ~ N(0,1) [M, 768].  Every sample's W is compared with the oracle's
    Gaussian.sample, the outputs with the reference's rows, the log-probs with the reference's values."""
    from util import linear768_input, linear768_layer, module_checksum

    g = np.load(f"{golden_dir}/linear768_c2.npz")
    S, N, K = int(g["S"]), 768, 768
    base = int(g[f"{kind}/base"])
    layer = linear768_layer(kind)
    # (the MOPED rho = log(exp(delta |w|) - 1) is rebuilt with THIS host's libm: an ulp here and there against the build
    # container's, 2e-8 of the checksum — far inside every tolerance below)
    assert module_checksum(layer) == pytest.approx(float(g[f"{kind}/checksum"]), rel=1e-6)
    layer = layer.cuda()
    layer.layer_id = 0
    if kind == "moped":  # a frozen mean under its MOPED prior: the kernel reads 8 instead of 16 bytes per scalar
        assert ops.prior_alias(layer.weight, layer.weight_prior) == pytest.approx(float(np.log1p(np.exp(np.float32(1.0)))), rel=1e-6)
        assert ops.prior_alias(layer.bias, layer.bias_prior) is not None
    x = linear768_input(M)
    rows = g[f"rows_{M}"]
    y_ref = g[f"{kind}/y{M}"]                      # [S, len(rows), 768] from the reference
    mu_w, rho_w = layer.weight.mu.detach().cpu(), layer.weight.rho.detach().cpu()
    sig = bo.sigma(rho_w)
    y, lp = run_layer(layer, x.cuda(), S, base)     # bf16 MFMA path (M = 32: the single fused kernel)
    bf.set_compute_dtype("fp32")
    try:
        y32, lp32 = run_layer(layer, x.cuda(), S, base)
    finally:
        bf.set_compute_dtype("bf16")
    Ws = ops.sample_logprob([layer.weight], [layer.weight_prior], [0], S, SEED, base, out_dtype=torch.float32)[0][0].cpu()
    tol16 = 2.0 ** -7 * float(x.norm(dim=1).max()) * float((mu_w.abs() + 4 * sig).norm(dim=1).max())
    for s in range(S):
        assert float(lp[s, 0]) == pytest.approx(g[f"{kind}/log_prior"][s], rel=LOGPROB_RTOL)
        assert float(lp[s, 1]) == pytest.approx(g[f"{kind}/lvp"][s], rel=LOGPROB_RTOL)
        assert float(lp32[s, 0]) == pytest.approx(g[f"{kind}/log_prior"][s], rel=LOGPROB_RTOL)
        assert float(lp32[s, 1]) == pytest.approx(g[f"{kind}/lvp"][s], rel=LOGPROB_RTOL)
        W = bo.gaussian_sample(mu_w, rho_w, bo.eps_tensor((N, K), SEED, base + s, 0, 0))
        assert bool(((Ws[s] - W).abs() <= 4e-6 * sig + 1e-7 * W.abs()).all())          # EVERY sample's weights
        ref = torch.from_numpy(y_ref[s])
        scale = float(ref.abs().max())
        assert (y32[s].cpu()[rows] - ref).abs().max().item() <= Y_FP32_RTOL * scale     # exact-fp32 path vs the reference
        assert (y[s].float().cpu()[rows] - ref).abs().max().item() < tol16              # bf16 path, stated tolerance


def test_moped_alias_path_equals_general_path_and_tracks_edits(golden_dir):
    """The aliased MOPED prior (frozen mean, constant prior sigma: /root/reference/bayeformers/nn/layers/linear.py:147-150)
    gives the log-probs of the general Gaussian-prior path; a trainable mean is never aliased; an in-place edit of the prior
    ends the alias and the kernel follows the edited prior."""
    from util import linear768_layer

    g = np.load(f"{golden_dir}/linear768_c2.npz")
    S, base = 4, int(g["moped/base"])
    layer = linear768_layer("moped").cuda()
    layer.layer_id = 0
    gs, prs = [layer.weight, layer.bias], [layer.weight_prior, layer.bias_prior]
    lp_alias = ops.sample_logprob(gs, prs, [0, 1], S, SEED, base, out_dtype=None)[1].cpu()
    assert ops.prior_alias(layer.weight, layer.weight_prior) is not None
    layer.weight.mu.requires_grad_(True)
    layer.bias.mu.requires_grad_(True)
    assert ops.prior_alias(layer.weight, layer.weight_prior) is None
    lp_general = ops.sample_logprob(gs, prs, [0, 1], S, SEED, base, out_dtype=None)[1].cpu()
    for s in range(S):
        assert float(lp_alias[s, 0]) == pytest.approx(float(lp_general[s, 0]), rel=LOGPROB_RTOL)
        assert float(lp_alias[s, 0]) == pytest.approx(g["moped/log_prior"][s], rel=LOGPROB_RTOL)
        assert float(lp_alias[s, 1]) == float(lp_general[s, 1])
    # edited prior: no longer one constant sigma
    layer.weight.mu.requires_grad_(False)
    layer.bias.mu.requires_grad_(False)
    assert ops.prior_alias(layer.weight, layer.weight_prior) is not None
    with torch.no_grad():
        layer.weight_prior.rho[5, 7] = 3.0
        layer.weight_prior.mu[2, 3] += 0.25
    assert ops.prior_alias(layer.weight, layer.weight_prior) is None
    lp_edit = ops.sample_logprob(gs, prs, [0, 1], 1, SEED, base, out_dtype=None)[1].cpu()
    t = lambda p: p.detach().cpu()
    eps_w, eps_b = bo.eps_tensor((768, 768), SEED, base, 0, 0), bo.eps_tensor((768,), SEED, base, 0, 1)
    lp64, lq64 = bo.linear_logprobs_f64(t(layer.weight.mu), t(layer.weight.rho), t(layer.bias.mu), t(layer.bias.rho), eps_w,
                                        eps_b, ("gaussian", t(layer.weight_prior.mu), t(layer.weight_prior.rho)),
                                        ("gaussian", t(layer.bias_prior.mu), t(layer.bias_prior.rho)))
    assert float(lp_edit[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL) and float(lp_edit[0, 0]) != float(lp_alias[0, 0])
    assert float(lp_edit[0, 1]) == pytest.approx(lq64, rel=LOGPROB_RTOL)


def _logprob_oracle(layer, prior_w, prior_b, sample):
    t = lambda p: p.detach().cpu()
    N, K = layer.weight.mu.shape
    eps_w, eps_b = bo.eps_tensor((N, K), SEED, sample, 0, 0), bo.eps_tensor((N,), SEED, sample, 0, 1)
    return bo.linear_logprobs_f64(t(layer.weight.mu), t(layer.weight.rho), t(layer.bias.mu), t(layer.bias.rho), eps_w, eps_b,
                                  prior_w, prior_b)


@pytest.mark.parametrize("M", [32, 300])   # the single fused kernel and the sampling launch + tiled GEMM
def test_prior_edited_through_data_is_never_silently_stale(M):
    """The reference's own idiom for touching a prior is an in-place edit through `.data`
    (/root/reference/bayeformers/nn/layers/linear.py:140-150), which moves no version counter — the MOPED-alias verdict
    and the mixture constants cached on the host cannot see it.  The kernels re-check what they were told against the
    tensors (bf_prior_t): the forward after such an edit has a NaN log_prior (never a wrong finite one), the next one
    warns, drops the caches and is right; `bayeformers_amd.invalidate_caches` right after the edit skips the NaN step."""
    import warnings

    from util import linear768_layer

    x = torch.randn(M, 768, device="cuda")
    t = lambda p: p.detach().cpu()

    def gauss_priors(layer):
        return (("gaussian", t(layer.weight_prior.mu), t(layer.weight_prior.rho)),
                ("gaussian", t(layer.bias_prior.mu), t(layer.bias_prior.rho)))

    # --- MOPED prior of a frozen mean (aliased: the kernel reads neither prior.mu nor prior.rho)
    layer = linear768_layer("moped").cuda()
    layer.layer_id = 0
    _, lp = run_layer(layer, x, 2, 50)
    lp64, _ = _logprob_oracle(layer, *gauss_priors(layer), 50)
    assert float(lp[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL)
    layer.weight_prior.rho.data.fill_(0.5)
    if M > ops.fused_small_rows(768, 768):
        _, lp = run_layer(layer, x, 2, 50)
        assert bool(torch.isnan(lp[:, 0]).all()) and bool(torch.isfinite(lp[:, 1]).all())     # loud, not wrong
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            _, lp = run_layer(layer, x, 2, 50)
        assert any("edited in place" in str(i.message) for i in w)
    else:  # the single fused kernel reads a Gaussian prior's tensors themselves: nothing cached, right at once
        _, lp = run_layer(layer, x, 2, 50)
    lp64, lq64 = _logprob_oracle(layer, *gauss_priors(layer), 50)
    assert float(lp[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL) and float(lp[0, 1]) == pytest.approx(lq64, rel=LOGPROB_RTOL)
    # the public way: say so after the edit — no NaN step (a scaled prior mean: no longer the posterior's)
    layer.weight_prior.mu.data.mul_(1.5)
    layer.bias_prior.rho.data.fill_(-1.0)
    bf.invalidate_caches(layer)
    _, lp = run_layer(layer, x, 2, 50)
    lp64, _ = _logprob_oracle(layer, *gauss_priors(layer), 50)
    assert float(lp[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL)

    # --- scale-mixture prior: its three constants live in 0-d device tensors and are cached on the host
    torch.manual_seed(5)
    prior = bnn.ScaledGaussianMixture(0.5, 1.0, float(np.exp(-6)))
    layer = bnn.Linear(768, 768, prior=prior).cuda()
    layer.layer_id = 0
    mix = lambda: ("mixture", float(prior.pi), float(prior.sigma1), float(prior.sigma2))
    _, lp = run_layer(layer, x, 2, 60)
    lp64, _ = _logprob_oracle(layer, mix(), mix(), 60)
    assert float(lp[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL)
    prior.sigma1.data.fill_(2.0)
    prior.pi.data.fill_(0.25)
    _, lp = run_layer(layer, x, 2, 60)
    assert bool(torch.isnan(lp[:, 0]).all())
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, lp = run_layer(layer, x, 2, 60)
    assert any("edited in place" in str(i.message) for i in w)
    lp64, _ = _logprob_oracle(layer, mix(), mix(), 60)
    assert float(lp[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL)
    prior.sigma2.data.fill_(0.01)
    bf.invalidate_caches(layer)
    _, lp = run_layer(layer, x, 2, 60)
    lp64, _ = _logprob_oracle(layer, mix(), mix(), 60)
    assert float(lp[0, 0]) == pytest.approx(lp64, rel=LOGPROB_RTOL)


@pytest.mark.parametrize("S,M,N,K", [(2, 300, 200, 128), (1, 40, 24, 72), (2, 513, 259, 192)])
def test_gemm_fused_gelu(S, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(7)
    w = torch.randn(S, N, K, device="cuda", generator=g).bfloat16()
    x = torch.randn(S, M, K, device="cuda", generator=g).bfloat16()
    bias = torch.randn(S, N, device="cuda", generator=g)
    pre = torch.einsum("smk,snk->smn", x.double(), w.double()) + bias[:, None, :].double()
    ref = torch.nn.functional.gelu(pre)
    for ydt in (torch.bfloat16, torch.float32):
        y = ops.gemm_nt(x, w, bias, S, M, N, K, M * K, ydt, act=1)
        tol = 2.0 ** -8 if ydt == torch.bfloat16 else 1e-5
        assert (y.double() - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-5 * np.sqrt(K)


def test_empty_and_ragged_inputs(cases):
    c = load_case(cases, "mix_bias")
    layer = layer_from_case(c)
    bf.manual_seed(SEED, next_sample=0)
    with torch.no_grad():
        y = layer(torch.empty(0, 40, device="cuda"))
    assert y.shape == (0, 24)
    _, _, _, lp64, lq64, (mag_p, mag_q) = oracle_layer(c, 0)
    assert float(layer.log_prior) == pytest.approx(lp64, abs=1e-5 * mag_p)      # fp32 attribute
    assert float(layer.log_variational_posterior) == pytest.approx(lq64, abs=1e-5 * mag_q)
    # rows not divisible by the sample count
    model = bnn.Model(layer)
    with torch.no_grad(), model.monte_carlo(3), pytest.raises(bf._C.BayeFormersAMDError, match="multiple of the sample count"):
        model(torch.randn(7, 40, device="cuda"))
    # 3-D input, batch of one row
    with torch.no_grad():
        y = layer(torch.randn(1, 1, 40, device="cuda"))
    assert y.shape == (1, 1, 24)


def test_many_samples_in_one_call(cases):
    """S = 64 (config 4's sample count) in one launch: every sample equals its serial counterpart."""
    c = load_case(cases, "moped")
    x = torch.from_numpy(c["x"]).cuda()
    layer = layer_from_case(c)
    y, lp = run_layer(layer, x, 64, 1000)
    for s in (0, 17, 63):
        ys, lps = run_layer(layer, x, 1, 1000 + s)
        assert torch.equal(ys[0], y[s]) and torch.equal(lps[0], lp[s])
    _, _, _, lp64, lq64, (mag_p, mag_q) = oracle_layer(c, 1063)
    assert float(lp[63, 0]) == pytest.approx(lp64, abs=LOGPROB_RTOL * mag_p)
    assert float(lp[63, 1]) == pytest.approx(lq64, abs=LOGPROB_RTOL * mag_q)


def test_sample_counter_wraps_modulo_2_32():
    """Monte-Carlo sample indices are 32-bit Philox counter words: a forward that straddles 2^32 uses
    ..., 2^32 - 1, 0, 1, ... — checked against the oracle's epsilon at the wrapped indices."""
    from oracle import bayes_oracle as bo

    layer = bnn.Linear(64, 48).cuda()
    layer.layer_id = 0
    model = bnn.Model(layer)
    S, B = 8, 4
    x = torch.randn(S * B, 64, device="cuda")
    base = 2 ** 32 - 3
    bf.manual_seed(SEED, next_sample=base)
    bf.set_compute_dtype("fp32")
    try:
        with torch.no_grad(), model.monte_carlo(S):
            y = model(x)
    finally:
        bf.set_compute_dtype("bf16")
    lp = model.log_prob_samples().cpu().numpy()
    mu, rho = layer.weight.mu.detach().cpu(), layer.weight.rho.detach().cpu()
    bmu, brho = layer.bias.mu.detach().cpu(), layer.bias.rho.detach().cpu()
    for s in range(S):
        idx = (base + s) & 0xFFFFFFFF
        ew, eb = bo.eps_tensor(mu.shape, SEED, idx, 0, 0), bo.eps_tensor(bmu.shape, SEED, idx, 0, 1)
        w = mu + torch.nn.functional.softplus(rho) * ew
        b = bmu + torch.nn.functional.softplus(brho) * eb
        ref = torch.nn.functional.linear(x[s * B:(s + 1) * B].cpu(), w, b)
        assert (y[s * B:(s + 1) * B].cpu() - ref).abs().max().item() < 1e-4
        lq = bo.gaussian_log_prob_f64(ew, mu, rho) + bo.gaussian_log_prob_f64(eb, bmu, brho)
        assert abs(lp[s, 1] / lq - 1) < 1e-6
    assert bf.random.get_state()[1] == (base + S) & 0xFFFFFFFF


@pytest.mark.parametrize("S,M,N,K", [(2, 300, 200, 128), (3, 1000, 776, 192), (1, 513, 259 * 8 // 8 * 8, 256), (10, 4096, 768, 768),
                                     (2, 4096, 2304, 768), (2, 2048, 768, 3072)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_ring_kernel_is_bit_identical_to_the_burst_kernel(S, M, N, K, dt):
    """The five-slot-ring forward kernel (16-bit outputs, csrc/bf_gemm256_r5.hip) and the burst kernel (which still serves
    fp32 outputs, csrc/bf_gemm256.hip) run the same MFMA sequence: the ring's 16-bit result must be the burst kernel's
    fp32 result rounded once — bit for bit, ragged tile edges included.  A wrong ring slot, a DMA piece read before it
    landed or after it was overwritten would show here."""
    g = torch.Generator(device="cuda").manual_seed(S * 1000 + M + N + K)
    x = torch.randn(S, M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(S, N, K, device="cuda", generator=g) * 0.1).to(dt)
    b = torch.randn(S, N, device="cuda", generator=g)
    y32 = ops.gemm_nt(x, w, b, S, M, N, K, M * K, torch.float32, 0)
    for _ in range(3):  # a race would not fail every time
        y16 = ops.gemm_nt(x, w, b, S, M, N, K, M * K, dt, 0)
        assert torch.equal(y16, y32.to(dt))
    # one x shared by all samples (sample stride 0), no bias
    x0 = x[0].contiguous()
    assert torch.equal(ops.gemm_nt(x0, w, None, S, M, N, K, 0, dt, 0), ops.gemm_nt(x0, w, None, S, M, N, K, 0, torch.float32, 0).to(dt))
    # and against fp64 on a slice (both kernels could be wrong together)
    ref = torch.einsum("mk,nk->mn", x[S - 1].double(), w[S - 1].double()) + b[S - 1].double()
    tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert (y16[S - 1].double() - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-5 * K ** 0.5


@pytest.mark.parametrize("S,M,N,K,what", [
    (3, (1 << 19) + 300, 256, 1024, "ring kernel, samples 1 and 2 start beyond 2^30 / 2^31 bytes of x"),
    (1, (1 << 20) + 300, 256, 1024, "one sample of x >= 2^30 elements: beyond the ring kernel's 32-bit byte offsets -> burst kernel"),
    (2, (1 << 17) + 44, 8192, 128, "outputs: a sample of y is 2^31 bytes, the second starts beyond 2^32 bytes"),
    (1, (1 << 18) + 8, 8192, 128, "a sample of y >= 2^31 elements: beyond the tiled kernels' output index -> generic kernel"),
])
def test_operands_across_the_32_bit_boundaries(S, M, N, K, what):
    """Offsets inside one sample are 32-bit in the tiled kernels (byte offsets of the DMA pieces, element indices of the
    output rows), sample bases are 64-bit: operands whose extents straddle 2^30 / 2^31 / 2^32 must come out right in their
    first and last rows and on both sides of every boundary, or be routed to a kernel that takes them
    (csrc/bf_gemm256.hip: bf_gemm256_supported, csrc/bf_gemm256_r5.hip: bf_gemm256_r5_supported)."""
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.empty(S, M, K, device="cuda", dtype=dt)
    for s in range(S):  # filled per sample: no fp32 temporary of the whole tensor
        x[s] = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(S, N, K, device="cuda", generator=g) * 0.1).to(dt)
    b = torch.randn(S, N, device="cuda", generator=g)
    y = ops.gemm_nt(x, w, b, S, M, N, K, M * K, dt, 0)
    rows = {0, 1, 255, 256, M - 257, M - 2, M - 1}
    for boundary in (1 << 29, 1 << 30, 1 << 31, 1 << 32):  # in elements and in bytes, of x rows and of y rows
        for per_row in (K, 2 * K, N, 2 * N):
            r = boundary // per_row
            rows.update(q for q in (r - 1, r, r + 1) if 0 <= q < M)
    idx = torch.tensor(sorted(rows), device="cuda")
    for s in range(S):
        ref = x[s, idx].double() @ w[s].double().T + b[s].double()
        err = (y[s, idx].double() - ref).abs().max().item()
        assert err <= 2.0 ** -8 * ref.abs().max().item() + 1e-5 * K ** 0.5, (what, s, err)
    del x, y
    torch.cuda.empty_cache()
