"""Backward of the sampled-weight linear layer (bf_linear_bwd) against gradients of the REAL reference's autograd
graph (tests/golden/linear_grads.npz; identical Philox epsilon, loss = sum_s <y_s, g_s>)."""
import numpy as np
import pytest
import torch

import bayeformers_amd as bf
import bayeformers_amd.nn as bnn
from util import SEED, load_case

pytestmark = pytest.mark.gpu
CASES = ["mix_bias", "nobias_oddK", "moped_frozen", "big"]


def build(c, frozen):
    N, K = c["w_mu"].shape
    layer = bnn.Linear(K, N, bias="b_mu" in c)
    layer.weight.mu.data, layer.weight.rho.data = torch.from_numpy(c["w_mu"]), torch.from_numpy(c["w_rho"])
    if "b_mu" in c:
        layer.bias.mu.data, layer.bias.rho.data = torch.from_numpy(c["b_mu"]), torch.from_numpy(c["b_rho"])
    if frozen:
        layer.weight.mu.requires_grad = False
        if "b_mu" in c:
            layer.bias.mu.requires_grad = False
    layer.layer_id = 0
    return layer.cuda()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype,rtol", [("fp32", 2e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("planned", [False, True])
def test_gradients_match_reference(golden_dir, name, dtype, rtol, planned):
    g = np.load(f"{golden_dir}/linear_grads.npz")
    c = load_case(g, name)
    S, base = int(c["S"]), int(c["base"])
    frozen = "dw_mu" not in c
    layer = build(c, frozen)
    model = bnn.Model(layer)
    model.cross_layer_sampling = planned
    x = torch.from_numpy(c["x"]).cuda().repeat(S, 1).requires_grad_(True)   # sample-major slabs
    gy = torch.from_numpy(c["g"]).cuda()
    bf.manual_seed(SEED, next_sample=base)
    bf.set_compute_dtype(dtype)
    try:
        with model.monte_carlo(S):
            y = model(x)
        loss = (y.view(S, -1, layer.out_features) * gy).sum()
        loss.backward()
    finally:
        bf.set_compute_dtype("bf16")

    def check(got, ref, what):
        ref = torch.from_numpy(ref)
        scale = ref.abs().max().item() + 1e-30
        err = (got.detach().float().cpu() - ref).abs().max().item()
        assert err <= rtol * scale, (what, err, scale)

    M = c["x"].shape[0]
    check(x.grad.view(S, M, -1).sum(0), c["dx"], "dx")
    check(layer.weight.rho.grad, c["dw_rho"], "dw_rho")
    if frozen:
        assert layer.weight.mu.grad is None
    else:
        check(layer.weight.mu.grad, c["dw_mu"], "dw_mu")
    if "b_mu" in c:
        check(layer.bias.rho.grad, c["db_rho"], "db_rho")
        if not frozen:
            check(layer.bias.mu.grad, c["db_mu"], "db_mu")
    # the log-probs carry no gradient, as in the reference
    assert not model.log_prior().requires_grad


@pytest.mark.parametrize("M,N,K,S", [(1024, 320, 192, 3), (512, 200, 136, 2), (2048, 768, 768, 2), (448, 776, 264, 2), (500, 96, 72, 2)])
@pytest.mark.parametrize("planned", [False, True])
def test_gradients_large_shapes_fast_paths(M, N, K, S, planned):
    """Shapes that take the backward's fast paths (16-byte transposes, split-K weight-gradient GEMM, vectorised
    column sums) against fp64 autograd of the oracle's restatement of linear.py:97,104 with the same epsilon."""
    from oracle import bayes_oracle as bo

    g = torch.Generator().manual_seed(M + N)
    layer = bnn.Linear(K, N)
    layer.weight.mu.data = torch.randn(N, K, generator=g) * 0.05
    layer.weight.rho.data = -3.0 + 0.3 * torch.randn(N, K, generator=g)
    layer.bias.mu.data = torch.randn(N, generator=g) * 0.05
    layer.bias.rho.data = -3.0 + 0.3 * torch.randn(N, generator=g)
    layer.layer_id = 0
    layer = layer.cuda()
    model = bnn.Model(layer)
    model.cross_layer_sampling = planned
    x = torch.randn(S * M, K, generator=g)
    gy = torch.randn(S * M, N, generator=g)
    xd = x.cuda().bfloat16().requires_grad_(True)
    base = 9
    bf.manual_seed(SEED, next_sample=base)
    with model.monte_carlo(S):
        y = model(xd)
    y.backward(gy.cuda().bfloat16())

    xr = x.bfloat16().double().requires_grad_(True)
    gyr = gy.bfloat16().double()
    ps = [p.detach().cpu().double().requires_grad_(True)
          for p in (layer.weight.mu, layer.weight.rho, layer.bias.mu, layer.bias.rho)]
    loss = 0
    for s in range(S):
        w = ps[0] + torch.nn.functional.softplus(ps[1]) * bo.eps_tensor((N, K), SEED, base + s, 0, 0).double()
        b = ps[2] + torch.nn.functional.softplus(ps[3]) * bo.eps_tensor((N,), SEED, base + s, 0, 1).double()
        loss = loss + (torch.nn.functional.linear(xr[s * M:(s + 1) * M], w, b) * gyr[s * M:(s + 1) * M]).sum()
    loss.backward()
    got = [xd.grad, layer.weight.mu.grad, layer.weight.rho.grad, layer.bias.mu.grad, layer.bias.rho.grad]
    ref = [xr.grad] + [p.grad for p in ps]
    for a, r, what in zip(got, ref, ("dx", "dmu_w", "drho_w", "dmu_b", "drho_b")):
        err = (a.double().cpu() - r).abs().max().item()
        # dx leaves in bf16 through a bf16 W; the parameter gradients are exact bf16 products accumulated in fp32
        tol = 3e-2 if what == "dx" else 1e-3
        assert err <= tol * r.abs().max().item(), (what, err, r.abs().max().item())


def test_backward_through_a_small_mlp_is_finite_and_deterministic():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 10))
    bmodel = bf.to_bayesian(net, delta=0.05).cuda()
    x = torch.randn(32, 64, device="cuda")
    grads = []
    for _ in range(2):
        bmodel.zero_grad()
        bf.manual_seed(SEED)
        with bmodel.monte_carlo(4):
            out = bmodel(x.repeat(4, 1))
        out.view(4, 32, 10).mean(0).logsumexp(1).sum().backward()
        grads.append([p.grad.clone() for p in bmodel.parameters() if p.grad is not None])
    assert len(grads[0]) == 8  # mu and rho of two weights and two biases
    for a, b in zip(*grads):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("name", ["mix_bias", "moped"])
def test_kl_gradient_opt_in_matches_autograd_of_the_closed_forms(golden_dir, name):
    """set_kl_gradient(True): d/d(mu,rho) of sum_s a_s*log_prior_s + b_s*log_q_s == torch autograd through the
    oracle's fp64 closed forms with the same Philox epsilon (the reference itself has no such gradient)."""
    from oracle import bayes_oracle as bo
    from util import layer_from_case, oracle_priors

    g = np.load(f"{golden_dir}/linear_cases.npz")
    c = load_case(g, name)
    S, base = 3, 40
    layer = layer_from_case(c)
    for p_ in (layer.weight.mu, layer.weight.rho, layer.bias.mu, layer.bias.rho):
        p_.requires_grad_(True)
    model = bnn.Model(layer)
    coef = torch.tensor([[0.7, -1.3], [-0.2, 0.9], [1.1, 0.4]], dtype=torch.float64, device="cuda")
    x = torch.from_numpy(c["x"]).cuda().repeat(S, 1)
    bf.manual_seed(SEED, next_sample=base)
    bf.set_kl_gradient(True)
    try:
        with model.monte_carlo(S):
            model(x)
        lps = model.log_prob_samples()
        assert lps.requires_grad
        (lps * coef).sum().backward()
    finally:
        bf.set_kl_gradient(False)
    assert not model.log_prob_samples().requires_grad  # default: detached, like the reference

    # oracle: differentiable fp64 closed forms
    t = lambda k: torch.from_numpy(c[k]).double().requires_grad_(True)
    mu_w, rho_w, mu_b, rho_b = t("w_mu"), t("w_rho"), t("b_mu"), t("b_rho")
    pw, pb = oracle_priors(c)
    total = 0.0
    for s in range(S):
        for mu, rho, pr, tid in ((mu_w, rho_w, pw, 0), (mu_b, rho_b, pb, 1)):
            eps = bo.eps_tensor(mu.shape, SEED, base + s, 0, tid).double()
            sig = torch.nn.functional.softplus(rho)
            w = mu + eps * sig
            lq = (-bo.LOG_SQRT_2PI - torch.log(sig) - 0.5 * eps ** 2).sum()
            if pr[0] == "mixture":
                lp = bo.mixture_log_prob_terms_f64(w, *pr[1:]).sum()
            else:
                sp = torch.nn.functional.softplus(pr[2].double())
                lp = (-bo.LOG_SQRT_2PI - torch.log(sp) - (w - pr[1].double()) ** 2 / (2 * sp ** 2)).sum()
            total = total + coef[s, 0].item() * lp + coef[s, 1].item() * lq
    total.backward()
    for got, ref, what in ((layer.weight.mu.grad, mu_w.grad, "dmu_w"), (layer.weight.rho.grad, rho_w.grad, "drho_w"),
                           (layer.bias.mu.grad, mu_b.grad, "dmu_b"), (layer.bias.rho.grad, rho_b.grad, "drho_b")):
        scale = ref.abs().max().item()
        # the mixture's score changes by ~1e5 per unit of w around the component crossover (|w| ~ 0.0086): an
        # fp32 ulp of w there is worth ~1e-4 of gradient, so the mixture case gets a looser bound
        tol = 2e-3 if pr[0] == "mixture" else 2e-5
        err = (got.double().cpu() - ref).abs()
        # each gradient is a signed sum over S samples of per-sample scores of magnitude up to ~1 (coef * w / sigma^2):
        # fp32 rounding of those terms leaves a few 1e-7 absolute, whatever the size of the sum that survives
        floor = 3e-7
        assert err.max().item() <= tol * scale + floor, what
        assert (err > 2e-5 * scale + floor).double().mean().item() < 0.1, what  # only the few values near the crossover


def test_training_loop_reduces_the_elbo():
    """End to end: to_bayesian(MLP) trained for a few steps with the reference's recipe (examples/mlp_mnist.py:
    mean prediction over S samples, loss = (lvp - log_prior)/n_batches + nll) and the opt-in KL gradient."""
    from bayeformers_amd.sampling import elbo, sample_bayesian

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4), torch.nn.LogSoftmax(dim=1))
    bmodel = bf.to_bayesian(net, delta=0.05).cuda()
    x = torch.randn(256, 32, device="cuda")
    labels = (x[:, :4].argmax(1)).long()
    opt = torch.optim.Adam([p for p in bmodel.parameters() if p.requires_grad], lr=2e-2)
    bf.manual_seed(SEED)
    bf.set_kl_gradient(True)
    losses = []
    try:
        for _ in range(30):
            opt.zero_grad()
            raw, mean, lp, lq = sample_bayesian(bmodel, x, 4)
            nll = torch.nn.functional.nll_loss(mean[0], labels, reduction="sum")
            loss = elbo(lp, lq, nll.double(), 10)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        bf.set_kl_gradient(False)
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.85 * losses[0], (losses[0], losses[-1])


def test_kl_gradient_with_gradient_buckets_keeps_both_gradients():
    """with set_kl_gradient(True) every mu / rho receives the layer's gradient AND the KL term's.  Under
    GradientBuckets (what training_step builds on sharded ranks) the layer's backward used to write its part into the bucket
    slot and the KL part was dropped.  The bucketed step must leave the gradients of the plain step."""
    from bayeformers_amd.sampling import elbo, sample_bayesian
    from bayeformers_amd.training import GradientBuckets

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 4), torch.nn.LogSoftmax(dim=1))
    bmodel = bf.to_bayesian(net, delta=0.05).cuda()
    x = torch.randn(128, 64, device="cuda")
    labels = torch.randint(0, 4, (128,), device="cuda")
    params = [p for p in bmodel.parameters() if p.requires_grad]

    def step(buckets):
        bf.manual_seed(SEED)
        if buckets is not None:
            buckets.zero()
        else:
            for p in params:
                p.grad = None
        _, mean, lp, lq = sample_bayesian(bmodel, x, 3)
        loss = elbo(lp, lq, torch.nn.functional.nll_loss(mean[0], labels, reduction="sum").double(), 10)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        return {i: p.grad.clone() for i, p in enumerate(params) if p.grad is not None}

    bf.set_kl_gradient(True)
    try:
        ref = step(None)
        buckets = GradientBuckets(params, bucket_bytes=1 << 16)
        for _ in range(2):  # the second step runs on the settled layout
            got = step(buckets)
            assert set(got) == set(ref) and len(ref) >= 8
            for i in ref:
                assert torch.allclose(got[i], ref[i], rtol=1e-5, atol=1e-6 * float(ref[i].abs().max())), i
        buckets.remove()
        # and the KL part is really there: without it the gradients differ
        bf.set_kl_gradient(False)
        nll_only = step(None)
        assert any(not torch.allclose(nll_only[i], ref[i], rtol=1e-3) for i in ref)
    finally:
        bf.set_kl_gradient(False)


def test_backward_after_a_later_forward_regenerates_the_samples():
    """The backward reads the forward's sampled weights while the sampling plan still holds them; if another forward
    has overwritten them in between (fwd 1, fwd 2, bwd 1) it regenerates them from the Philox counter — same grads."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 256))
    bmodel = bf.to_bayesian(net, delta=0.05).cuda()
    S = 2
    x = torch.randn(S * 256, 256, device="cuda").bfloat16()
    gy = torch.randn(S * 256, 256, device="cuda").bfloat16()

    def grads(interleave):
        bmodel.zero_grad()
        bf.manual_seed(SEED)
        with bmodel.monte_carlo(S):
            y = bmodel(x)
            if interleave:
                with torch.no_grad():
                    bmodel(x)  # draws the next sample indices into the same arenas
        assert bmodel._plan is not None
        y.backward(gy)
        return [p.grad.clone() for p in bmodel.parameters() if p.grad is not None]

    for a, b in zip(grads(False), grads(True)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K,S", [(512, 256, 192, 2), (300, 264, 128, 3), (96, 72, 64, 2)])
def test_fused_gelu_training_path(M, N, K, S):
    """A layer with its GELU fused into the GEMM (bayeformers_amd.fuse_activations) under autograd: the forward's one
    launch returns gelu(y) and keeps y, the backward folds gelu' and the bias column sums into one pass.  Against fp64
    autograd of F.gelu(F.linear(x, W_s, b_s)) with the oracle's epsilon (linear.py:97,104)."""
    from oracle import bayes_oracle as bo

    g = torch.Generator().manual_seed(M * 7 + N)
    layer = bnn.Linear(K, N)
    layer.weight.mu.data = torch.randn(N, K, generator=g) * 0.05
    layer.weight.rho.data = -3.0 + 0.3 * torch.randn(N, K, generator=g)
    layer.bias.mu.data = torch.randn(N, generator=g) * 0.05
    layer.bias.rho.data = -3.0 + 0.3 * torch.randn(N, generator=g)
    layer.layer_id = 0
    layer.activation = "gelu"
    layer = layer.cuda()
    model = bnn.Model(layer)
    model.cross_layer_sampling = True
    x = torch.randn(S * M, K, generator=g)
    gy = torch.randn(S * M, N, generator=g)
    xd = x.cuda().bfloat16().requires_grad_(True)
    base = 3
    bf.manual_seed(SEED, next_sample=base)
    with model.monte_carlo(S):
        y = model(xd)
    y.backward(gy.cuda().bfloat16())

    xr = x.bfloat16().double().requires_grad_(True)
    ps = [p.detach().cpu().double().requires_grad_(True)
          for p in (layer.weight.mu, layer.weight.rho, layer.bias.mu, layer.bias.rho)]
    ys = []
    for s in range(S):
        w = ps[0] + torch.nn.functional.softplus(ps[1]) * bo.eps_tensor((N, K), SEED, base + s, 0, 0).double()
        b = ps[2] + torch.nn.functional.softplus(ps[3]) * bo.eps_tensor((N,), SEED, base + s, 0, 1).double()
        ys.append(torch.nn.functional.gelu(torch.nn.functional.linear(xr[s * M:(s + 1) * M], w.detach().bfloat16().double() + (w - w.detach()), b)))
    yr = torch.cat(ys)
    (yr * gy.bfloat16().double()).sum().backward()
    # forward: gelu of the bf16-rounded pre-activation, rounded to bf16
    assert (y.double().cpu() - yr.detach()).abs().max().item() <= 2.0 ** -6 * yr.abs().max().item()
    got = [xd.grad, layer.weight.mu.grad, layer.weight.rho.grad, layer.bias.mu.grad, layer.bias.rho.grad]
    ref = [xr.grad] + [p.grad for p in ps]
    for a, r, what in zip(got, ref, ("dx", "dmu_w", "drho_w", "dmu_b", "drho_b")):
        err = (a.double().cpu() - r).abs().max().item()
        assert err <= 3e-2 * r.abs().max().item(), (what, err, r.abs().max().item())


def test_gemm_act_pre_outputs():
    """bf_gemm_nt_act_pre: d_pre is the plain GEMM output, d_y = gelu(d_pre) — on the 256-wide kernel's two-store
    epilogue and on the fallback (GEMM, then the elementwise kernel) for a shape that kernel does not take."""
    from bayeformers_amd import ops

    for (S, M, N, K) in [(2, 300, 264, 128), (1, 40, 24, 72)]:
        g = torch.Generator(device="cuda").manual_seed(5)
        w = torch.randn(S, N, K, device="cuda", generator=g).bfloat16()
        x = torch.randn(S, M, K, device="cuda", generator=g).bfloat16()
        bias = torch.randn(S, N, device="cuda", generator=g)
        y, pre = ops.gemm_nt_act_pre(x, w, bias, S, M, N, K, M * K, torch.bfloat16, 1)
        plain = ops.gemm_nt(x, w, bias, S, M, N, K, M * K, torch.bfloat16, 0)
        assert torch.equal(pre, plain)
        ref = torch.nn.functional.gelu(pre.double())
        assert (y.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6


def test_inference_forward_does_not_store_the_pre_activation(monkeypatch):
    """Under no_grad the fused-GELU layer must take the one-output launch: the second (pre-activation) store belongs to
    training steps only (trainable parameters alone must not switch it on)."""
    from bayeformers_amd import ops

    layer = bnn.Linear(128, 256)
    layer.layer_id = 0
    layer.activation = "gelu"
    layer = layer.cuda()
    model = bnn.Model(layer)
    model.cross_layer_sampling = True
    x = torch.randn(2 * 256, 128, device="cuda").bfloat16()

    def boom(*a, **k):
        raise AssertionError("bf_gemm_nt_act_pre called in an inference forward")

    monkeypatch.setattr(ops, "gemm_nt_act_pre", boom)
    bf.manual_seed(SEED, next_sample=0)
    with torch.no_grad(), model.monte_carlo(2):
        y = model(x)
    assert y.shape == (512, 256)


def test_stacked_qkv_training_matches_per_layer_autograd(monkeypatch):
    """Three layers that read the same activations (fuse_shared_inputs) under autograd: one stacked forward launch and ONE
    input gradient (bf_gemm_nn_layers) against the same model run layer by layer."""
    import copy

    from bayeformers_amd import ops

    calls = []
    real = ops.gemm_nn_layers
    monkeypatch.setattr(ops, "gemm_nn_layers", lambda *a: (calls.append(1), real(*a))[1])

    class QKV(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.query, self.key, self.value = (torch.nn.Linear(128, 192) for _ in range(3))

        def forward(self, x):
            q, k, v = self.query(x), self.key(x), self.value(x)
            return q * 0.5 + k * k.detach().sign() * 0.25 + v * 1.5

    torch.manual_seed(3)
    net = QKV()
    S, B = 2, 384
    x = torch.randn(B, 128)
    results = []
    for stacked in (False, True):
        bmodel = bf.to_bayesian(copy.deepcopy(net), delta=0.05, freeze=True).cuda().to(torch.bfloat16)
        if stacked:
            assert bf.fuse_shared_inputs(bmodel) == 1
        xd = x.cuda().bfloat16().repeat(S, 1).requires_grad_(True)
        bf.manual_seed(SEED, next_sample=5)
        with bmodel.monte_carlo(S):
            y = bmodel(xd)
        (y.float() ** 2).sum().backward()
        grads = {n: p.grad.detach().float().cpu() for n, p in bmodel.named_parameters() if p.grad is not None}
        results.append((y.detach().float().cpu(), xd.grad.float().cpu(), grads))
    (y0, dx0, g0), (y1, dx1, g1) = results
    assert len(calls) == 1                           # the stacked model took the one-input-gradient path, once
    assert torch.equal(y0, y1)                       # same kernels' outputs, launch grouping aside
    assert (dx0 - dx1).abs().max().item() <= 2.0 ** -7 * dx0.abs().max().item()   # one fp32 sum instead of three bf16 ones
    assert g0.keys() == g1.keys() and len(g0) >= 6
    for n in g0:
        assert torch.allclose(g0[n], g1[n], rtol=1e-5, atol=1e-6 * g0[n].abs().max().item()), n


def _tiny_train_setup(golden_dir, dtype, fused):
    """The tiny BERT of tests/golden/bert_tiny_train.npz, converted and (optionally) rewritten as bench.py's training step."""
    from transformers import BertConfig, BertForSequenceClassification

    def checksum(module):
        return float(sum(p.detach().double().abs().sum() for p in module.parameters()))

    g = np.load(f"{golden_dir}/bert_tiny_train.npz")
    B, L = int(g["B"]), int(g["L"])
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                     vocab_size=1000, max_position_embeddings=64)
    torch.manual_seed(0)
    model = BertForSequenceClassification(cfg).eval()
    bmodel = bf.to_bayesian(model, delta=float(g["delta"]), freeze=True).eval()
    assert checksum(bmodel) == pytest.approx(float(g["checksum"]), rel=1e-6)
    torch.manual_seed(int(g["input_seed"]))
    ids = torch.randint(0, cfg.vocab_size, (B, L))
    mask = torch.ones(B, L, dtype=torch.long)
    labels = torch.randint(0, 2, (B,))
    assert int(ids.sum()) == int(g["ids_sum"]) and np.array_equal(labels.numpy(), g["labels"])
    bmodel = bmodel.cuda()
    if dtype != "fp32":
        bmodel = bmodel.to(torch.bfloat16)
    params = dict(bmodel.named_parameters())  # names as the reference has them, before any module is rewrapped
    if fused:
        assert bf.fuse_activations(bmodel) == 2 and bf.fuse_residual_layernorm(bmodel) == 4
        assert bf.fuse_shared_inputs(bmodel) == 2 and bf.fuse_attention(bmodel) and bf.fuse_embeddings(bmodel) == 1
    inputs = {"input_ids": ids.cuda(), "attention_mask": mask.cuda()}
    return g, bmodel, params, inputs, labels.cuda()


def _check_tiny_train_grads(g, params, tol, tol_vec):
    names = [str(n) for n in g["names"]]
    assert sorted(names) == sorted(n for n, p in params.items() if p.grad is not None)
    worst = {}
    gmax = max(float(g[f"stat/{n}"][2]) for n in names)
    for n in names:
        got = params[n].grad.detach().double().cpu().numpy()
        ref_sum, ref_abs, ref_max = g[f"stat/{n}"]
        assert np.isfinite(got).all(), n
        if ref_max < 1e-6 * gmax:  # mathematically zero (the key bias: softmax is shift invariant): rounding noise only
            assert np.abs(got).max() < 1e-4 * gmax, (n, np.abs(got).max())
            continue
        assert abs(np.abs(got).sum() - ref_abs) <= 2 * tol_vec * ref_abs + 1e-12, (n, np.abs(got).sum(), ref_abs)
        if f"grad/{n}" in g.files:
            ref = g[f"grad/{n}"].astype(np.float64)
            # matrices: largest deviation against the largest entry; bias vectors (sums of 16-bit rounded rows whose
            # terms largely cancel): relative L2 distance
            err = np.abs(got - ref).max() / ref_max if ref.ndim == 2 else np.linalg.norm(got - ref) / np.linalg.norm(ref)
            worst[n] = err / (tol if ref.ndim == 2 else tol_vec)
    assert len(worst) == 14
    assert all(e <= 1.0 for e in worst.values()), worst


@pytest.mark.parametrize("dtype,fused,tol,tol_vec", [("fp32", False, 1e-4, 1e-4), ("fp32", True, 1e-4, 1e-4),
                                                     ("bf16", True, 5e-2, 1.5e-1)])
def test_training_step_gradients_match_reference_bert_tiny(golden_dir, dtype, fused, tol, tol_vec):
    """One training step of the REAL reference on the tiny BERT (tests/golden/bert_tiny_train.npz, generated by
    make_golden.py:bert_train_case = examples/bert_glue.py:63-66, 234-239 with dropout off): the gradient of every
    trainable tensor.  `fused` = the rewrites bench.py's training step runs on (GELU in the GEMM epilogue with its
    backward, residual+LayerNorm, q/k/v as one autograd node, the attention kernels, the embedding block).
    Tolerances, per tensor stored in full: weight matrices max |g - g_ref| <= tol * max |g_ref| (fp32 1e-4, measured
    <= 2e-5; bf16 activations end to end 5e-2, measured <= 2.6e-2); bias vectors |g - g_ref|_2 <= tol_vec * |g_ref|_2
    (bf16 1.5e-1: the query bias of the first layer, a sum over 16-bit rounded rows behind the softmax backward's
    cancellation, measures 9e-2; all others <= 1e-2); every other tensor: sum |g| within 2 tol_vec."""
    from bayeformers_amd.sampling import elbo, sample_bayesian

    g, bmodel, params, inputs, labels = _tiny_train_setup(golden_dir, dtype, fused)
    S, NB = int(g["S"]), int(g["n_batches"])
    bf.manual_seed(SEED)
    bf.set_compute_dtype(dtype)
    try:
        raw, mean, lp, lq = sample_bayesian(bmodel, inputs, S)
        nll = torch.nn.functional.cross_entropy(mean[0].float(), labels)
        loss = elbo(lp, lq, nll.double(), NB)
        loss.backward()
    finally:
        bf.set_compute_dtype("bf16")
    assert np.abs(raw[0].detach().float().cpu().numpy() - g["logits"]).max() < max(tol, 2e-4) * max(1.0, np.abs(g["logits"]).max())
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-3)
    _check_tiny_train_grads(g, params, tol, tol_vec)


@pytest.mark.parametrize("dtype,tol,tol_vec", [("fp32", 1e-4, 1e-4), ("bf16", 5e-2, 1.5e-1)])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_training_step_gradients_match_reference_bert_tiny(golden_dir, dtype, tol, tol_vec, world):
    """The S-sharded training step (bayeformers_amd.training) against the reference's single-process gradients: the
    shards of `world` ranks run here one after the other on one GPU, each exactly as its rank would — forward of its
    own samples (`monte_carlo(count, span=(start, S))`), the loss on the all-reduced mean logits with only the local part
    in the autograd graph (what sampling._all_reduce_sum builds), backward — and their gradients are SUMMED, which is
    what GradientBuckets' all-reduce does.  The sum must be the gradient of examples/bert_glue.py:234-239."""
    from bayeformers_amd.sampling import elbo, repeat_inputs, shard_span

    g, bmodel, params, inputs, labels = _tiny_train_setup(golden_dir, dtype, True)
    S, NB, B = int(g["S"]), int(g["n_batches"]), int(g["B"])
    spans = [shard_span(S, r, world) for r in range(world)]
    assert sum(c for _, c in spans) == S and all(c > 0 for _, c in spans)
    bf.set_compute_dtype(dtype)
    try:
        def forward(start, count):
            bf.manual_seed(SEED)  # every rank starts the step from the same global sample counter
            with bmodel.monte_carlo(count, span=(start, S)):
                out = bmodel(**repeat_inputs(inputs, count))
            return out.logits.reshape(count, B, -1), bmodel.log_prob_samples().sum(0)

        with torch.no_grad():  # what the all-reduce delivers: the sums over all ranks
            parts = [forward(st, c) for st, c in spans]
            total = sum(p[0].double().sum(0) for p in parts)
            lp_total = sum(p[1] for p in parts) / S
        for st, c in spans:
            logits, _ = forward(st, c)
            local = logits.double().sum(0)
            mean = ((local + (total - local.detach())) / S).to(logits.dtype)
            nll = torch.nn.functional.cross_entropy(mean.float(), labels)
            loss = elbo(lp_total[0], lp_total[1], nll.double(), NB)
            loss.backward()  # accumulates into .grad: the sum over the ranks
    finally:
        bf.set_compute_dtype("bf16")
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-3)
    _check_tiny_train_grads(g, params, tol, tol_vec)


@pytest.mark.parametrize("device_counter,reentrant", [(False, False), (True, False), (False, True), (True, True)])
def test_checkpointed_blocks_recompute_the_forwards_epsilon(device_counter, reentrant):
    """torch.utils.checkpoint re-runs a block's forward DURING backward, outside any bnn.Model forward.  The Bayesian
    layers inside must then draw the epsilon of the forward they repeat (bayeformers_amd.random.recompute_context) —
    the same sample indices, also with the device-resident counter, which has moved on by then: outputs, log-probs and
    every gradient equal those of the run without checkpointing.  use_reentrant=True: there the autograd nodes
    built DURING the recomputation are the ones that run backward — they must regenerate epsilon from the counter value the
    forward saw, not from the live counter."""
    from torch.utils.checkpoint import checkpoint

    class Block(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.a, self.b = torch.nn.Linear(d, 2 * d), torch.nn.Linear(2 * d, d)

        def forward(self, x):
            return x + self.b(torch.nn.functional.gelu(self.a(x)))

    class Net(torch.nn.Module):
        def __init__(self, d, ckpt):
            super().__init__()
            self.blocks = torch.nn.ModuleList([Block(d) for _ in range(3)])
            self.head = torch.nn.Linear(d, 4)
            self.ckpt, self.reentrant = ckpt, False

        def forward(self, x):
            for blk in self.blocks:
                x = checkpoint(blk, x, use_reentrant=self.reentrant) if self.ckpt else blk(x)
            return self.head(x)

    d, S, B = 128, 3, 80   # 80 rows per sample: the planned (cross-layer sampled) path; the head (N = 4) runs on its own
    torch.manual_seed(0)
    net = Net(d, False)
    bmodel = bf.to_bayesian(net, delta=0.05).cuda()
    x = torch.randn(B, d, device="cuda")
    target = torch.randn(S * B, 4, device="cuda")

    def run(ckpt):
        net_b = bmodel.model
        net_b.ckpt, net_b.reentrant = ckpt, reentrant
        for p in bmodel.parameters():
            p.grad = None
        bf.manual_seed(SEED, next_sample=7)
        with bmodel.monte_carlo(S):
            # (the reentrant form only checkpoints a block whose input requires a gradient)
            out = bmodel(x.repeat(S, 1).requires_grad_(reentrant))
        lps = bmodel.log_prob_samples().clone()
        # a second forward between forward and backward would move a host-side counter; the device counter has moved anyway
        ((out - target) ** 2).mean().backward()
        grads = {n: p.grad.clone() for n, p in bmodel.named_parameters() if p.grad is not None}
        return out.detach().clone(), lps, grads

    if device_counter:
        bf.use_device_counter(True)
    try:
        out0, lp0, g0 = run(False)
        out1, lp1, g1 = run(True)
    finally:
        if device_counter:
            bf.use_device_counter(False)
    assert torch.equal(out0, out1) and torch.equal(lp0, lp1)
    assert g0.keys() == g1.keys() and len(g0) >= 14
    for n in g0:
        scale = g0[n].abs().max().item() + 1e-30
        assert (g0[n] - g1[n]).abs().max().item() <= 1e-6 * scale, n


def test_device_counter_is_copied_once_per_forward_and_not_at_all_without_gradients(monkeypatch):
    """In device-counter mode every Bayesian layer's autograd node keeps the counter value its forward saw.  That used to
    be one 4 us copy kernel per layer and forward — 36 of them in a BERT-base inference step, which needs none.  A forward
    under no_grad makes no copy; a forward that records gradients makes ONE, shared by its layers; the gradients are those
    of the host-counter run."""
    from bayeformers_amd import random as bfr

    d, S, B = 128, 3, 80
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(d, 2 * d), torch.nn.GELU(), torch.nn.Linear(2 * d, d), torch.nn.Linear(d, 4))
    bmodel = bf.to_bayesian(net, delta=0.05).cuda()
    x = torch.randn(B, d, device="cuda")
    copies = []
    real = bfr.counter_snapshot

    def counting(needed=True):
        r = real(needed)
        if r is not None:
            copies.append(r)
        return r

    monkeypatch.setattr(bfr, "counter_snapshot", counting)

    def run(grad):
        for p in bmodel.parameters():
            p.grad = None
        bf.manual_seed(SEED, next_sample=5)
        with torch.set_grad_enabled(grad), bmodel.monte_carlo(S):
            out = bmodel(x.repeat(S, 1))
        if grad:
            (out ** 2).mean().backward()
        return out.detach().clone(), {n: p.grad.clone() for n, p in bmodel.named_parameters() if p.grad is not None}

    out_host, g_host = run(True)
    assert not copies  # host-side counter: nothing to copy
    bf.use_device_counter(True)
    try:
        out_ng, _ = run(False)
        assert not copies, "a forward under no_grad copied the device counter"
        out_dev, g_dev = run(True)
        assert len(copies) >= 3 and len({id(c) for c in copies}) == 1, "the layers of one forward share one copy"
    finally:
        bf.use_device_counter(False)
    assert torch.equal(out_host, out_ng) and torch.equal(out_host, out_dev)
    assert g_host.keys() == g_dev.keys() and len(g_host) >= 6
    for n in g_host:
        assert torch.equal(g_host[n], g_dev[n]), n


# ------------------------------------------------------------------------------------------------------------------------
# Round 5: the bias gradients' column sums of the query / key / value layers ride in the kernel that PRODUCES their output
# gradients (the one-tile attention backward) instead of three passes over those gradients.  (The same inside the residual +
# LayerNorm backward was built, measured slower than a column-sum pass of its own and removed: LABBOOK.md, round 5.)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("S,Bs,H,p", [(2, 3, 2, 0.0), (5, 4, 12, 0.0), (3, 2, 4, 0.1)])
def test_attention_backward_leaves_the_column_sums_of_dq_dk_dv(dtype, S, Bs, H, p):
    """bf_attention_bwd_colsum (one-tile sequences): dq / dk / dv bit-identical to bf_attention_bwd's, plus the per-sample
    column sums of each — what the Bayesian query / key / value layers take as their bias gradients."""
    from bayeformers_amd import ops

    B, T = S * Bs, 128
    g = torch.Generator().manual_seed(B * 31 + H)
    q, k, v = (torch.randn(B, T, H * 64, generator=g).cuda().to(dtype).view(B, T, H, 64).transpose(1, 2) for _ in range(3))
    go = torch.randn(B, T, H, 64, generator=g).cuda().to(dtype)
    drop = ops.Dropout(p, SEED, 1, 2) if p > 0 else None
    if drop is not None:
        out, lse, keep = ops.attention_forward(q, k, v, None, 0.125, None, want_lse=True, drop=drop, want_keep=True)
    else:
        (out, lse), keep = ops.attention_forward(q, k, v, None, 0.125, None, want_lse=True), None
    plain = ops.attention_backward(q, k, v, None, None, out, go, lse, 0.125, p, keep)
    folded = ops.attention_backward(q, k, v, None, None, out, go, lse, 0.125, p, keep, colsum_samples=S)
    for t, (a, b) in enumerate(zip(plain, folded)):
        assert torch.equal(a, b)
        flat = b.reshape(B * T, H * 64)          # the [S M, N] rows the projection's backward sees: a view, same storage
        cs = ops.take_colsum(flat, S, H * 64)
        assert cs is not None, t
        ref = flat.view(S, Bs * T, H * 64).double().sum(1)
        assert (cs.double() - ref).abs().max().item() <= 1e-5 * max(1.0, float(flat.float().abs().sum(0).max())), t


def test_folded_column_sums_give_the_bias_gradients_of_the_plain_path():
    """End to end on a 2-layer BERT with 128-token sequences in training mode: every Bayesian layer's bias gradient with the
    query / key / value column sums taken from the attention backward equals the gradient with a column-sum pass per layer."""
    from transformers import BertConfig, BertForSequenceClassification

    from bayeformers_amd import ops
    from bayeformers_amd.sampling import elbo, sample_bayesian

    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=500,
                     max_position_embeddings=128)
    torch.manual_seed(0)
    bmodel = bf.to_bayesian(BertForSequenceClassification(cfg), delta=0.05, freeze=True).cuda().to(torch.bfloat16)
    bf.fuse_activations(bmodel), bf.fuse_residual_layernorm(bmodel), bf.fuse_shared_inputs(bmodel)
    bf.fuse_attention(bmodel), bf.fuse_embeddings(bmodel)
    bmodel.train()
    ids = torch.randint(0, 500, (4, 128), generator=torch.Generator().manual_seed(1)).cuda()
    inputs = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    labels = torch.randint(0, 2, (4,), generator=torch.Generator().manual_seed(2)).cuda()
    params = {n: p_ for n, p_ in bmodel.named_parameters() if p_.requires_grad}

    def grads(fold):
        ops._NO_COLSUM_FOLD = not fold
        ops.COLSUMS_FOLDED[0] = 0
        try:
            for p_ in params.values():
                p_.grad = None
            bf.manual_seed(SEED)
            torch.manual_seed(1234)   # (the embedding block's dropout is the framework's: the same mask in both runs)
            _, mean, lp, lq = sample_bayesian(bmodel, inputs, 3)
            loss = elbo(lp, lq, torch.nn.functional.cross_entropy(mean[0].float(), labels).double(), 10)
            loss.backward()
            return {n: p_.grad.float().clone() for n, p_ in params.items() if p_.grad is not None}, ops.COLSUMS_FOLDED[0]
        finally:
            ops._NO_COLSUM_FOLD = False

    plain, n_plain = grads(False)
    folded, n_folded = grads(True)
    assert n_plain == 0 and n_folded == 2 * 3, n_folded      # per layer: query, key, value
    assert set(plain) == set(folded)
    for n in plain:
        scale = float(plain[n].abs().max())
        # fp32 sums in another order for the biases and the LayerNorm parameters; everything else: the same bits
        tol = 1e-5 * max(scale, 1e-6) if ("bias" in n or "LayerNorm" in n) else 0.0
        if "embeddings" in n and "LayerNorm" not in n:
            tol = 5e-2 * scale   # the framework's embedding backward scatters with bf16 atomics: not run-to-run reproducible
        assert (plain[n] - folded[n]).abs().max().item() <= tol, n


@pytest.mark.parametrize("train_mode", [False, True])
def test_graphed_training_step_is_the_eager_step(golden_dir, train_mode):
    """training.GraphedTrainingStep on the tiny BERT of bert_tiny_train.npz in bench.py's rewritten form: two eager steps, one
    capture, three replays — step k (loss and EVERY parameter afterwards) is the k-th eager training_step bit for bit, also
    with the model in train() (HF dropout 0.1 inside the kernels: the device-resident call counter gives every replay the masks
    of its own step), also when a new batch is copied in."""
    import copy

    from bayeformers_amd import random as bfr
    from bayeformers_amd.training import GraphedTrainingStep, training_step

    g, bmodel0, _, inputs, labels = _tiny_train_setup(golden_dir, "bf16", True)
    S, NB = int(g["S"]), int(g["n_batches"])
    batches = [inputs, {"input_ids": inputs["input_ids"].flip(0).contiguous(), "attention_mask": inputs["attention_mask"]}]
    order = [0, 0, 1, 0, 1]
    nll = lambda mean: torch.nn.functional.cross_entropy(mean[0].float(), labels)
    bf.set_compute_dtype("bf16")

    def fresh():
        m = copy.deepcopy(bmodel0)
        m.train(train_mode)
        for mod in m.modules():   # only dropout decides training mode here; make sure it is active / inactive everywhere
            if isinstance(mod, torch.nn.Dropout):
                mod.train(train_mode)
        bf.fuse_attention(m)
        ps = [p for p in m.parameters() if p.requires_grad]
        # (the learning rate as a device tensor in BOTH runs: GraphedTrainingStep turns a python float into one — a float would
        # be baked into the capture — and the fused update rounds lr = 1e-3 differently as fp32 tensor and as double)
        return m, torch.optim.AdamW(ps, lr=torch.tensor(1e-3, device="cuda"), eps=1e-8, weight_decay=0.0, fused=True, capturable=True)

    torch.manual_seed(11)
    bf.manual_seed(SEED)
    m_e, opt_e = fresh()
    losses_e = [float(training_step(m_e, batches[b], S, nll, opt_e, NB, max_grad_norm=1.0)) for b in order]
    assert bfr.STATE.device_counter is None

    torch.manual_seed(11)
    bf.manual_seed(SEED)
    m_g, opt_g = fresh()
    step = GraphedTrainingStep(m_g, batches[0], S, nll, opt_g, NB, max_grad_norm=1.0, eager_steps=2)
    try:
        losses_g = [float(step(batches[b])) for b in order]
        assert step.captures == 1 and step.steps == len(order) and bfr.STATE.device_counter is not None
    finally:
        step.close()
    assert bfr.STATE.device_counter is None and bfr.get_state()[1] == S * len(order)
    assert losses_g == losses_e, (losses_g, losses_e)
    for (n, pe), (_, pg) in zip(m_e.named_parameters(), m_g.named_parameters()):
        assert torch.equal(pe, pg), n
    if train_mode:  # the steps really dropped: the same model without dropout ends elsewhere
        assert losses_e[1] != pytest.approx(losses_e[0], rel=1e-9)


def test_device_resident_dropout_call_draws_the_host_counters_masks(golden_dir):
    """Device-counter mode (what a captured training step runs in): the dropout kernels add a device-resident copy of the call
    counter to `call` = 0 — forward k must drop exactly what forward k drops with the host-side counter, in the forward and
    in the masks its backward regenerates."""
    g, bmodel, params, inputs, labels = _tiny_train_setup(golden_dir, "bf16", True)
    bmodel.train()
    bf.set_compute_dtype("bf16")

    def run(device_counter):
        bf.manual_seed(SEED)
        torch.manual_seed(5)   # torch's own dropout (the embedding block)
        if device_counter:
            bf.use_device_counter(True)
        try:
            outs = []
            for _ in range(3):
                for p in bmodel.parameters():
                    p.grad = None
                with bmodel.monte_carlo(2):
                    out = bmodel(**{k: v.repeat(2, 1) for k, v in inputs.items()}).logits
                out.float().square().sum().backward()
                outs.append((out.detach().clone(), {n: p.grad.clone() for n, p in params.items() if p.grad is not None}))
        finally:
            if device_counter:
                bf.use_device_counter(False)
        return outs

    host, dev = run(False), run(True)
    assert not torch.equal(host[0][0], host[1][0])   # the forwards differ from one another (fresh epsilon, fresh masks)
    for k, ((oh, gh), (od, gd)) in enumerate(zip(host, dev)):
        assert torch.equal(oh, od), k
        assert gh.keys() == gd.keys()
        for n in gh:
            assert torch.equal(gh[n], gd[n]), (k, n)


def test_deferred_weight_reduction_gives_the_per_layer_gradients(golden_dir, monkeypatch):
    """training.DeferredParamGrads: from the second step of a shape on, training_step leaves every layer's per-sample weight
    gradients in standing buffers and reduces them with ONE bf_param_grad_table launch after the backward pass — the same
    arithmetic in the same order: parameters after every step equal those of the per-layer path bit for bit; a step with
    another batch shape falls back (and re-arms for that shape); the gradients are views of the manager's flat buffer."""
    import copy

    from bayeformers_amd import training
    from bayeformers_amd.training import training_step

    g, bmodel0, _, inputs, labels = _tiny_train_setup(golden_dir, "bf16", True)
    S, NB = int(g["S"]), int(g["n_batches"])
    half = {k: v[: v.shape[0] // 2].contiguous() for k, v in inputs.items()}
    plan = [(inputs, labels), (inputs, labels), (inputs, labels), (half, labels[: labels.shape[0] // 2]), (inputs, labels)]
    bf.set_compute_dtype("bf16")

    def run(deferred):
        monkeypatch.setattr(training, "_NO_DEFERRED", not deferred)
        m = copy.deepcopy(bmodel0)
        bf.fuse_attention(m)
        opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, eps=1e-8, weight_decay=0.0, fused=True)
        bf.manual_seed(SEED)
        snaps, armed = [], []
        for x, y in plan:
            nll = lambda mean: torch.nn.functional.cross_entropy(mean[0].float(), y)  # noqa: E731
            orig_finish = training.DeferredParamGrads.finish
            was = {}

            def finish(self, _o=orig_finish, _w=was):
                _w["armed"] = self.armed
                return _o(self)

            monkeypatch.setattr(training.DeferredParamGrads, "finish", finish)
            loss = float(training_step(m, x, S, nll, opt, NB, max_grad_norm=1.0))
            monkeypatch.setattr(training.DeferredParamGrads, "finish", orig_finish)
            armed.append(bool(was.get("armed")))
            snaps.append((loss, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}))
        return snaps, armed, m

    ref, armed_ref, _ = run(False)
    got, armed, m = run(True)
    assert armed_ref == [False] * 5 and armed == [False, True, True, False, False]   # observe, defer, defer, new shape, re-observed
    for k, ((l0, p0), (l1, p1)) in enumerate(zip(ref, got)):
        assert l0 == l1, k
        for n in p0:
            assert torch.equal(p0[n], p1[n]), (k, n)
    mgr = m._pgrad
    assert mgr.table is not None and len(mgr.table["layers"]) == len([l for l in m.fused_children()])


def test_ffn_pair_node_matches_the_two_layer_path():
    """fuse_ffn_pairs: dense -> GELU -> dense of a transformer layer as one autograd node whose backward takes the GELU's
    derivative in the epilogue of the down-projection's input-gradient GEMM (bf_gemm_nn_actgrad).  Same forward bit for bit;
    every gradient agrees with the two-node path to bf16 rounding (the fused form rounds the intermediate gradient once
    instead of twice), and the kernel alone agrees with its definition in fp32."""
    import copy

    from transformers import BertConfig, BertForSequenceClassification

    from bayeformers_amd import ops

    # the kernel against its definition
    g = torch.Generator(device="cuda").manual_seed(3)
    S, M, N, K = 3, 512, 256, 1024
    dy = torch.randn(S, M, N, device="cuda", generator=g).bfloat16()
    w = (torch.randn(S, N, K, device="cuda", generator=g) * 0.05).bfloat16()
    pre = torch.randn(S, M, K, device="cuda", generator=g).bfloat16()
    assert ops.gemm_nn_actgrad_supported(dy, w, pre)
    got = ops.gemm_nn_actgrad(dy, w, pre).float()
    x = pre.float()
    dgelu = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    want = torch.einsum("smn,snk->smk", dy.float(), w.float()) * dgelu
    assert (got - want).abs().max().item() <= 2 ** -7 * want.abs().max().item()
    assert torch.equal(ops.gemm_nn(dy, w), ops.gemm_nn(dy, w))  # (the plain form is untouched)

    cfg = BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=500,
                     max_position_embeddings=128)
    torch.manual_seed(0)
    base = bf.to_bayesian(BertForSequenceClassification(cfg).eval(), delta=0.05).cuda().to(torch.bfloat16)
    ids = torch.randint(0, 500, (4, 128), device="cuda")
    labels = torch.randint(0, 2, (4,), device="cuda")
    bf.set_compute_dtype("bf16")

    def run(pair):
        m = copy.deepcopy(base)
        bf.fuse_activations(m), bf.fuse_residual_layernorm(m), bf.fuse_shared_inputs(m), bf.fuse_attention(m)
        if pair:
            assert bf.fuse_ffn_pairs(m) == 2
        bf.manual_seed(SEED)
        S = 3
        with m.monte_carlo(S):
            out = m(input_ids=ids.repeat(S, 1)).logits
        torch.nn.functional.cross_entropy(out.float().view(S, 4, 2).mean(0), labels).backward()
        return out.detach(), {n: p.grad.float() for n, p in m.named_parameters() if p.grad is not None}

    o0, g0 = run(False)
    o1, g1 = run(True)
    assert torch.equal(o0, o1)
    assert g0.keys() == g1.keys() and len(g0) > 40
    for n in g0:
        scale = g0[n].abs().max().item() + 1e-20
        assert (g0[n] - g1[n]).abs().max().item() <= 3e-2 * scale, (n, (g0[n] - g1[n]).abs().max().item() / scale)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("S,M,N,K", [(2, 200, 128, 1000),      # fewer rows than one tile, output width not a multiple of 256
                                     (1, 4100, 256, 264),      # a ragged last row unit; one whole and one 8-wide column tile
                                     (3, 33, 512, 520),        # two row units, the second with one row
                                     (5, 1024, 768, 3072)])    # the BERT-base shape, scaled down in rows
def test_gemm_nn_actgrad_ragged_shapes(dtype, S, M, N, K):
    """bf_gemm_nn_actgrad at the edges of its tiling: the pre-activation chunks are fetched two row blocks ahead of their use,
    so rows past M and columns past the last whole tile must neither be read nor written.  The output is allocated inside a
    guard band that has to come back untouched."""
    from bayeformers_amd import ops

    g = torch.Generator(device="cuda").manual_seed(11)
    dy = torch.randn(S, M, N, device="cuda", generator=g).to(dtype)
    w = (torch.randn(S, N, K, device="cuda", generator=g) * 0.05).to(dtype)
    pre = (torch.randn(S, M, K, device="cuda", generator=g) * 1.5).to(dtype)
    assert ops.gemm_nn_actgrad_supported(dy, w, pre)
    from bayeformers_amd import _C

    band, mark = 8192, 12345.0
    buf = torch.full((S * M * K + 2 * band,), mark, device="cuda", dtype=dtype)
    got = buf[band:band + S * M * K].view(S, M, K)
    _C.check(_C.lib().bf_gemm_nn_actgrad(dy.data_ptr(), w.data_ptr(), got.data_ptr(), pre.data_ptr(), ops._TORCH2BF[dtype], S, M, N, K,
                                         1, ops._stream_ptr()), "bf_gemm_nn_actgrad")
    assert (buf[:band] == mark).all() and (buf[band + S * M * K:] == mark).all()
    x = pre.float()
    dgelu = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    want = torch.einsum("smn,snk->smk", dy.float(), w.float()) * dgelu
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10  # one rounding of the product to the output type
    assert (got.float() - want).abs().max().item() <= tol * want.abs().max().item()
    # the same launch is the plain NN product followed by the derivative, to one more rounding of the intermediate
    two_step = (ops.gemm_nn(dy, w).float() * dgelu)
    assert (got.float() - two_step).abs().max().item() <= 2 * tol * want.abs().max().item()
    assert torch.equal(got, ops.gemm_nn_actgrad(dy, w, pre))  # deterministic
