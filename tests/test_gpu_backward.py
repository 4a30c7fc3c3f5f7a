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
