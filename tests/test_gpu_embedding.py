"""bnn.Embedding (extension; the reference has no Bayesian embedding, so parity is against the oracle's restatement
of `F.embedding(ids, weight.sample())` with the reference's Gaussian.sample / log_prob, gaussian.py:81-116)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import bayeformers_amd as bf
import bayeformers_amd.nn as bnn
from util import SEED

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import bayes_oracle as bo  # noqa: E402

pytestmark = pytest.mark.gpu


def make(V, D, seed, delta=None, padding_idx=None):
    g = torch.Generator().manual_seed(seed)
    emb = torch.nn.Embedding(V, D, padding_idx=padding_idx)
    emb.weight.data = torch.randn(V, D, generator=g) * 0.05
    layer = bnn.Embedding.from_frequentist(emb, delta=delta)
    if delta is None:
        layer.weight.mu.data = torch.randn(V, D, generator=g) * 0.05
        layer.weight.rho.data = -4.0 + 0.5 * torch.randn(V, D, generator=g)
    layer.layer_id = 3
    return layer.cuda()


def oracle_rows(layer, ids, S, base):
    """fp64 rows of the S table draws + fp64 closed-form log-probs per sample."""
    mu, rho = layer.weight.mu.detach().cpu(), layer.weight.rho.detach().cpu()
    rows, lps = [], []
    for s in range(S):
        eps = bo.eps_tensor(mu.shape, SEED, base + s, layer.layer_id, 0)
        w = mu.double() + torch.nn.functional.softplus(rho.double()) * eps.double()
        rows.append(w[ids])
        pr = layer.weight_prior
        if isinstance(pr, bnn.ScaledGaussianMixture):
            lp = bo.mixture_log_prob_f64(w, float(pr.pi), float(pr.sigma1), float(pr.sigma2))
        else:
            lp = bo.gaussian_log_prob_f64(None, pr.mu.detach().cpu(), pr.rho.detach().cpu(), x=w)
        lps.append((lp, bo.gaussian_log_prob_f64(eps, mu, rho)))
    return torch.stack(rows).numpy(), np.array(lps)


@pytest.mark.parametrize("delta", [None, 0.05])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3)])
def test_forward_and_logprobs(delta, dtype, tol):
    V, D, B, T, S, base = 523, 96, 3, 17, 4, 11
    layer = make(V, D, 1, delta).to(dtype)
    assert layer.weight.mu.dtype == torch.float32
    model = bnn.Model(layer)
    ids = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(2))
    bf.manual_seed(SEED, next_sample=base)
    with model.monte_carlo(S):
        out = model(ids.repeat(S, 1).cuda())
    assert out.shape == (S * B, T, D) and out.dtype == dtype
    ref, lps = oracle_rows(layer, ids, S, base)          # [S, B, T, D]
    got = out.detach().view(S, B, T, D).double().cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol * scale
    lp = model.log_prob_samples().cpu().numpy()
    np.testing.assert_allclose(lp, lps, rtol=2e-6)
    np.testing.assert_allclose(float(model.log_prior()), lps[:, 0].mean(), rtol=1e-5)
    np.testing.assert_allclose(float(layer.log_variational_posterior), lps[:, 1].mean(), rtol=1e-5)


def test_backward_matches_autograd():
    V, D, B, T, S, base = 211, 64, 2, 29, 3, 5
    layer = make(V, D, 4, padding_idx=7)
    model = bnn.Model(layer)
    ids = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(5))
    ids[0, :4] = 7          # padding rows, and repeated ids exercise the atomics
    ids[1, :6] = 9
    gy = torch.randn(S, B, T, D, generator=torch.Generator().manual_seed(6))
    bf.manual_seed(SEED, next_sample=base)
    with model.monte_carlo(S):
        out = model(ids.repeat(S, 1).cuda())
    (out.view(S, B, T, D) * gy.cuda()).sum().backward()

    mu = layer.weight.mu.detach().cpu().double().requires_grad_(True)
    rho = layer.weight.rho.detach().cpu().double().requires_grad_(True)
    loss = 0
    for s in range(S):
        eps = bo.eps_tensor((V, D), SEED, base + s, layer.layer_id, 0).double()
        w = mu + torch.log1p(torch.exp(rho)) * eps
        loss = loss + (torch.nn.functional.embedding(ids, w, padding_idx=7) * gy[s].double()).sum()
    loss.backward()
    for got, ref in ((layer.weight.mu.grad, mu.grad), (layer.weight.rho.grad, rho.grad)):
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 1e-5 * ref.abs().max().item()
    assert layer.weight.mu.grad[7].abs().max().item() == 0


def test_opt_in_conversion_and_sample_independence():
    """to_bayesian converts nn.Embedding only after enable_embedding(); S batched == S serial forwards."""
    net = torch.nn.Sequential(torch.nn.Embedding(50, 32), torch.nn.Linear(32, 8))
    assert not any(isinstance(m, bnn.Embedding) for m in bf.to_bayesian(net).modules())
    bf.enable_embedding()
    try:
        bmodel = bf.to_bayesian(net, delta=0.1).cuda()
    finally:
        bf.enable_embedding(False)
    assert isinstance(bmodel.model[0], bnn.Embedding) and isinstance(bmodel.model[1], bnn.Linear)
    ids = torch.randint(0, 50, (4, 6)).cuda()
    S = 3
    bf.manual_seed(SEED)
    with bmodel.monte_carlo(S):
        batched = bmodel(ids.repeat(S, 1)).view(S, 4, 6, 8)
    lp_b = bmodel.log_prob_samples().clone()
    bf.manual_seed(SEED)
    for s in range(S):
        y = bmodel(ids)
        assert torch.equal(y, batched[s])
        assert torch.allclose(bmodel.log_prob_samples()[0], lp_b[s], rtol=1e-12)
