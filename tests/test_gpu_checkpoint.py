"""Checkpoint compatibility: a state dict written by the REAL reference after forwards
(/root/reference/examples/bert_glue.py:303-309 saves `b_model.state_dict()`) loads into `to_bayesian(model)` of this
package as it is — duplicated shared-prior keys and the two per-layer log-prob scalars included — and the HIP path
reproduces the reference's forward of that checkpoint (tests/golden/checkpoint.npz, made by make_golden.py from the
imported reference with the oracle's epsilon).  Also: this package's own state_dict() carries the log-prob scalars of
the last forward, as the reference's does."""
import numpy as np
import pytest
import torch

import bayeformers_amd as bf
from bayeformers_amd.sampling import sample_bayesian

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def _net():
    return torch.nn.Sequential(torch.nn.Linear(48, 96), torch.nn.Tanh(), torch.nn.Linear(96, 32), torch.nn.Tanh(),
                               torch.nn.Linear(32, 6, bias=False))


@pytest.mark.parametrize("variant,kw", [("mixture", {}), ("moped", {"delta": 0.07, "freeze": True})])
@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 4e-2)])
def test_reference_checkpoint_evaluates_on_the_hip_path(golden_dir, variant, kw, dtype, tol):
    g = np.load(f"{golden_dir}/checkpoint.npz")
    keys = [str(k) for k in g[f"{variant}/keys"]]
    torch.manual_seed(777)  # a different initialisation: everything must come from the checkpoint
    bmodel = bf.to_bayesian(_net(), **kw)
    assert list(bmodel.state_dict().keys()) == keys
    assert sorted(n for n, p in bmodel.named_parameters() if p.requires_grad) == [str(k) for k in g[f"{variant}/requires_grad"]]
    sd = {k: torch.from_numpy(np.array(g[f"{variant}/sd/{k}"])) for k in keys}
    res = bmodel.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    bmodel = bmodel.cuda()
    S, base = int(g[f"{variant}/S"]), int(g[f"{variant}/base"])
    x = torch.from_numpy(g[f"{variant}/x"]).cuda()
    bf.set_compute_dtype(dtype)
    try:
        bf.manual_seed(SEED, next_sample=base)
        with torch.no_grad():
            raw, mean, lp, lq = sample_bayesian(bmodel, x, S)
        lps = bmodel.log_prob_samples().cpu().numpy()
        y = raw[0].float().cpu().numpy()
        ref = g[f"{variant}/y"]
        assert np.abs(y - ref).max() <= tol * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(lps[:, 0], g[f"{variant}/log_prior"], rtol=2e-6)
        np.testing.assert_allclose(lps[:, 1], g[f"{variant}/lvp"], rtol=2e-6)
        # the reference took its state dict after the forward of sample base + S - 1: one such forward here, and this
        # package's state_dict() must carry the same per-layer log-prob scalars (fresh, not those of an older forward)
        bf.manual_seed(SEED, next_sample=base + S - 1)
        with torch.no_grad():
            bmodel(x)
        mine = bmodel.state_dict()
        for k in keys:
            if k.endswith("log_prior") or k.endswith("log_variational_posterior"):
                assert float(mine[k]) == pytest.approx(float(sd[k]), rel=5e-6), k
            else:
                assert torch.equal(mine[k].cpu(), sd[k]), k
    finally:
        bf.set_compute_dtype("bf16")


def test_state_dict_round_trip_through_a_file(tmp_path):
    """torch.save / torch.load of this package's own state dict (what a training script does), then the same outputs."""
    torch.manual_seed(5)
    a = bf.to_bayesian(_net(), delta=0.05).cuda()
    x = torch.randn(8, 48, device="cuda")
    bf.manual_seed(SEED)
    with torch.no_grad():
        ya = a(x)
    path = tmp_path / "ckpt.pth"
    torch.save({"model": a.state_dict(), "delta": 0.05}, path)
    torch.manual_seed(6)
    b = bf.to_bayesian(_net(), delta=0.05).cuda()
    b.load_state_dict(torch.load(path)["model"])
    bf.manual_seed(SEED)
    with torch.no_grad():
        yb = b(x)
    assert torch.equal(ya, yb)
    assert float(b.log_prior()) == float(a.log_prior())
    assert float(b.state_dict()["model.0.log_prior"]) == float(a.state_dict()["model.0.log_prior"]) != 0.0


def test_loaded_logprob_scalars_survive_an_earlier_forward():
    """load_state_dict after a forward: the layer's lazily refreshed log-prob scalars must be the checkpoint's, not the
    pre-load forward's (nn/layers/base.py::_load_from_state_dict), until the next forward replaces them."""
    torch.manual_seed(5)
    bmodel = bf.to_bayesian(_net(), delta=0.05).cuda()
    x = torch.randn(8, 48, device="cuda")
    bf.manual_seed(SEED)
    with torch.no_grad():
        bmodel(x)                                   # leaves a pending refresh in every layer
    sd = {k: v.clone() for k, v in bmodel.state_dict().items()}
    first = next(k for k in sd if k.endswith("0.log_prior"))
    sd[first] = torch.tensor(-123.5)
    sd[first.replace("log_prior", "log_variational_posterior")] = torch.tensor(77.25)
    with torch.no_grad():
        bmodel(x)                                   # a later forward: pending refresh with OTHER values
    bmodel.load_state_dict(sd, strict=True)
    layer = bmodel.model[0]
    assert float(layer.log_prior) == -123.5 and float(layer.log_variational_posterior) == 77.25
    assert float(bmodel.state_dict()[first]) == -123.5
    with torch.no_grad():
        bmodel(x)
    assert float(layer.log_prior) != -123.5       # the next forward owns the attributes again
