"""The S-sample harness and its S-sharding over ranks, on CPU: gloo, world_size 2 (the HIP forward itself is
replaced by a stub that is a pure function of the GLOBAL sample index, which is exactly the property the
Philox contract gives the real kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bayeformers_amd.nn as bnn
from bayeformers_amd import random as bfr
from bayeformers_amd.sampling import elbo, repeat_inputs, sample_bayesian


class StubModel(bnn.Model):
    """forward(x[S*B, F]) -> logits[S*B, 3] that depend on (global sample index, x); log-probs likewise."""

    def forward(self, x):
        ctx = bfr.STATE.ctx
        S, base = ctx.S, ctx.sample_base
        B = x.shape[0] // S
        idx = (base + torch.arange(S, dtype=torch.float64)).repeat_interleave(B)
        self._lp_buf = torch.stack([-(base + torch.arange(S, dtype=torch.float64)) ** 2,
                                    3.0 * (base + torch.arange(S, dtype=torch.float64))], dim=1)[None]
        return torch.stack([x.double().sum(1) * idx, idx, idx ** 2], dim=1).float()


def expected(x, S, base=0):
    s = np.arange(base, base + S, dtype=np.float64)
    xs = x.double().sum(1).numpy()
    logits = np.stack([np.outer(s, xs), np.repeat(s[:, None], len(xs), 1), np.repeat(s[:, None] ** 2, len(xs), 1)], -1)
    return logits, (-(s ** 2)).mean(), (3 * s).mean()


def test_repeat_inputs_is_sample_major():
    d = repeat_inputs({"a": torch.arange(6).view(3, 2), "n": 5}, 2)
    assert d["n"] == 5 and torch.equal(d["a"], torch.arange(6).view(3, 2).repeat(2, 1))


def test_single_process_harness():
    bfr.manual_seed(1)
    x = torch.randn(4, 5)
    raw, mean, lp, lq = sample_bayesian(StubModel(), x, 6)
    logits, elp, elq = expected(x, 6)
    assert raw[0].shape == (6, 4, 3)
    np.testing.assert_allclose(raw[0].numpy(), logits, rtol=1e-5)
    np.testing.assert_allclose(mean[0].numpy(), logits.mean(0), rtol=1e-5)
    assert float(lp) == pytest.approx(elp) and float(lq) == pytest.approx(elq)
    # the second step uses the next 6 sample indices
    raw2, _, lp2, _ = sample_bayesian(StubModel(), x, 6)
    assert float(lp2) == pytest.approx(expected(x, 6, 6)[1])
    assert float(elbo(lp, lq, torch.tensor(2.0), 10)) == pytest.approx((elq - elp) / 10 + 2.0)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bfr.manual_seed(1)
        torch.manual_seed(0)
        x = torch.randn(4, 5)
        raw, mean, lp, lq = sample_bayesian(StubModel(), x, 6, gather_raw=True)
        raw_local, _, _, _ = sample_bayesian(StubModel(), x, 6)
        q.put((rank, raw[0].numpy(), mean[0].numpy(), float(lp), float(lq), raw_local[0].numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    x = torch.randn(4, 5)
    logits, elp, elq = expected(x, 6)
    for rank, raw, mean, lp, lq, raw_local in res:
        np.testing.assert_allclose(raw, logits, rtol=1e-5)            # all-gathered per-sample outputs, in order
        np.testing.assert_allclose(mean, logits.mean(0), rtol=1e-5)   # all-reduced mean == single-process mean
        assert lp == pytest.approx(elp) and lq == pytest.approx(elq)
        # second step: this rank ran global samples [6 + 3*rank, 6 + 3*rank + 3)
        np.testing.assert_allclose(raw_local, expected(x, 3, 6 + 3 * rank)[0], rtol=1e-5)


def test_repeated_inputs_are_cached_until_the_input_changes():
    """repeat_inputs keeps the S-fold copies of resident inputs between steps (no_grad): same tensor object, unchanged
    -> the same repeated tensor; modified in place, another tensor or another S -> repeated again."""
    from bayeformers_amd.sampling import repeat_inputs

    x = torch.arange(12).reshape(3, 4)
    with torch.no_grad():
        a = repeat_inputs({"input_ids": x}, 5)["input_ids"]
        b = repeat_inputs({"input_ids": x}, 5)["input_ids"]
        assert a is b and a.shape == (15, 4) and torch.equal(a[3:6], x)
        x[0, 0] = 99                                   # in-place edit bumps the version counter
        c = repeat_inputs({"input_ids": x}, 5)["input_ids"]
        assert c is not a and int(c[0, 0]) == 99 and int(c[3, 0]) == 99
        d = repeat_inputs({"input_ids": x}, 2)["input_ids"]
        assert d.shape == (6, 4)
        assert repeat_inputs(x, 1) is x
    y = torch.ones(2, 3, requires_grad=True)
    r = repeat_inputs(y, 3)                            # differentiable inputs are never cached
    assert r.requires_grad and r.shape == (6, 3)
