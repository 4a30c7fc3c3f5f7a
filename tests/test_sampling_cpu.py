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


def _worker(rank, world, port, q, S=6):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bfr.manual_seed(1)
        torch.manual_seed(0)
        x = torch.randn(4, 5)
        raw, mean, lp, lq = sample_bayesian(StubModel(), x, S, gather_raw=True)
        raw_local, _, _, _ = sample_bayesian(StubModel(), x, S)
        q.put((rank, raw[0].numpy(), mean[0].numpy(), float(lp), float(lq), raw_local[0].numpy()))
    finally:
        dist.destroy_process_group()


def _run_ranks(world, S):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, S)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_shard_span_partitions_the_samples():
    from bayeformers_amd.sampling import shard_span

    assert [shard_span(10, r, 8) for r in range(8)] == [(0, 2), (2, 2), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1)]
    assert [shard_span(64, r, 8) for r in range(8)] == [(8 * r, 8) for r in range(8)]
    for S, G in [(10, 3), (5, 8), (1, 4), (7, 7), (13, 5)]:
        spans = [shard_span(S, r, G) for r in range(G)]
        assert sum(c for _, c in spans) == S and spans[0][0] == 0
        assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(G - 1))
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_monte_carlo_span_checks_its_range():
    m = StubModel()
    with pytest.raises(ValueError):
        with m.monte_carlo(3, span=(8, 10)):
            pass
    with m.monte_carlo(2, span=(8, 10)):
        assert m._mc_span == (8, 10)
    with m.monte_carlo(4, shard=(1, 2)):
        assert m._mc_span == (4, 8)


@pytest.mark.parametrize("world,S", [(3, 10), (3, 2)])
def test_uneven_shards_match_single_process(world, S):
    """S = 10 over 3 ranks (4, 3, 3 samples) — BASELINE config 5's `S = 10 on 8 GPUs` in small — and S = 2 over 3 ranks
    (one rank idles): all-gathered per-sample outputs in global order, all-reduced means and log-probs equal to the
    single-process run; the second step starts at global sample S on every rank."""
    from bayeformers_amd.sampling import shard_span

    res = _run_ranks(world, S)
    torch.manual_seed(0)
    x = torch.randn(4, 5)
    logits, elp, elq = expected(x, S)
    for rank, raw, mean, lp, lq, raw_local in res:
        assert raw.shape[0] == S
        np.testing.assert_allclose(raw, logits, rtol=1e-5)
        np.testing.assert_allclose(mean, logits.mean(0), rtol=1e-5)
        assert lp == pytest.approx(elp) and lq == pytest.approx(elq)
        start, count = shard_span(S, rank, world)
        assert raw_local.shape[0] == count
        if count:
            np.testing.assert_allclose(raw_local, expected(x, count, S + start)[0], rtol=1e-5)


def test_two_rank_sharding_matches_single_process():
    res = _run_ranks(2, 6)
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


def test_graphed_sampler_refuses_what_it_cannot_replay():
    """No GPU needed: a model in training mode (per-step dropout masks) and host-resident inputs are refused before
    anything is captured; the sample counter stays on the host."""
    from bayeformers_amd.sampling import GraphedSampler

    x = torch.randn(4, 5)
    with pytest.raises(RuntimeError, match="training mode"):
        GraphedSampler(StubModel().train(), x, 3)
    with pytest.raises(RuntimeError, match="on the GPU"):
        GraphedSampler(StubModel().eval(), x, 3)
    with pytest.raises(RuntimeError, match="on the GPU"):
        GraphedSampler(StubModel().eval(), {"n": 5}, 3)
    assert bfr.STATE.device_counter is None


def test_local_step_and_finish_are_sample_bayesian():
    """sample_bayesian = _local_step + _finish_step (what GraphedSampler captures and what it runs after the replay)."""
    from bayeformers_amd.sampling import _finish_step, _local_step

    m, x = StubModel().eval(), torch.randn(4, 5)
    bfr.manual_seed(1, next_sample=7)
    with torch.no_grad():
        raw, mean, lp, lq = sample_bayesian(m, x, 3)
    bfr.manual_seed(1, next_sample=7)
    with torch.no_grad():
        raw2, sizes, local = _local_step(m, repeat_inputs(x, 3), 3, None, 0, 1, repeated=True)
        mean2, lp2, lq2 = _finish_step(raw2, sizes, local, 3, None, False)
    assert torch.equal(raw[0], raw2[0]) and torch.equal(mean[0], mean2[0]) and lp == lp2 and lq == lq2
    assert len(local) == 1 and local[0].dtype == torch.float64 and local[0].numel() == sizes[0] + 2


def test_select_key_recognises_a_selection_only_by_value():
    """the cache key of a `select` callable must not rest on id()s of captured objects (an id is reused once its
    object is gone; a captured list can be mutated in place)."""
    import functools

    from bayeformers_amd.sampling import _select_key

    def pick(i):
        return lambda out: (out[i],)

    assert _select_key(pick(1)) == _select_key(pick(1)) and _select_key(pick(1)) != _select_key(pick(2))
    assert _select_key(None) == _select_key(None)
    idx = [0]
    assert _select_key(lambda out: (out[idx[0]],)) is None       # a captured list: could be mutated between calls
    t = torch.zeros(1)
    assert _select_key(lambda out: (out + t,)) is None            # a captured tensor

    def plain(out):
        return (out,)

    assert _select_key(plain) == _select_key(plain)
    assert _select_key(functools.partial(plain)) == _select_key(functools.partial(plain))
    assert _select_key(functools.partial(pick, 1)) == _select_key(functools.partial(pick, 1))
    assert _select_key(functools.partial(pick, [1])) is None

    def outer():
        def inner(out):
            return late(out)   # noqa: F821 - a cell that is still empty when the key is taken
        key = _select_key(inner)
        late = plain   # noqa: F841
        return key

    assert outer() is None
