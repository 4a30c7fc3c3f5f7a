"""GraphedSampler: the harness step replayed from a HIP graph must be the eager step — replay k draws the epsilon of
the k-th eager `sample_bayesian` call (device-resident sample counter), new batches are copied into the captured buffers,
and closing it hands the counter back to the host."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 0xBEEF


def _build(half=True):
    import bayeformers_amd as bf
    from transformers import BertConfig, BertForSequenceClassification

    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, vocab_size=1000,
                     max_position_embeddings=64)
    torch.manual_seed(0)
    model = BertForSequenceClassification(cfg).eval()
    bmodel = bf.to_bayesian(model, delta=0.05, freeze=True).eval().cuda()
    if half:
        bmodel = bmodel.to(torch.bfloat16)
    bf.fuse_activations(bmodel), bf.fuse_residual_layernorm(bmodel), bf.fuse_shared_inputs(bmodel)
    bf.fuse_attention(bmodel), bf.fuse_embeddings(bmodel)
    g = torch.Generator().manual_seed(7)
    batches = []
    for _ in range(2):
        ids = torch.randint(0, cfg.vocab_size, (4, 32), generator=g).cuda()
        batches.append({"input_ids": ids, "attention_mask": torch.ones(4, 32, dtype=torch.long, device="cuda")})
    return bmodel, batches


def _host(res):
    raw, mean, lp, lq = res
    return raw[0].float().cpu().numpy().copy(), mean[0].float().cpu().numpy().copy(), float(lp), float(lq)


@pytest.mark.parametrize("S", [1, 3])
def test_replays_are_the_eager_steps(S):
    import bayeformers_amd as bf
    from bayeformers_amd import random as bfr
    from bayeformers_amd.sampling import GraphedSampler, sample_bayesian

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    order = [0, 0, 1, 0]  # the batch fed at each step
    bf.manual_seed(SEED)
    with torch.no_grad():
        eager = [_host(sample_bayesian(bmodel, batches[b], S)) for b in order]
    assert bfr.STATE.device_counter is None and bfr.get_state()[1] == len(order) * S

    sampler = GraphedSampler(bmodel, batches[0], S)
    assert bfr.STATE.device_counter is not None
    bf.manual_seed(SEED)  # rewinds the device-resident counter too
    for step, b in enumerate(order):
        got = _host(sampler(batches[b]) if step else sampler())
        want = eager[step]
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), step
        assert got[2] == want[2] and got[3] == want[3], step
    # a batch of another shape is refused, not silently truncated
    with pytest.raises(ValueError, match="differs from the captured one"):
        sampler({k: v[:2] for k, v in batches[0].items()})
    sampler.close()
    assert bfr.STATE.device_counter is None and bfr.get_state()[1] == len(order) * S  # the counter came back, advanced
    with torch.no_grad():
        nxt = _host(sample_bayesian(bmodel, batches[0], S))
    assert not np.array_equal(nxt[0], eager[0][0])  # fresh sample indices, not a rewind
    with pytest.raises(RuntimeError, match="closed"):
        sampler()


def test_training_mode_is_refused():
    from bayeformers_amd.sampling import GraphedSampler

    bmodel, batches = _build()
    bmodel.train()
    with pytest.raises(RuntimeError, match="training mode"):
        GraphedSampler(bmodel, batches[0], 2)


def test_two_samplers_share_the_device_counter_until_the_last_one_closes():
    """The captured kernels hold the counter's address: closing one sampler must not move the counter back to the host
    while another one is open; the two draw from ONE sequence of sample indices, like two eager callers."""
    import bayeformers_amd as bf
    from bayeformers_amd import random as bfr
    from bayeformers_amd.sampling import GraphedSampler, sample_bayesian

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    bf.manual_seed(SEED)
    with torch.no_grad():
        eager = [_host(sample_bayesian(bmodel, batches[0], 2)), _host(sample_bayesian(bmodel, batches[1], 3)),
                 _host(sample_bayesian(bmodel, batches[1], 3))]
    a = GraphedSampler(bmodel, batches[0], 2)
    b = GraphedSampler(bmodel, batches[1], 3)
    bf.manual_seed(SEED)
    got_a = _host(a())
    a.close()
    assert bfr.STATE.device_counter is not None  # b is still open
    got_b = [_host(b()), _host(b())]
    b.close()
    assert bfr.STATE.device_counter is None and bfr.get_state()[1] == 2 + 3 + 3
    for got, want in zip([got_a] + got_b, eager):
        assert np.array_equal(got[0], want[0]) and got[2] == want[2] and got[3] == want[3]


def test_sample_bayesian_graph_flag_caches_two_signatures():
    """sample_bayesian(graph=True): the evaluation-loop form — a sampler per batch signature, the last two kept with the model;
    a replay equals the eager call from the same counter, needs no_grad."""
    import bayeformers_amd as bf
    from bayeformers_amd import random as bfr
    from bayeformers_amd.sampling import sample_bayesian

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    with pytest.raises(RuntimeError, match="no_grad"):
        sample_bayesian(bmodel, batches[0], 2, graph=True)
    from bayeformers_amd import sampling

    with torch.no_grad():
        sample_bayesian(bmodel, batches[0], 2, graph=True)  # captures
        cache = sampling.graphed_samplers(bmodel)
        first = cache[0][1]
        bf.manual_seed(SEED)
        got = _host(sample_bayesian(bmodel, batches[1], 2, graph=True))  # same signature: the cached sampler, new batch
        assert len(cache) == 1 and cache[0][1] is first
        bf.manual_seed(SEED)
        want = _host(sample_bayesian(bmodel, batches[1], 2))
        assert np.array_equal(got[0], want[0]) and got[2] == want[2] and got[3] == want[3]
        small = {k: v[:2].clone() for k, v in batches[0].items()}
        tiny = {k: v[:1].clone() for k, v in batches[0].items()}
        sample_bayesian(bmodel, small, 2, graph=True)
        sample_bayesian(bmodel, tiny, 2, graph=True)
        assert len(cache) == 2 and first.graph is None  # the oldest signature was closed
        for _, s in cache:
            s.close()
        cache.clear()
    assert bfr.STATE.device_counter is None


def test_graph_follows_seed_dtype_and_starts_at_the_callers_sample_index():
    """the seed is a kernel argument, the compute dtype and the sampling plan are fixed at capture — a replay after
    `manual_seed(other)` / `set_compute_dtype` must not keep drawing from the old key or dtype; and the capture's warm-up
    steps must not eat sample indices: `manual_seed(s); sample_bayesian(graph=True)` IS the eager call from that state."""
    import copy

    import bayeformers_amd as bf
    from bayeformers_amd import random as bfr
    from bayeformers_amd import sampling
    from bayeformers_amd.sampling import sample_bayesian

    bmodel, batches = _build(half=False)   # fp32 activations go with every compute dtype
    bf.set_compute_dtype("bf16")
    S = 2
    pick = lambda: (lambda out: (out.logits,))   # a select written at the call site: a new object per call
    try:
        with torch.no_grad():
            for step, (seed, dtype) in enumerate([(SEED, "bf16"), (SEED + 1, "bf16"), (SEED + 1, "fp16"), (SEED, "bf16")]):
                bf.set_compute_dtype(dtype)
                bf.manual_seed(seed)
                want = [_host(sample_bayesian(bmodel, batches[0], S)), _host(sample_bayesian(bmodel, batches[1], S))]
                bf.manual_seed(seed)   # NO extra rewind after the capture: the first graphed call starts here
                got0 = sample_bayesian(bmodel, batches[0], S, select=pick(), graph=True)
                mean0 = got0[1][0]
                got0 = _host(got0)
                got1 = _host(sample_bayesian(bmodel, batches[1], S, select=pick(), graph=True))
                for got, w in zip((got0, got1), want):
                    assert np.array_equal(got[0], w[0]) and np.array_equal(got[1], w[1]), (step, seed, dtype)
                    assert got[2] == w[2] and got[3] == w[3], (step, seed, dtype)
                # the convenience path returns copies of the means: the second call did not overwrite the first's
                assert np.array_equal(mean0.float().cpu().numpy(), want[0][1])
                cache = sampling.graphed_samplers(bmodel)
                assert len(cache) == 1 and cache[0][1].captures == step + 1   # one sampler, captured again per change
        # the model carries no graph: it can be copied while samplers exist
        assert "_bf_graphed" not in bmodel.__dict__
        clone = copy.deepcopy(bmodel)
        assert sum(p.numel() for p in clone.parameters()) == sum(p.numel() for p in bmodel.parameters())
    finally:
        bf.set_compute_dtype("bf16")
        for _, sm in sampling.graphed_samplers(bmodel):
            sm.close()
    assert bfr.STATE.device_counter is None


def test_model_call_replays_the_reference_loop_and_equals_eager():
    """bnn.Model.__call__ under no_grad in eval mode, one sample per call (the reference's caller loop): the third call of a
    signature captures, later ones replay — call k is the k-th eager call bit for bit, new batch VALUES are copied in, the
    outputs are copies, log_prior() / the layers' lazy scalars follow the replays, another signature runs eagerly."""
    import bayeformers_amd as bf
    from bayeformers_amd import random as bfr

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    order = [0, 1, 0, 1, 1, 0]
    first = bmodel.fused_children()[0]

    def loop(replay):
        bmodel.graph_replay = replay
        bf.manual_seed(SEED)
        outs = []
        with torch.no_grad():
            for b in order:
                o = bmodel(**batches[b])
                outs.append((o.logits, float(bmodel.log_prior()), float(bmodel.log_variational_posterior()),
                             float(first.log_prior), float(first.log_variational_posterior)))
        return outs

    try:
        eager = loop(False)
        assert not bmodel._graphs.forwards and bfr.STATE.device_counter is None
        got = loop(True)
        assert len(bmodel._graphs.forwards) == 1 and bfr.STATE.device_counter is not None
        # every forward took one dropout call number, replayed or not, and the capture's warm-up forward gave its own back: the
        # counter stands where the all-eager history left it (a training step that follows draws the same masks either way)
        assert int(bfr.STATE.device_drop_counter.item()) == len(order)
        for k, (e, r) in enumerate(zip(eager, got)):
            assert torch.equal(e[0], r[0]), k
            assert e[1:] == r[1:], k
        assert len({o[0].data_ptr() for o in got}) == len(got)   # every call handed out its own copy
        # a batch of another shape is another signature: eager until seen twice, the captured one stays
        small = {k: v[:2].clone() for k, v in batches[0].items()}
        with torch.no_grad():
            bmodel(**small)
        assert len(bmodel._graphs.forwards) == 1
        # training mode and recorded gradients never replay
        with torch.no_grad():
            bmodel.train()
            bmodel(**batches[0])
            bmodel.eval()
        n = bmodel._graphs.forwards[0][1].captures
        bmodel(**batches[0])   # gradients enabled: eager
        assert bmodel._graphs.forwards[0][1].captures == n
        # another seed: captured again, and still the eager result
        bf.manual_seed(SEED + 1)
        with torch.no_grad():
            r2 = bmodel(**batches[0]).logits
        assert bmodel._graphs.forwards[0][1].captures == n + 1
        bmodel.graph_replay = False
        bf.manual_seed(SEED + 1)
        with torch.no_grad():
            assert torch.equal(bmodel(**batches[0]).logits, r2)
    finally:
        bmodel.graph_replay = True
        bmodel._graphs.close()
    assert bfr.STATE.device_counter is None


def test_dropped_model_releases_its_graphs():
    """the samplers kept for `sample_bayesian(graph=True)` (and the forwards __call__ captured) must not keep their
    model alive: `del model; gc.collect()` ends them, and the sample counter returns to the host."""
    import gc
    import weakref

    import bayeformers_amd as bf
    from bayeformers_amd import random as bfr
    from bayeformers_amd.sampling import sample_bayesian

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    bf.manual_seed(SEED)
    with torch.no_grad():
        sample_bayesian(bmodel, batches[0], 2, graph=True)
        for _ in range(3):
            bmodel(**batches[0])
    assert bfr.STATE.device_counter is not None
    assert len(bmodel._graphs.samplers) == 1 and len(bmodel._graphs.forwards) == 1
    ref = weakref.ref(bmodel)
    del bmodel
    gc.collect()
    assert ref() is None
    assert bfr.STATE.device_counter is None and bfr.get_state()[1] == 2 + 3


def test_model_call_stops_capturing_when_signatures_cycle():
    """A loop over more batch shapes than a model keeps captured forwards for must not capture for ever: after a few evictions new
    signatures stay eager (with a warning) — and every call still returns what the eager call returns."""
    import warnings

    import bayeformers_amd as bf
    from bayeformers_amd import graphs

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    shapes = [{k: v[:n].clone() for k, v in batches[0].items()} for n in (1, 2, 3, 4)]
    try:
        with torch.no_grad(), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(6):
                for x in shapes:
                    bmodel(**x)
        cache = bmodel._graphs
        assert cache.evictions == graphs.MAX_EVICTIONS and len(cache.forwards) <= graphs.KEEP
        assert any("batch signatures" in str(w.message) for w in caught)
        bmodel.graph_replay = False
        bf.manual_seed(SEED)
        with torch.no_grad():
            want = bmodel(**shapes[3]).logits
        bmodel.graph_replay = True
        bf.manual_seed(SEED)
        with torch.no_grad():
            assert torch.equal(bmodel(**shapes[3]).logits, want)
    finally:
        bmodel._graphs.close()


def test_replay_captured_under_inference_mode_takes_later_batches_under_no_grad():
    """A forward captured while the caller runs under torch.inference_mode() keeps ordinary input buffers: a later call under
    no_grad (or inference mode again) can still copy its batch in, and both equal the eager results."""
    import bayeformers_amd as bf
    from bayeformers_amd.sampling import GraphedSampler

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    try:
        bmodel.graph_replay = False
        bf.manual_seed(SEED)
        with torch.no_grad():
            want = [bmodel(**batches[b]).logits.clone() for b in (0, 0, 0, 1, 1)]
        bmodel.graph_replay = True
        bf.manual_seed(SEED)
        got = []
        with torch.inference_mode():
            for b in (0, 0, 0):          # the third call captures, inside inference mode
                got.append(bmodel(**batches[b]).logits.clone())
        with torch.no_grad():
            got.append(bmodel(**batches[1]).logits.clone())
        with torch.inference_mode():
            got.append(bmodel(**batches[1]).logits.clone())
        assert len(bmodel._graphs.forwards) == 1
        for k, (g, w) in enumerate(zip(got, want)):
            assert torch.equal(g, w), k
        with torch.inference_mode():
            sampler = GraphedSampler(bmodel, batches[0], 2)
        with torch.no_grad():
            sampler(batches[1])
        sampler.close()
    finally:
        bmodel._graphs.close()


def test_bench_serial_workload_reports_the_replayed_reference_loop():
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "bert_base_serial", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["hip_graph"] == "bnn.Model.__call__ replay" and "SERIAL" in d["config"]["workload"]
    assert d["config"]["samples_per_step"] == 10 and d["value"] > 0 and d["roofline"]["launches_per_step"] == 480


def test_replay_of_the_reference_call_with_labels():
    """The reference's loop passes the labels with the inputs and reads `model(**inputs)[1]` (examples/bert_glue.py:63): loss and
    logits of a replayed forward equal the eager ones."""
    import bayeformers_amd as bf

    bmodel, batches = _build()
    bf.set_compute_dtype("bf16")
    labels = torch.tensor([0, 1, 1, 0], device="cuda")
    inputs = dict(batches[0], labels=labels)

    def loop(replay):
        bmodel.graph_replay = replay
        bf.manual_seed(SEED)
        with torch.no_grad():
            outs = [bmodel(**inputs) for _ in range(5)]
        return [(o[0].clone(), o[1].clone()) for o in outs]

    try:
        eager, got = loop(False), loop(True)
        assert len(bmodel._graphs.forwards) == 1
        for k, (e, g) in enumerate(zip(eager, got)):
            assert torch.equal(e[0], g[0]) and torch.equal(e[1], g[1]), k
    finally:
        bmodel.graph_replay = True
        bmodel._graphs.close()
