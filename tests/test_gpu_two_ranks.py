"""The S-sharded harness and training step with the REAL HIP kernels under torch.distributed: two ranks that share the one
GPU of the test box (gloo backend on CUDA tensors — RCCL refuses two ranks on one device; what differs from an N-GPU run is
only the transport of the collectives).  Rank r runs its slice of the Monte-Carlo samples through the fused path; the
all-reduced means, the all-gathered per-sample outputs, and — for the training step — the bucketed, all-reduced gradients
and the updated parameters must equal the single-process run's."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def _build():
    import bayeformers_amd as bf
    from transformers import BertConfig, BertForSequenceClassification

    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, vocab_size=1000,
                     max_position_embeddings=64)
    torch.manual_seed(0)
    model = BertForSequenceClassification(cfg).eval()
    bmodel = bf.to_bayesian(model, delta=0.05, freeze=True).eval().cuda().to(torch.bfloat16)
    assert bf.fuse_activations(bmodel) == 2 and bf.fuse_residual_layernorm(bmodel) == 4
    assert bf.fuse_shared_inputs(bmodel) == 2 and bf.fuse_attention(bmodel) and bf.fuse_embeddings(bmodel) == 1
    torch.manual_seed(7)
    ids = torch.randint(0, cfg.vocab_size, (4, 32)).cuda()
    labels = torch.randint(0, 2, (4,)).cuda()
    inputs = {"input_ids": ids, "attention_mask": torch.ones(4, 32, dtype=torch.long, device="cuda")}
    return bmodel, inputs, labels


def _run(S, steps):
    import bayeformers_amd as bf
    from bayeformers_amd.sampling import sample_bayesian
    from bayeformers_amd.training import GradientBuckets, training_step

    bmodel, inputs, labels = _build()
    bf.set_compute_dtype("bf16")
    bf.manual_seed(SEED)
    with torch.no_grad():
        raw, mean, lp, lq = sample_bayesian(bmodel, inputs, S, gather_raw=True)
    fwd = (raw[0].float().cpu().numpy(), mean[0].float().cpu().numpy(), float(lp), float(lq))
    params = [p for p in bmodel.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.5)
    buckets = GradientBuckets(params, bucket_bytes=1 << 20)
    grads = None
    for _ in range(steps):
        loss = training_step(bmodel, inputs, S, lambda m: torch.nn.functional.cross_entropy(m[0].float(), labels), opt,
                             n_batches=100, buckets=buckets, max_grad_norm=1.0)
        if grads is None:
            grads = {n: p.grad.detach().float().cpu().numpy().copy() for n, p in bmodel.named_parameters() if p.grad is not None}
    after = {n: p.detach().float().cpu().numpy().copy() for n, p in bmodel.named_parameters() if p.requires_grad}
    return fwd, float(loss), grads, after


def _run_graphed(S, replays):
    """The forward harness through GraphedSampler: `replays` steps of S samples, fresh epsilon each."""
    import bayeformers_amd as bf
    from bayeformers_amd.sampling import GraphedSampler

    bmodel, inputs, _ = _build()
    bf.set_compute_dtype("bf16")
    sampler = GraphedSampler(bmodel, inputs, S)
    bf.manual_seed(SEED)
    out = []
    for _ in range(replays):
        raw, mean, lp, lq = sampler()
        out.append((mean[0].float().cpu().numpy().copy(), float(lp), float(lq)))
    sampler.close()
    return out


def _worker(rank, world, port, S, steps, q, rccl=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    if rccl:  # one rank per GPU, the process group bench.py builds
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + (_run(S, steps) if steps > 0 else (_run_graphed(S, -steps),)))
    finally:
        dist.destroy_process_group()


def _gpus_visible() -> int:
    # (device_count does not initialise the HIP runtime in the parent; the ranks are spawned processes)
    return torch.cuda.device_count()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("rccl", [False, pytest.param(True, marks=pytest.mark.skipif(
    _gpus_visible() < 2, reason="RCCL wants one GPU per rank: needs at least two visible GPUs"))])
def test_two_ranks_on_one_gpu_match_single_process(rccl):
    """rccl=False: both ranks share GPU 0 over gloo (runs on the one-GPU test box).  rccl=True: one rank per GPU over RCCL —
    the transport `bench.py --gpus N` and a real N-GPU training job use; skipped where fewer than two GPUs are visible."""
    S, steps, world = 5, 2, 2  # uneven shards: 3 + 2 samples
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, steps, q, rccl)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    deadline = time.monotonic() + 500
    while len(res) < len(procs):  # a rank that died must fail the test at once, not after the timeout
        try:
            res.append(q.get(timeout=2))
        except Exception:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"a rank exited with {dead}"
            assert time.monotonic() < deadline, "timed out"
    res = sorted(res, key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    fwd, loss, grads, after = _run(S, steps)
    for rank, f, l, g, a in res:
        # forward: every sample's outputs are bit-identical whatever rank ran it; the reduced quantities agree to rounding
        assert np.array_equal(f[0], fwd[0]), rank
        np.testing.assert_allclose(f[1], fwd[1], rtol=1e-2, atol=1e-3)
        assert f[2] == pytest.approx(fwd[2], rel=1e-12) and f[3] == pytest.approx(fwd[3], rel=1e-12)
        assert l == pytest.approx(loss, rel=1e-5)
        # training: the ranks' summed gradients = the single-process gradients (bf16 activations: the per-rank partial
        # sums are rounded separately), and every rank holds the same updated parameters
        assert g.keys() == grads.keys()
        for n in grads:
            scale = np.abs(grads[n]).max() + 1e-30
            assert np.abs(g[n] - grads[n]).max() <= 3e-2 * scale, (rank, n)
        for n in after:
            assert np.array_equal(a[n], res[0][4][n]), (rank, n)


@pytest.mark.timeout(600)
def test_graphed_sampler_on_two_ranks_matches_single_process():
    """GraphedSampler under an S-shard group: every rank replays its own slice (3 + 2 samples) from its HIP graph, the
    step's collective runs eagerly after the replay; three replays = the three eager steps of one process."""
    S, replays, world = 5, 3, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, -replays, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    deadline = time.monotonic() + 500
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=2))
        except Exception:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"a rank exited with {dead}"
            assert time.monotonic() < deadline, "timed out"
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    import bayeformers_amd as bf
    from bayeformers_amd.sampling import sample_bayesian

    bmodel, inputs, _ = _build()
    bf.set_compute_dtype("bf16")
    bf.manual_seed(SEED)
    with torch.no_grad():
        want = [sample_bayesian(bmodel, inputs, S) for _ in range(replays)]
    for rank, got in res:
        for k in range(replays):
            np.testing.assert_allclose(got[k][0], want[k][1][0].float().cpu().numpy(), rtol=1e-2, atol=1e-3)
            assert got[k][1] == pytest.approx(float(want[k][2]), rel=1e-12)
            assert got[k][2] == pytest.approx(float(want[k][3]), rel=1e-12)


def test_bench_strong_scaling_shards_one_total_unevenly():
    """`bench.py --strong --samples 5 --gpus 2` (BASELINE configs[4]'s kind of split: the step's samples are a TOTAL,
    sharded 3 + 2) must report the same ELBO as the one-rank run of the same 5 samples, and say what it did."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--workload", "mlp", "--samples", "5", "--steps", "2", "--warmup", "1",
            "--no-traffic", "--no-cpu-baseline", "--graph", "off"]  # a captured step runs extra steps before the timed ones

    def run(extra, env_extra):
        env = dict(os.environ, **env_extra)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        out = subprocess.run(base + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    one = run([], {})
    two = run(["--gpus", "2", "--strong"], {"BF_BENCH_SHARE_GPU": "1", "BF_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and one["scaling"] == "weak" and one["config"]["samples_per_step"] == 5
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["samples_per_gpu"] == [3, 2] and two["config"]["samples_per_step"] == 5
    assert two["config"]["samples_total"] == 10
    assert abs(two["config"]["last_elbo"] - one["config"]["last_elbo"]) <= 1e-9 * abs(one["config"]["last_elbo"])


def _bench_line(args, env_extra):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + ["--no-traffic", "--no-cpu-baseline"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


SHARE = {"BF_BENCH_SHARE_GPU": "1", "BF_BENCH_BACKEND": "gloo"}


def test_bench_two_ranks_bert_base_forward_through_graphed_sampler():
    """the BERT-base forward workload through bench.py's own launcher on two ranks (sharing this GPU, gloo on
    the CUDA tensors) — strong scaling, S = 3 as shards of 2 + 1, each rank replaying ITS shard from a GraphedSampler graph and
    the packed ELBO all-reduce run eagerly after the replay — must report the one-rank ELBO of the same 3 samples."""
    one = _bench_line(["--workload", "bert_base", "--samples", "3", "--steps", "2", "--warmup", "1"], {})
    two = _bench_line(["--workload", "bert_base", "--samples", "3", "--steps", "2", "--warmup", "1", "--gpus", "2", "--strong"], SHARE)
    assert one["config"]["hip_graph"] == "GraphedSampler" and two["config"]["hip_graph"] == "GraphedSampler"
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["samples_per_gpu"] == [2, 1]
    assert two["config"]["preflight"]["ranks_counted"] == 2 and len(two["roofline"]["by_rank"]["ms_per_step"]) == 2
    assert two["config"]["last_elbo"] == pytest.approx(one["config"]["last_elbo"], rel=1e-7)


def test_bench_two_ranks_bert_base_training_step_through_gradient_buckets():
    """... and the training workload: two ranks of 2 samples each (weak scaling: 4 samples per step), every rank backpropagating
    its samples, the gradients all-reduced in buckets under backward (training.GradientBuckets), against one rank running the
    same 4 samples: the loss after three optimizer steps agrees (gradient sums in another order: not bit for bit)."""
    one = _bench_line(["--workload", "bert_base_train", "--samples", "4", "--steps", "2", "--warmup", "1", "--graph", "off"], {})
    two = _bench_line(["--workload", "bert_base_train", "--samples", "2", "--steps", "2", "--warmup", "1", "--gpus", "2"], SHARE)
    assert one["config"]["samples_per_step"] == 4 and two["config"]["samples_per_step"] == 4 and two["n_gpus"] == 2
    assert two["config"]["hip_graph"] is False           # S-sharded ranks run the eager step
    assert two["metric"].endswith("(fwd+ELBO+backward+AdamW)")
    assert two["config"]["last_elbo"] == pytest.approx(one["config"]["last_elbo"], rel=1e-6)
