#!/usr/bin/env python3
"""bench.py — MC-samples/sec (fwd+ELBO) of the MI355X Monte-Carlo variational forward path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bert_base|linear768|linear768_m32|mlp]

With --gpus N > 1 and no torch.distributed environment (WORLD_SIZE unset) this process only LAUNCHES the job: it starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>` as a child process (it never touches a
GPU itself), relays the child's output and exits with its return code.  A rank refuses to run (exit 2) when the world
size it finds differs from --gpus, so a line with `"n_gpus": N` always comes from N ranks.

One step = one pass of the hot path over one synthetic batch: S Monte-Carlo samples of the converted model
(sampling + log-probs + MFMA GEMMs for every Bayesian linear, everything else of the wrapped model in torch),
the mean over samples, the NLL on the mean logits and the ELBO scalar read back to the host
(the `sample_bayesian` + loss recipe of /root/reference/examples/bert_glue.py:56-73,:234-235).

Default workload (the one BASELINE.json's metric is quoted on): to_bayesian(BERT-base seq-cls, delta=0.05,
freeze=True), S=10 samples per GPU, B=32, L=128, bf16, random-init weights, synthetic token ids.
N>1 (launched by torch.distributed.run): weak scaling over the sample axis — every rank runs S=10 samples of the
same batch at distinct global sample indices, one RCCL all-reduce of the packed ELBO terms per step.

Prints ONE JSON line on rank 0 with `roofline` (the MFMA GEMM, HIP events on the launch stream) and
`cpu_baseline` (the oracle's CPU port of the reference op sequence, rank 0 at N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = 2500.0   # MI355X dense bf16/fp16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0  # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bert_base", choices=["bert_base", "bert_large_qa", "linear768", "linear768_m32", "mlp", "bert_base_train"])
    ap.add_argument("--samples", type=int, default=None, help="MC samples per GPU per step (default: workload's)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --samples is the TOTAL number of MC samples per step, sharded over the ranks "
                         "(uneven shards allowed: BASELINE.json configs[4] asks for S=10 on 8 GPUs = 2,2,1,1,1,1,1,1); "
                         "the default is weak scaling, --samples per GPU, which is what the driver's scaling run measures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-mode", action="store_true",
                    help="bert_base_train only: put the model in .train() like examples/bert_glue.py:221 (HF dropout p = 0.1 "
                         "active: attention and embeddings take the framework's paths); default: dropout off")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group check only: every rank joins the group (RCCL on GPUs, gloo without), "
                         "one all-reduce, rank 0 prints a JSON line with n_gpus; no kernels run")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="capture the step in a HIP graph (device-resident sample counter); auto = on for the "
                         "launch-bound single-layer / MLP workloads")
    return ap.parse_args()


_STRONG = False  # --strong: S is the total per step


def _world() -> int:
    """What a workload multiplies its S by to get the samples of a step.  Weak scaling (default): S is the number of
    Monte-Carlo samples PER GPU, a step draws S * world samples and sample_bayesian hands every rank its S of them.
    --strong: S is the total, sample_bayesian shards it (unevenly if it must)."""
    return 1 if _STRONG else _ranks()


def _ranks() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class Workload:
    """name, S (per GPU), dtype, step() -> python float (the ELBO), config dict, cpu_baseline() -> dict."""


def build_bert(device, dtype):
    """The benchmarked BERT-base model, exactly as timed: conversion, the four rewrites of the callers around the
    Bayesian layers, bf16, and the synthetic batch.  tests/test_gpu_models.py builds its parity model with this
    function, so what is benchmarked is what is compared with the reference's outputs (tests/golden/bert_c3.npz)."""
    import bayeformers_amd as bf
    from transformers import BertConfig, BertForSequenceClassification

    B, L = 32, 128
    torch.manual_seed(0)
    cfg = BertConfig()
    model = BertForSequenceClassification(cfg).eval()
    bmodel = bf.to_bayesian(model, delta=0.05, freeze=True).eval().to(device)
    n_fused = bf.fuse_activations(bmodel)  # BertIntermediate: dense + exact GELU in one GEMM epilogue
    n_ln = bf.fuse_residual_layernorm(bmodel) if os.environ.get("BF_BENCH_NO_LN_FUSION") is None else 0
    n_qkv = bf.fuse_shared_inputs(bmodel) if os.environ.get("BF_BENCH_NO_QKV_FUSION") is None else 0
    attn = bf.fuse_attention(bmodel) if os.environ.get("BF_BENCH_NO_ATTENTION") is None else False
    n_emb = bf.fuse_embeddings(bmodel) if os.environ.get("BF_BENCH_NO_EMBED_FUSION") is None else 0
    if dtype != "fp32":
        bmodel = bmodel.to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    g = torch.Generator().manual_seed(321)
    ids = torch.randint(0, cfg.vocab_size, (B, L), generator=g)
    labels = torch.randint(0, 2, (B,), generator=g)
    inputs = {"input_ids": ids.to(device), "attention_mask": torch.ones(B, L, dtype=torch.long, device=device)}
    info = {"gelu_fused_into_gemm": n_fused, "residual_layernorm_fused": n_ln, "qkv_in_one_launch": n_qkv,
            "attention_kernel": bool(attn), "embeddings_in_one_launch": n_emb}
    return bmodel, model, inputs, ids, labels, info


def make_bert(device, S, dtype, train=False, train_mode=False):
    from bayeformers_amd.sampling import elbo, sample_bayesian

    B, L, n_batches = 32, 128, 2105  # SST-2: 67,349 train sentences / 32
    bmodel, model, inputs, ids, labels, info = build_bert(device, dtype)
    labels_d = labels.to(device)

    def step():
        with torch.no_grad():
            raw, mean, lp, lq = sample_bayesian(bmodel, inputs, S * _world())
            nll = torch.nn.functional.cross_entropy(mean[0].float(), labels_d)
            return elbo(lp, lq, nll.double(), n_batches)

    if train:
        # the reference's training step (examples/bert_glue.py:227-241) — forward, ELBO, backward through
        # every sampled-weight layer (eps regenerated from the Philox counter), Adam on the unfrozen parameters
        # (examples/bert_glue.py:215 uses transformers' AdamW(lr, eps) with its default weight_decay = 0: torch's AdamW
        # with weight_decay = 0 is the same update; fused = one multi-tensor kernel for all 85 parameter tensors)
        from bayeformers_amd.training import GradientBuckets, training_step

        if train_mode:
            bmodel.train()
        params = [p for p in bmodel.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=2e-5, eps=1e-8, weight_decay=0.0, fused=True)
        # world > 1: flat gradient buffers, all-reduced over the ranks while backward runs
        buckets = GradientBuckets(params) if _ranks() > 1 or os.environ.get("BF_BENCH_TRAIN_BUCKETS") is not None else None

        def nll_fn(mean):
            return torch.nn.functional.cross_entropy(mean[0].float(), labels_d)

        def step():  # noqa: F811
            # examples/bert_glue.py:227-241: forward, ELBO, backward, clip_grad_norm_(1), optimizer step
            return training_step(bmodel, inputs, S * _world(), nll_fn, opt, n_batches, buckets=buckets,
                                 max_grad_norm=None if os.environ.get("BF_BENCH_TRAIN_NO_CLIP") is not None else 1.0)

    def cpu_baseline():
        from oracle.model_oracle import log_probs, to_oracle

        if train:
            return None
        omodel = to_oracle(model, delta=0.05).eval()
        n = 5  # one warm-up, then the mean of >= 5 (about 25 s of CPU work on the GPU box's host)
        with torch.no_grad():
            omodel(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long))  # warm-up sample
            t0 = time.perf_counter()
            for _ in range(n):
                out = omodel(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long))
                log_probs(omodel)
            dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "MC-samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} serial MC samples (fwd + log-probs) of the same BERT-base B=32 L=128 batch, "
                          f"torch-CPU fp32, {dt:.1f}s"}

    cfgd = {"workload": "to_bayesian(BERT-base seq-cls, delta=0.05, freeze=True) " +
                        (("training step: fwd+ELBO+backward+clip+AdamW, " + ("model.train(): dropout 0.1" if train_mode else "dropout off"))
                         if train else "fwd+ELBO"), "samples_per_gpu": S,
            "batch": B, "seq_len": L, "bayesian_linears": len(bmodel.fused_children()), "bayesian_scalars": 85609730}
    cfgd.update(info)
    return step, cpu_baseline, cfgd, bmodel


def make_bert_large_qa(device, S, dtype):
    """BASELINE config 5: to_bayesian(BERT-large QA) SQuAD-shaped forward + ELBO, seq=384, batch=16."""
    import bayeformers_amd as bf
    from bayeformers_amd.sampling import elbo, sample_bayesian
    from transformers import BertConfig, BertForQuestionAnswering

    B, L, n_batches = 16, 384, 5475  # SQuAD v1.1: 87,599 train questions / 16
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    model = BertForQuestionAnswering(cfg).eval()
    bmodel = bf.to_bayesian(model, delta=0.05, freeze=True).eval().to(device)
    n_fused = bf.fuse_activations(bmodel)
    n_ln = bf.fuse_residual_layernorm(bmodel) if os.environ.get("BF_BENCH_NO_LN_FUSION") is None else 0
    n_qkv = bf.fuse_shared_inputs(bmodel) if os.environ.get("BF_BENCH_NO_QKV_FUSION") is None else 0
    attn = bf.fuse_attention(bmodel) if os.environ.get("BF_BENCH_NO_ATTENTION") is None else False
    n_emb = bf.fuse_embeddings(bmodel) if os.environ.get("BF_BENCH_NO_EMBED_FUSION") is None else 0
    if dtype != "fp32":
        bmodel = bmodel.to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    g = torch.Generator().manual_seed(654)
    ids = torch.randint(0, cfg.vocab_size, (B, L), generator=g)
    sp, ep = torch.randint(0, L, (B,), generator=g).to(device), torch.randint(0, L, (B,), generator=g).to(device)
    inputs = {"input_ids": ids.to(device), "attention_mask": torch.ones(B, L, dtype=torch.long, device=device)}

    def step():
        with torch.no_grad():
            raw, mean, lp, lq = sample_bayesian(bmodel, inputs, S * _world())
            ce = torch.nn.functional.cross_entropy
            nll = 0.5 * (ce(mean[0].float(), sp) + ce(mean[1].float(), ep))  # examples/bert_squad.py:474-481
            return elbo(lp, lq, nll.double(), n_batches)

    def cpu_baseline():
        from oracle.model_oracle import log_probs, to_oracle

        omodel = to_oracle(model, delta=0.05).eval()
        with torch.no_grad():
            t0 = time.perf_counter()
            omodel(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long))
            log_probs(omodel)
            dt = time.perf_counter() - t0
        return {"value": 1 / dt, "unit": "MC-samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"1 MC sample (fwd + log-probs) of the same BERT-large B=16 L=384 batch, torch-CPU fp32, {dt:.1f}s"}

    cfgd = {"workload": "to_bayesian(BERT-large QA, delta=0.05, freeze=True) fwd+ELBO", "samples_per_gpu": S, "batch": B,
            "allreduce_values": 2 * B * L + 2,
            "seq_len": L, "bayesian_linears": len(bmodel.fused_children()), "gelu_fused_into_gemm": n_fused,
            "residual_layernorm_fused": n_ln, "qkv_in_one_launch": n_qkv,
            "attention_kernel": bool(attn), "embeddings_in_one_launch": n_emb}
    return step, cpu_baseline, cfgd, bmodel


def make_linear(device, S, dtype, M):
    import bayeformers_amd as bf
    import bayeformers_amd.nn as bnn
    from bayeformers_amd.sampling import elbo, sample_bayesian

    torch.manual_seed(0)
    layer = bnn.Linear(768, 768)
    model = bnn.Model(layer).to(device)
    x = torch.randn(M, 768)
    xd = x.to(device)
    if dtype != "fp32":
        xd = xd.to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    tgt = torch.randint(0, 768, (M,), device=device)

    def step():
        with torch.no_grad():
            raw, mean, lp, lq = sample_bayesian(model, xd, S * _world())
            nll = torch.nn.functional.cross_entropy(mean[0].float(), tgt)
            return elbo(lp, lq, nll.double(), 100)

    def cpu_baseline():
        from oracle import bayes_oracle as bo

        mu_w, rho_w = layer.weight.mu.detach().cpu(), layer.weight.rho.detach().cpu()
        mu_b, rho_b = layer.bias.mu.detach().cpu(), layer.bias.rho.detach().cpu()
        prior = ("mixture", 0.5, 1.0, float(torch.tensor(-6.0).exp()))
        n = 3 if M <= 64 else 2
        with torch.no_grad():
            bo.cpu_reference_step(x, mu_w, rho_w, mu_b, rho_b, S, prior)
            t0 = time.perf_counter()
            for _ in range(n):
                bo.cpu_reference_step(x, mu_w, rho_w, mu_b, rho_b, S, prior)
            dt = time.perf_counter() - t0
        return {"value": n * S / dt, "unit": "MC-samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} steps of S={S} serial samples, x=[{M},768], torch-CPU fp32, {dt:.1f}s"}

    cfgd = {"workload": f"bnn.Linear(768,768) default init + mixture prior, x=[{M},768], fwd+ELBO",
            "samples_per_gpu": S, "batch": M}
    return step, cpu_baseline, cfgd, model


def make_mlp(device, S, dtype):
    import bayeformers_amd as bf
    from bayeformers_amd.sampling import elbo, sample_bayesian

    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(784, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                              torch.nn.Linear(512, 10), torch.nn.LogSoftmax(dim=1))
    bmodel = bf.to_bayesian(mlp, delta=0.05).to(device)
    x = torch.rand(128, 784)
    xd, labels = x.to(device), torch.randint(0, 10, (128,), device=device)

    def step():
        with torch.no_grad():
            raw, mean, lp, lq = sample_bayesian(bmodel, xd, S * _world())
            nll = torch.nn.functional.nll_loss(mean[0], labels, reduction="sum")
            return elbo(lp, lq, nll.double(), 469)

    def cpu_baseline():
        from oracle.model_oracle import log_probs, to_oracle

        om = to_oracle(mlp, delta=0.05)
        n = 20
        with torch.no_grad():
            om(x)
            t0 = time.perf_counter()
            for _ in range(n * S):
                om(x)
                log_probs(om)
            dt = time.perf_counter() - t0
        return {"value": n * S / dt, "unit": "MC-samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} steps of S={S} serial samples, MLP 784-512-512-10 B=128, torch-CPU fp32, {dt:.1f}s"}

    return step, cpu_baseline, {"workload": "to_bayesian(MLP 784-512-512-10, delta=0.05) fwd+ELBO", "samples_per_gpu": S,
                                "batch": 128}, bmodel


def alg_gemm_bytes(bmodel, cfgd, S, dtype):
    """Algorithmic HBM bytes per 256x256-tile GEMM launch, averaged over the step's launches of that kernel
    (M = rows per sample >= 128): S*(M*K + N*K + M*N)*elem for a single layer, S*(M*K + L*(N*K + M*N))*elem for L
    layers that share x in one launch (query/key/value)."""
    es = 4 if dtype == "fp32" else 2
    M = cfgd.get("batch", 1) * cfgd.get("seq_len", 1)
    if M < 128:
        return None
    stacked = getattr(getattr(bmodel, "_plan", None), "stacked", {}) or {}
    tot = []
    for l in bmodel.fused_children():
        if getattr(l, "_small_m", True):
            continue
        run = getattr(l, "_shared_input", None)
        if run is not None and id(run[0]) in stacked:
            if run[0] is l:
                tot.append(S * (M * l.in_features + len(run) * (l.out_features * l.in_features + M * l.out_features)) * es)
            continue
        tot.append(S * (M * l.in_features + l.out_features * l.in_features + M * l.out_features) * es)
    return round(sum(tot) / len(tot)) if tot else None


def measure_traffic(args):
    """roofline.traffic: HBM bytes per GEMM launch from the PMC counters, collected as MI355X_MICROARCH.md's HBM
    section prescribes — FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only) over a short
    run of this same workload, values in KiB, FETCH_SIZE doubled (gfx950 reports half the bytes of wide 16 B/lane
    reads, which is what the LDS-DMA loads are; calibrated here on the sampling kernel's known 16 B/scalar reads),
    WRITE_SIZE as is for the 16-byte epilogue stores (calibrated on their known byte count).  Returns None if the
    profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="bf_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-traffic", "--graph", "off"]
            if args.samples:
                cmd += ["--samples", str(args.samples)]
            if args.dtype:
                cmd += ["--dtype", args.dtype]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for f in files:
                for r in csv.DictReader(open(f)):
                    if ("gemm256_ring5" in r["Kernel_Name"] or "gemm256_sched" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            out[counter] = sum(vals) / len(vals) * 1024.0
        return {"bytes_per_launch": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"], "fetch_bytes": 2.0 * out["FETCH_SIZE"],
                "write_bytes": out["WRITE_SIZE"]}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def visible_gpus():
    """Number of GPUs the ranks will see, WITHOUT touching the HIP runtime: the KFD topology in sysfs lists every node
    (CPUs have simd_count 0), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  None if sysfs has no KFD
    topology (no amdgpu driver): the ranks then find out themselves."""
    import glob

    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip()]))
    return n


def launch_ranks(args) -> int:
    """--gpus N without a torch.distributed environment: run the N ranks as a CHILD process tree (one rank per GPU,
    torch.distributed.run) and relay its output.  This process never initialises a GPU: it counts them in sysfs."""
    import socket
    import subprocess

    have = visible_gpus()
    if not args.dry_run and have is not None and have < args.gpus and os.environ.get("BF_BENCH_SHARE_GPU") is None:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def dry_run(args, world, rank, device):
    """Process-group check: one all-reduce over all ranks, rank 0 reports how many took part."""
    t = torch.ones(1, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": None, "n_gpus": world, "ranks_counted": int(t.item()),
                          "backend": dist.get_backend() if world > 1 else None, "steps": 0, "warmup": 0}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    global _STRONG
    args = parse()
    _STRONG = bool(args.strong)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never report an N-rank flag over a different number of ranks
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
              f"`python bench.py --gpus {args.gpus}` or torch.distributed.run --nproc-per-node {args.gpus}",
              file=sys.stderr)
        sys.exit(2)
    have_gpu = torch.cuda.is_available()
    if args.dry_run and not have_gpu:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        return dry_run(args, world, rank, torch.device("cpu"))
    # developer switches for exercising the N-rank path on a ONE-GPU box (never set by the driver): all ranks on cuda:0 and
    # gloo collectives on the CUDA tensors (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    share_gpu = os.environ.get("BF_BENCH_SHARE_GPU") is not None
    if share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group(os.environ.get("BF_BENCH_BACKEND", "gloo"))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if args.dry_run:
        return dry_run(args, world, rank, device)

    import bayeformers_amd as bf
    from bayeformers_amd import _C

    defaults = {"bert_base": (10, "bf16"), "bert_base_train": (10, "bf16"), "bert_large_qa": (10, "fp16"), "linear768": (10, "bf16"), "linear768_m32": (10, "bf16"), "mlp": (5, "bf16")}
    S = args.samples or defaults[args.workload][0]
    dtype = args.dtype or defaults[args.workload][1]
    bf.set_compute_dtype(dtype)
    bf.manual_seed(0x5EED)
    if args.workload == "bert_base":
        step, cpu_baseline, cfgd, bmodel = make_bert(device, S, dtype)
    elif args.workload == "bert_base_train":
        step, cpu_baseline, cfgd, bmodel = make_bert(device, S, dtype, train=True, train_mode=args.train_mode)
    elif args.workload == "bert_large_qa":
        step, cpu_baseline, cfgd, bmodel = make_bert_large_qa(device, S, dtype)
    elif args.workload == "linear768":
        step, cpu_baseline, cfgd, bmodel = make_linear(device, S, dtype, 4096)
    elif args.workload == "linear768_m32":
        step, cpu_baseline, cfgd, bmodel = make_linear(device, S, dtype, 32)
    else:
        step, cpu_baseline, cfgd, bmodel = make_mlp(device, S, dtype)

    def barrier():
        if world > 1:
            dist.barrier()

    # Launch-bound workloads (tens of microseconds of GPU work per step) are replayed from a HIP graph: the whole
    # step — sampling, GEMMs, ELBO — is captured once; the Monte-Carlo sample counter lives on the device so every
    # replay draws fresh epsilon (bayeformers_amd.use_device_counter).
    use_graph = args.graph == "on" or (args.graph == "auto" and world == 1 and
                                       args.workload in ("linear768", "linear768_m32", "mlp"))
    if use_graph:
        bf.use_device_counter(True, device=device)
        for _ in range(3):
            step()  # builds plans / workspaces outside the capture
        torch.cuda.synchronize()
        static_out = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = step()
        eager_step = step

        def step():  # noqa: F811
            graph.replay()
            return static_out

    # Every step's ELBO lands in host-visible (pinned) memory through an asynchronous copy on the compute stream;
    # the host only waits once, after the K-th step, so consecutive steps are not serialised on a readback.
    elbo_host = torch.empty(max(args.steps, 1), dtype=torch.float64, pin_memory=True)
    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        elbo_host[i:i + 1].copy_(step().reshape(1), non_blocking=True)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    last = float(elbo_host[args.steps - 1])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    # roofline leg: the same steps again with HIP events around every GEMM / sampling launch
    lib = _C.lib()
    lib.bf_profile_reset()
    lib.bf_profile_enable(1)
    prof_steps = max(1, min(args.steps, 5))
    prof_step = eager_step if use_graph else step
    for _ in range(prof_steps):
        prof_step()
    torch.cuda.synchronize()
    lib.bf_profile_enable(0)
    prof = {}
    for kind, name in ((_C.BF_PROF_GEMM, "gemm"), (_C.BF_PROF_SAMPLE, "sample"), (_C.BF_PROF_FUSED_SMALL, "fused")):
        n, ms, work = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        _C.check(lib.bf_profile_read(kind, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work)), "bf_profile_read")
        prof[name] = (n.value, ms.value, work.value)
    lib.bf_profile_reset()

    # the step's one collective on its own: the packed [sum of outputs | sum log_prior | sum lvp] fp64 buffer
    allreduce_ms = None
    if world > 1:
        n_vals = int(cfgd.get("allreduce_values", 66))
        buf = torch.zeros(n_vals, dtype=torch.float64, device=device)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = e0.elapsed_time(e1) / 50

    if rank == 0:
        gn, gms, gflop = prof["gemm"]
        sn, sms, sbytes = prof["sample"]
        plan = getattr(bmodel, "_plan", None)
        if sbytes == 0 and plan is not None:
            # cross-layer launches: algorithmic bytes = mu,rho (+ Gaussian prior mu,rho) read once, S samples written
            from bayeformers_amd.nn import Gaussian
            per_read = 16 if isinstance(plan.layers[0].weight_prior, Gaussian) else 8
            esz = 4 if dtype == "fp32" else 2
            sbytes = float(plan.scalars) * (per_read + S * esz) * prof_steps
        fn, fms, fflop = prof["fused"]
        fused = {"kernel": "fused_small_kernel (sampling + log-probs + MFMA in one launch, M <= 64)", "bound": "latency",
                 "launches_per_step": fn // prof_steps, "avg_launch_us": round(1e3 * fms / max(fn, 1), 2),
                 "flop_per_step": fflop / prof_steps} if fn else None
        if gn == 0 and fn:
            # small-M workloads: the only matrix kernel of the step is the single fused launch, a latency-bound kernel
            # (a few microseconds of work); the MFMA rate is reported for completeness, it is not what bounds it
            gn, gms, gflop = fn, fms, fflop
            kernel, bound, fused = fused["kernel"], "latency", None
        else:
            kernel = "gemm256_ring5_kernel (sampled-weight GEMM, five-slot LDS ring; mean over the step's tiled-GEMM launches)"
            bound = "mfma"
        tflops = gflop / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
        roofline = {"bound": bound, "kernel": kernel,
                    "achieved": round(tflops, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / PEAK_TFLOPS, 4), "traffic": None,
                    "launches_per_step": gn // prof_steps, "avg_launch_us": round(1e3 * gms / max(gn, 1), 2),
                    "flop_per_step": gflop / prof_steps, "gemm_ms_per_step": round(gms / prof_steps, 3),
                    "fused_small_kernel": fused,
                    "sample_kernel": {"bound": "hbm", "achieved": round(sbytes / (sms * 1e-3) / 1e9, 1) if sms > 0 else 0.0,
                                      "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(sbytes / (sms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if sms > 0 else 0.0,
                                      "launches_per_step": sn // prof_steps, "ms_per_step": round(sms / prof_steps, 3),
                                      "bytes_per_step": sbytes / prof_steps}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        if world == 1 and not args.no_traffic and gn > 0 and not args.workload.endswith("_train"):
            tr = measure_traffic(args)
            if tr is not None:
                roofline["traffic"] = round(tr["bytes_per_launch"])
                roofline["traffic_detail"] = {"unit": "bytes per GEMM launch (mean over the step's launches)",
                                              "hbm_fetch": round(tr["fetch_bytes"]), "hbm_write": round(tr["write_bytes"]),
                                              "algorithmic": alg_gemm_bytes(bmodel, cfgd, S, dtype)}
        n_ranks = dist.get_world_size() if world > 1 else 1
        per_step = S if args.strong else S * n_ranks
        total_samples = per_step * args.steps
        if args.strong:
            from bayeformers_amd.sampling import shard_span

            cfgd["samples_per_gpu"] = [shard_span(S, r, n_ranks)[1] for r in range(n_ranks)]
        cfgd.update({"parallelism": f"mc-sample-shard x{n_ranks}", "last_elbo": last, "hip_graph": bool(use_graph),
                     "samples_total": total_samples, "samples_per_step": per_step,
                     "allreduce_ms_per_step": round(allreduce_ms, 4) if allreduce_ms is not None else None})
        metric = "MC-samples/sec (fwd+ELBO+backward+AdamW)" if args.workload.endswith("_train") else "MC-samples/sec (fwd+ELBO)"
        out = {"metric": metric, "value": round(total_samples / dt, 3), "unit": "MC-samples/s",
               "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": cfgd, "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
