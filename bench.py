#!/usr/bin/env python3
"""bench.py — MC-samples/sec (fwd+ELBO) of the MI355X Monte-Carlo variational forward path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bert_base|bert_base_serial|bert_large_qa|bert_base_train|bert_large_qa_train|linear768|linear768_m32|mlp]

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
PEAK_TFLOPS_F32 = 157.3  # fp32-input MFMA (v_mfma_f32_16x16x4_f32): --dtype fp32, the reference's own precision
PEAK_HBM_GBS = 8000.0  # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bert_base", choices=["bert_base", "bert_base_serial", "bert_large_qa", "linear768", "linear768_m32", "mlp", "bert_base_train", "bert_large_qa_train"])
    ap.add_argument("--samples", type=int, default=None, help="MC samples per GPU per step (default: workload's)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --samples is the TOTAL number of MC samples per step, sharded over the ranks "
                         "(uneven shards allowed: BASELINE.json configs[4] asks for S=10 on 8 GPUs = 2,2,1,1,1,1,1,1); "
                         "the default is weak scaling, --samples per GPU, which is what the driver's scaling run measures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-mode", action="store_true",
                    help="(default since round 4, accepted for old command lines) bert_base_train runs with the model in "
                         ".train() like examples/bert_glue.py:221: HF dropout p = 0.1 active, inside the fused kernels")
    ap.add_argument("--no-dropout", action="store_true",
                    help="bert_base_train only: keep the wrapped model's modules in eval mode (dropout off), the step rounds "
                         "1-3 timed")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--calibrate-traffic", action="store_true",
                    help="(set by the PMC child passes) run the known-size streaming reads that calibrate the counters first")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group check only: every rank joins the group (RCCL on GPUs, gloo without), "
                         "one all-reduce, rank 0 prints a JSON line with n_gpus; no kernels run")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a HIP graph (device-resident sample counter: every replay draws fresh "
                         "epsilon); auto = the single-layer / MLP workloads on one rank (whole step captured) and the BERT "
                         "forward workloads through sampling.GraphedSampler: on one rank always, on several ranks when a "
                         "rank's shard is <= 4 samples; the training workloads through training.GraphedTrainingStep on one rank "
                         "(S-sharded ranks run the eager step: bucketed gradient all-reduce under backward)")
    return ap.parse_args()


_STRONG = False  # --strong: S is the total per step


def _world() -> int:
    """What a workload multiplies its S by to get the samples of a step.  Weak scaling (default): S is the number of
    Monte-Carlo samples PER GPU, a step draws S * world samples and sample_bayesian hands every rank its S of them.
    --strong: S is the total, sample_bayesian shards it (unevenly if it must)."""
    return 1 if _STRONG else _ranks()


def _ranks() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def honoured_env():
    """Every BF_* variable set in this process's environment: they are developer switches that change what is built or
    timed (BF_BENCH_NO_*_FUSION, BF_BENCH_TRAIN_*, BF_BENCH_SHARE_GPU, BF_PLAN_ARENA_BYTES, BF_LIB_PATH, ...), so a line
    produced under any of them says so in its `config`.  The defaults the driver measures show exactly ONE entry: the
    HSA_ENABLE_IPC_MODE_LEGACY setting, which is always echoed with where its value came from."""
    env = {k: os.environ[k] for k in sorted(os.environ) if k.startswith("BF_")}
    # the one runtime setting multi-process GPU work depends on here (dmabuf IPC; RCCL and CUDA-tensor sharing fail without
    # it on this driver): always echoed, with where its value came from
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = ipc_mode_setting()
    return env


_IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"


def ipc_mode_setting():
    v = os.environ.get(_IPC_VAR)
    if v is None:
        return "(unset)"
    return v + (" (bench.py default)" if os.environ.get("BF_BENCH_IPC_DEFAULTED") == "1" else " (environment)")


def default_ipc_mode(env) -> None:
    """HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller's environment says otherwise (any value it holds is kept): the host
    driver of this pool supports dmabuf IPC only — without it RCCL's hipIpcGetMemHandle fails with `invalid argument`.
    Must happen before the process (or its children) initialises the HIP runtime.  Logged on stderr."""
    if _IPC_VAR in env:
        if env.get("BF_BENCH_IPC_DEFAULTED") != "1":
            print(f"bench.py: {_IPC_VAR}={env[_IPC_VAR]} (from the environment)", file=sys.stderr)
        return
    env[_IPC_VAR] = "0"
    env["BF_BENCH_IPC_DEFAULTED"] = "1"
    print(f"bench.py: {_IPC_VAR} was unset, defaulting to 0 (dmabuf IPC; export it to override)", file=sys.stderr)


def timed_cpu(fn, n):
    """cpu_baseline timing (core count stated, mean and min of >= 5 after a warm-up): `fn` is
    first timed once at 8, 16, 32 and 64 threads — the port oversubscribes on a many-core host (measured: 1.5 s per
    BERT-base sample at 32 threads, 256 s at all 256) — then `n` times at the best count.  Returns (threads, [seconds], {threads: seconds})."""
    nproc = os.cpu_count() or 1
    cands = sorted({min(t, nproc) for t in (8, 16, 32, 64)})  # (all 256 threads of a GPU host: minutes per sample)
    prev = torch.get_num_threads()
    sweep = {}
    try:
        torch.set_num_threads(cands[0])
        fn()  # warm-up (allocations, oneDNN primitives)
        for t in cands:
            torch.set_num_threads(t)
            t0 = time.perf_counter()
            fn()
            sweep[t] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        times = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev)
    return best, times, sweep


def cpu_line(units_per_call, times, best, sweep, what):
    mean = sum(times) / len(times)
    return {"value": units_per_call / mean, "unit": "MC-samples/s", "cores": best, "kind": "port",
            "best": units_per_call / min(times), "host_threads_available": os.cpu_count(),
            "thread_sweep_s_per_call": {str(t): round(v, 3) for t, v in sweep.items()},
            "sample": f"{what}: mean of {len(times)} after a warm-up, torch-CPU fp32 at {best} threads "
                      f"(best of the sweep), {sum(times):.1f}s"}


class _Harness:
    """The S-sample forward of the BERT workloads: `sample_bayesian` called eagerly, or — graphed(True) — the same step
    replayed from a HIP graph by the library's GraphedSampler (the rank's forward and sums in the graph, the S-shard
    group's collective eagerly after it).  graphed(False) goes back to eager calls (the roofline leg times single launches)."""

    def __init__(self, bmodel, inputs, samples):
        self.bmodel, self.inputs, self.samples, self.sampler, self.on = bmodel, inputs, samples, None, False

    def graphed(self, on=True):
        from bayeformers_amd.sampling import GraphedSampler

        if on and self.sampler is None:
            self.sampler = GraphedSampler(self.bmodel, self.inputs, self.samples)
        self.on = bool(on)

    def __call__(self):
        from bayeformers_amd.sampling import sample_bayesian

        if self.on:
            return self.sampler()
        return sample_bayesian(self.bmodel, self.inputs, self.samples)


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
    n_ffn = bf.fuse_ffn_pairs(bmodel) if os.environ.get("BF_BENCH_NO_FFN_PAIR") is None else 0  # (acts when gradients are recorded)
    if dtype != "fp32":
        bmodel = bmodel.to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    g = torch.Generator().manual_seed(321)
    ids = torch.randint(0, cfg.vocab_size, (B, L), generator=g)
    labels = torch.randint(0, 2, (B,), generator=g)
    inputs = {"input_ids": ids.to(device), "attention_mask": torch.ones(B, L, dtype=torch.long, device=device)}
    info = {"gelu_fused_into_gemm": n_fused, "residual_layernorm_fused": n_ln, "qkv_in_one_launch": n_qkv,
            "attention_kernel": bool(attn), "embeddings_in_one_launch": n_emb, "ffn_pair_one_autograd_node": n_ffn}
    return bmodel, model, inputs, ids, labels, info


def _graphed_train_step(bmodel, inputs, samples, nll_fn, opt, n_batches, max_grad_norm, eager):
    """The training workloads' step through training.GraphedTrainingStep (one process): two eager steps, then the step replayed
    from a HIP graph.  If the capture fails the eager step keeps running and the line says so (`step.note`)."""
    from bayeformers_amd.training import GraphedTrainingStep

    state = {"g": GraphedTrainingStep(bmodel, inputs, samples, nll_fn, opt, n_batches, max_grad_norm=max_grad_norm), "note": None}

    def step():
        g = state["g"]
        if g is None:
            return eager()
        try:
            return g()
        except Exception as e:  # noqa: BLE001 - whatever the capture raised, the eager step is still the product path
            state["g"], state["note"] = None, f"capture failed, step ran eagerly ({type(e).__name__}: {str(e)[:160]})"
            print(f"bench.py: {state['note']}", file=sys.stderr)
            g.close()
            return eager()

    step.state, step.eager = state, eager
    return step


def make_bert(device, S, dtype, train=False, train_mode=False, serial=False, graph_train=False):
    from bayeformers_amd.sampling import elbo, sample_bayesian

    B, L, n_batches = 32, 128, 2105  # SST-2: 67,349 train sentences / 32
    bmodel, model, inputs, ids, labels, info = build_bert(device, dtype)
    labels_d = labels.to(device)

    harness = _Harness(bmodel, inputs, S * _world())

    def step():
        with torch.no_grad():
            raw, mean, lp, lq = harness()
            nll = torch.nn.functional.cross_entropy(mean[0].float(), labels_d)
            return elbo(lp, lq, nll.double(), n_batches)

    step.harness = harness
    if serial:
        # The reference's own caller loop, unchanged (/root/reference/examples/bert_glue.py:56-73): S serial forwards of ONE
        # sample each, the two log-prob sums read after every forward, then the means and the ELBO — what a user script that
        # switches packages without adopting sample_bayesian() runs.  bnn.Model.__call__ replays such evaluation forwards
        # from a HIP graph from the third call of a signature on (bayeformers_amd/graphs.py; BF_NO_AUTO_GRAPH=1: eager).
        def step():  # noqa: F811
            with torch.no_grad():
                logits, lps, lqs = [], [], []
                for _ in range(S):
                    logits.append(bmodel(**inputs).logits)
                    lps.append(bmodel.log_prior())
                    lqs.append(bmodel.log_variational_posterior())
                mean = torch.stack(logits).float().mean(0)
                nll = torch.nn.functional.cross_entropy(mean, labels_d)
                return elbo(torch.stack(lps).double().mean(), torch.stack(lqs).double().mean(), nll.double(), n_batches)

    if train:
        # the reference's training step (examples/bert_glue.py:227-241) — forward, ELBO, backward through
        # every sampled-weight layer (eps regenerated from the Philox counter), Adam on the unfrozen parameters
        # (examples/bert_glue.py:215 uses transformers' AdamW(lr, eps) with its default weight_decay = 0: torch's AdamW
        # with weight_decay = 0 is the same update; fused = one multi-tensor kernel for all 85 parameter tensors)
        from bayeformers_amd.training import GradientBuckets, training_step

        if train_mode:
            bmodel.train()
        params = [p for p in bmodel.parameters() if p.requires_grad]
        graph_train = graph_train and _ranks() == 1 and os.environ.get("BF_BENCH_TRAIN_BUCKETS") is None
        opt = torch.optim.AdamW(params, lr=2e-5, eps=1e-8, weight_decay=0.0, fused=True, capturable=graph_train)
        # world > 1: flat gradient buffers, all-reduced over the ranks while backward runs
        buckets = GradientBuckets(params) if _ranks() > 1 or os.environ.get("BF_BENCH_TRAIN_BUCKETS") is not None else None

        def nll_fn(mean):
            return torch.nn.functional.cross_entropy(mean[0].float(), labels_d)

        def step():  # noqa: F811
            # examples/bert_glue.py:227-241: forward, ELBO, backward, clip_grad_norm_(1), optimizer step
            return training_step(bmodel, inputs, S * _world(), nll_fn, opt, n_batches, buckets=buckets,
                                 max_grad_norm=None if os.environ.get("BF_BENCH_TRAIN_NO_CLIP") is not None else 1.0)

        if graph_train:
            step = _graphed_train_step(bmodel, inputs, S, nll_fn, opt, n_batches,
                                       None if os.environ.get("BF_BENCH_TRAIN_NO_CLIP") is not None else 1.0, step)

    def cpu_baseline():
        from oracle.model_oracle import log_probs, to_oracle

        if train:
            return None
        omodel = to_oracle(model, delta=0.05).eval()
        mask = torch.ones(B, L, dtype=torch.long)

        def one_sample():
            with torch.no_grad():
                omodel(input_ids=ids, attention_mask=mask)
                log_probs(omodel)

        best, times, sweep = timed_cpu(one_sample, 5)
        return cpu_line(1, times, best, sweep, "serial MC samples (fwd + log-probs) of the same BERT-base B=32 L=128 batch")

    cfgd = {"workload": "to_bayesian(BERT-base seq-cls, delta=0.05, freeze=True) " +
                        (("training step: fwd+ELBO+backward+clip+AdamW, " +
                          ("model.train(): HF dropout 0.1 inside the fused kernels" if train_mode else "dropout off (--no-dropout)"))
                         if train else ("fwd+ELBO as S SERIAL single-sample forwards (the reference's caller loop, examples/bert_glue.py:63-66)"
                                        if serial else "fwd+ELBO")), "samples_per_gpu": S,
            "batch": B, "seq_len": L, "bayesian_linears": len(bmodel.fused_children()), "bayesian_scalars": 85609730}
    cfgd.update(info)
    return step, cpu_baseline, cfgd, bmodel


def make_bert_large_qa(device, S, dtype, train=False, train_mode=False, graph_train=False):
    """BASELINE config 5: to_bayesian(BERT-large QA) SQuAD-shaped forward + ELBO, seq=384, batch=16.  train: the training
    step of the reference's SQuAD loop (examples/bert_squad.py:456-491: forward of S samples, ELBO with the mean of the
    start / end cross-entropies, backward, clip_grad_norm_(1), AdamW) on the same batch."""
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
    n_ffn = bf.fuse_ffn_pairs(bmodel) if os.environ.get("BF_BENCH_NO_FFN_PAIR") is None else 0
    if dtype != "fp32":
        bmodel = bmodel.to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    g = torch.Generator().manual_seed(654)
    ids = torch.randint(0, cfg.vocab_size, (B, L), generator=g)
    sp, ep = torch.randint(0, L, (B,), generator=g).to(device), torch.randint(0, L, (B,), generator=g).to(device)
    inputs = {"input_ids": ids.to(device), "attention_mask": torch.ones(B, L, dtype=torch.long, device=device)}

    harness = _Harness(bmodel, inputs, S * _world())

    def step():
        with torch.no_grad():
            raw, mean, lp, lq = harness()
            ce = torch.nn.functional.cross_entropy
            nll = 0.5 * (ce(mean[0].float(), sp) + ce(mean[1].float(), ep))  # examples/bert_squad.py:474-481
            return elbo(lp, lq, nll.double(), n_batches)

    step.harness = harness
    if train:
        from bayeformers_amd.training import GradientBuckets, training_step

        if train_mode:
            bmodel.train()
        params = [p for p in bmodel.parameters() if p.requires_grad]
        graph_train = graph_train and _ranks() == 1
        opt = torch.optim.AdamW(params, lr=3e-5, eps=1e-8, weight_decay=0.0, fused=True, capturable=graph_train)
        buckets = GradientBuckets(params) if _ranks() > 1 else None

        def nll_fn(mean):
            ce = torch.nn.functional.cross_entropy
            return 0.5 * (ce(mean[0].float(), sp) + ce(mean[1].float(), ep))

        def step():  # noqa: F811
            return training_step(bmodel, inputs, S * _world(), nll_fn, opt, n_batches, buckets=buckets, max_grad_norm=1.0)

        if graph_train:
            step = _graphed_train_step(bmodel, inputs, S, nll_fn, opt, n_batches, 1.0, step)

    def cpu_baseline():
        from oracle.model_oracle import log_probs, to_oracle

        if train:
            return None
        omodel = to_oracle(model, delta=0.05).eval()
        mask = torch.ones(B, L, dtype=torch.long)

        def one_sample():
            with torch.no_grad():
                omodel(input_ids=ids, attention_mask=mask)
                log_probs(omodel)

        best, times, sweep = timed_cpu(one_sample, 2)
        return cpu_line(1, times, best, sweep, "serial MC samples (fwd + log-probs) of the same BERT-large B=16 L=384 batch")

    what = "fwd+ELBO"
    if train:
        what = ("training step: fwd+ELBO+backward+clip+AdamW, " +
                ("model.train(): HF dropout 0.1 inside the fused kernels" if train_mode else "dropout off (--no-dropout)"))
    cfgd = {"workload": f"to_bayesian(BERT-large QA, delta=0.05, freeze=True) {what}", "samples_per_gpu": S, "batch": B,
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
        def one_step():
            with torch.no_grad():
                bo.cpu_reference_step(x, mu_w, rho_w, mu_b, rho_b, S, prior)

        best, times, sweep = timed_cpu(one_step, 5)
        return cpu_line(S, times, best, sweep, f"steps of S={S} serial samples, x=[{M},768]")

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

        def steps20():
            with torch.no_grad():
                for _ in range(20 * S):
                    om(x)
                    log_probs(om)

        best, times, sweep = timed_cpu(steps20, 5)
        return cpu_line(20 * S, times, best, sweep, f"blocks of 20 steps of S={S} serial samples, MLP 784-512-512-10 B=128")

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


GEMM_POSITIONS = ("qkv", "attn_out", "ffn_up_gelu", "ffn_down")  # order of the tiled launches inside a BERT layer


def calibration_probes(device, shapes=None, S=10, M=4096, dtype=torch.bfloat16):
    """Inside a rocprofv3 --pmc child pass (--calibrate-traffic), ahead of the workload: launches whose byte count and
    data home are KNOWN, so that the same pass yields the counters' calibration.
      * bf_probe_read_kernel over a 2 GiB buffer (eight times the 256 MiB Infinity Cache: every line from HBM), then
        three passes over a 96 MiB buffer (inside the cache, three times the aggregate L2): checks the FETCH_SIZE unit;
      * for each tiled-GEMM shape of the step (`shapes`: [(layers, N, K, act)]), the launch itself twice: once after the
        2 GiB read has pushed everything out of the Infinity Cache (operands from HBM), once more straight after (operands
        as cache-resident as that launch can have them).  The step's own launches of the same shape are then placed between
        these two by their mean fabric read latency (measure_traffic)."""
    from bayeformers_amd import _C, ops

    lib = _C.lib()
    big = torch.full((2 << 30,), 1, dtype=torch.uint8, device=device)
    small = torch.full((96 << 20,), 1, dtype=torch.uint8, device=device)
    sink = torch.zeros(1, dtype=torch.int32, device=device)
    st = torch.cuda.current_stream().cuda_stream

    def read(buf):
        _C.check(lib.bf_probe_stream_read(buf.data_ptr(), buf.numel(), sink.data_ptr(), st), "bf_probe_stream_read")

    torch.cuda.synchronize()
    read(big)
    for _ in range(3):
        read(small)
    for L, N, K, act in shapes or ():
        g = torch.Generator(device=device).manual_seed(N * 7 + K)
        x = torch.randn(S, M, K, device=device, generator=g).to(dtype)
        w = (torch.randn(L, S, N, K, device=device, generator=g) * 0.05).to(dtype)
        bias = torch.randn(L, S, N, device=device, generator=g)

        def launch():
            if L > 1:
                return ops.gemm_nt_layers(x, w, bias, L, S, M, N, K, M * K, dtype, act)
            return ops.gemm_nt(x, w[0], bias[0], S, M, N, K, M * K, dtype, act)

        y = launch()  # builds the tile schedule of the shape (a device allocation) outside the measured pair
        del y
        read(big)     # flush: nothing of x, w or y is left on chip
        y = launch()  # cold
        y2 = launch()  # hot
        del y, y2
    torch.cuda.synchronize()
    del big, small
    torch.cuda.empty_cache()


PROBE_BYTES = (2 << 30, 96 << 20, 96 << 20, 96 << 20)  # calibration_probes' first four bf_probe_read_kernel dispatches


def bert_gemm_shapes(workload):
    """[(layers in the launch, N, K, activation)] of the step's tiled launches in layer order, M, dtype."""
    if workload == "bert_base":
        return [(3, 768, 768, 0), (1, 768, 768, 0), (1, 3072, 768, 1), (1, 768, 3072, 0)], 32 * 128, torch.bfloat16
    if workload == "bert_large_qa":
        return [(3, 1024, 1024, 0), (1, 1024, 1024, 0), (1, 4096, 1024, 1), (1, 1024, 4096, 0)], 16 * 384, torch.float16
    return None, None, None


def measure_traffic(args):
    """roofline.traffic: HBM-side bytes per GEMM launch from the PMC counters, collected as MI355X_MICROARCH.md's HBM
    section prescribes — FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only) over a short
    run of this same workload, values in KiB, FETCH_SIZE doubled (gfx950 reports half the bytes of wide 16 B/lane
    reads, which is what the LDS-DMA loads are; re-measured in the same pass on bf_probe_read_kernel's known byte
    counts and reported as `fetch_unit_check`), WRITE_SIZE as is for the 16-byte epilogue stores.
    FETCH_SIZE counts the L2's fabric-side reads: Infinity-Cache hits and HBM reads alike, and rocprofv3 lists no
    memory-side (Infinity Cache / UMC) counter on gfx950 (profiles/r4a_rocprofv3_counter_blocks.txt).  A third pass
    (TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_LEVEL_sum) therefore takes the mean time a read request of the L2 spends on the
    fabric (LEVEL / REQ, L2 clocks) for every GEMM launch and places the step's launches, shape by shape, between the SAME
    launch run with all operands in HBM (after a 2 GiB flush) and run again straight after (operands as cache-resident as
    they can be): hit fraction ~ (t_cold - t_step) / (t_cold - t_hot), an ESTIMATE.  Returns None if the profiler is
    unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    shapes, _, _ = bert_gemm_shapes(args.workload)
    n_cal = 3 * len(shapes) if shapes else 0  # per shape: schedule-building launch, cold, hot
    gemm, probe = {}, {}
    tmp = tempfile.mkdtemp(prefix="bf_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    is_gemm = lambda name: "gemm256_ring5" in name or "gemm256_sched" in name
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum")):
            d = os.path.join(tmp, counters[0])
            cmd = [rocprof, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-traffic", "--graph", "off", "--calibrate-traffic"]
            if args.samples:
                cmd += ["--samples", str(args.samples)]
            if args.dtype:
                cmd += ["--dtype", args.dtype]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            try:
                subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            except Exception:
                if counters[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    raise
                continue  # the latency pass is optional
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            gv = {c: [] for c in counters}
            pv = {c: [] for c in counters}
            for f in files:
                for r in csv.DictReader(open(f)):
                    c = r["Counter_Name"]
                    if c not in gv:
                        continue
                    row = (int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"]))
                    if is_gemm(r["Kernel_Name"]):
                        gv[c].append(row)
                    elif "bf_probe_read_kernel" in r["Kernel_Name"]:
                        pv[c].append(row)
            for c in counters:
                gemm[c] = [v for _, v in sorted(gv[c])]   # in dispatch order: calibration launches first, then the steps
                probe[c] = [v for _, v in sorted(pv[c])]
        step = lambda c: gemm.get(c, [])[n_cal:]
        mean = lambda v: sum(v) / len(v)
        if not step("FETCH_SIZE") or not step("WRITE_SIZE"):
            return None
        fetch, write = 2.0 * 1024.0 * mean(step("FETCH_SIZE")), 1024.0 * mean(step("WRITE_SIZE"))
        res = {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write}
        pf = probe.get("FETCH_SIZE") or []
        if len(pf) >= len(PROBE_BYTES) and pf[0] > 0:
            # bytes of a known streaming read per byte FETCH_SIZE reports (the guide's gfx950 correction says 2)
            res["fetch_unit_check"] = round(PROBE_BYTES[0] / (pf[0] * 1024.0), 3)
        req, lvl = gemm.get("TCC_EA0_RDREQ_sum", []), gemm.get("TCC_EA0_RDREQ_LEVEL_sum", [])
        if shapes and len(req) > n_cal and len(req) == len(lvl) and len(gemm["FETCH_SIZE"]) == len(req):
            P = len(shapes)
            by_pos, hbm = {}, 0.0
            for pos in range(P):
                cold = lvl[3 * pos + 1] / req[3 * pos + 1]
                hot = lvl[3 * pos + 2] / req[3 * pos + 2]
                idx = [i for i in range(n_cal, len(req)) if (i - n_cal) % P == pos]
                lat = sum(lvl[i] for i in idx) / sum(req[i] for i in idx)
                fbytes = 2.0 * 1024.0 * mean([gemm["FETCH_SIZE"][i] for i in idx])
                f = min(1.0, max(0.0, (cold - lat) / (cold - hot))) if cold > hot else None
                by_pos[GEMM_POSITIONS[pos]] = {"fabric_fetch": round(fbytes), "ea_read_latency_clk": {
                    "operands_in_hbm": round(cold, 1), "operands_cache_resident": round(hot, 1), "in_step": round(lat, 1)},
                    "infinity_cache_hit_fraction": None if f is None else round(f, 3)}
                hbm += fbytes * (1.0 - (f if f is not None else 0.0))
            res["by_position"] = by_pos
            res["hbm_read_bytes_est"] = hbm / P
            res["infinity_cache_hit_fraction_est"] = round(1.0 - res["hbm_read_bytes_est"] / fetch, 3) if fetch > 0 else None
        return res
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
    default_ipc_mode(env)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def preflight(args, world, rank, device, samples=None):
    """Process-group preflight on the group the run has just built (RCCL through init_process_group("nccl", device_id=...)
    on GPUs): the real run's two kinds of message once — the packed fp64 ELBO buffer (66 values, latency-bound) and one
    128 MiB gradient bucket (bandwidth-bound, asynchronous like training.GradientBuckets sends it) — and who took part, with
    the Monte-Carlo samples each rank will run.  Returns the report; report["ok"] is False when an all-reduce of ones did
    not count exactly `world` ranks."""
    from bayeformers_amd.sampling import shard_span

    line = {"metric": "dry-run", "value": None, "n_gpus": world, "steps": 0, "warmup": 0,
            "backend": dist.get_backend() if world > 1 else None}
    packed = torch.ones(66, dtype=torch.float64, device=device)
    bucket = torch.ones((128 << 20) // 4, dtype=torch.float32, device=device)
    times = {}
    if samples is not None:
        start, count = shard_span(samples, rank, world) if args.strong else (rank * samples, samples)
    else:
        start = count = None
    if world > 1:
        for name, t in (("packed_fp64_66", packed), ("bucket_128MiB", bucket)):
            dist.all_reduce(t)  # first call: communicator / ring set-up
            if device.type == "cuda":
                torch.cuda.synchronize()
            t.fill_(1)
            t0 = time.perf_counter()
            work = dist.all_reduce(t, async_op=True)
            work.wait()
            if device.type == "cuda":
                torch.cuda.synchronize()
            times[name] = round(1e3 * (time.perf_counter() - t0), 3)
        names = [None] * world
        me = {"rank": rank, "device": str(device),
              "gpu": torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu",
              "ipc_mode_legacy": ipc_mode_setting(), "first_sample": start, "samples": count}
        dist.all_gather_object(names, me)
        line["ranks"] = names
    line["scaling"] = "strong" if args.strong else "weak"
    line["samples_per_step"] = None if samples is None else (samples if args.strong else samples * world)
    line["ranks_counted"] = int(packed[0].item())
    line["bucket_ranks_counted"] = int(bucket[-1].item())
    line["allreduce_ms"] = times or None
    if device.type == "cuda":
        try:
            line["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # pragma: no cover
            line["rccl_version"] = None
    line["ok"] = line["ranks_counted"] == world and line["bucket_ranks_counted"] == world
    del packed, bucket
    return line


def dry_run(args, world, rank, device, samples):
    """--dry-run: the preflight alone, so an environment problem shows up in seconds, before any model is built."""
    line = preflight(args, world, rank, device, samples)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not line["ok"]:
        sys.exit(3)


def _train_graph_note(step):
    st = getattr(step, "state", None)
    if st is None:
        return False
    return st["note"] or ("GraphedTrainingStep" if st["g"] is not None and st["g"].graph is not None else False)


DEFAULTS = {"bert_base": (10, "bf16"), "bert_base_serial": (10, "bf16"), "bert_base_train": (10, "bf16"), "bert_large_qa": (10, "fp16"),
            "bert_large_qa_train": (10, "bf16"), "linear768": (10, "bf16"), "linear768_m32": (10, "bf16"), "mlp": (5, "bf16")}


def main():
    global _STRONG
    args = parse()
    _STRONG = bool(args.strong)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    default_ipc_mode(os.environ)  # (ranks started by torch.distributed.run directly: before anything touches the GPU)
    S = args.samples or DEFAULTS[args.workload][0]
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
        return dry_run(args, world, rank, torch.device("cpu"), S)
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
        return dry_run(args, world, rank, device, S)
    pre = None
    if world > 1:
        # the same preflight inside the real run, before any model is built: a group that does not count N ranks ends the
        # run with a JSON error line instead of a number measured on fewer GPUs than claimed
        pre = preflight(args, world, rank, device, S)
        if not pre["ok"]:
            if rank == 0:
                print(json.dumps({"error": "process-group preflight failed: an all-reduce of ones did not count "
                                           f"{world} ranks", "n_gpus": args.gpus, "preflight": pre}), flush=True)
            dist.destroy_process_group()
            sys.exit(3)

    import bayeformers_amd as bf
    from bayeformers_amd import _C

    if args.calibrate_traffic:
        cal_shapes, cal_m, cal_dt = bert_gemm_shapes(args.workload)
        calibration_probes(device, cal_shapes, S=args.samples or 10, M=cal_m or 4096, dtype=cal_dt or torch.bfloat16)
    dtype = args.dtype or DEFAULTS[args.workload][1]
    bf.set_compute_dtype(dtype)
    bf.manual_seed(0x5EED)
    if args.workload == "bert_base":
        step, cpu_baseline, cfgd, bmodel = make_bert(device, S, dtype)
    elif args.workload == "bert_base_serial":
        if world > 1:
            print("bench.py: bert_base_serial is the single-process caller loop (--gpus 1)", file=sys.stderr)
            sys.exit(2)
        step, cpu_baseline, cfgd, bmodel = make_bert(device, S, dtype, serial=True)
        delattr(step, "harness") if hasattr(step, "harness") else None
    elif args.workload == "bert_base_train":
        step, cpu_baseline, cfgd, bmodel = make_bert(device, S, dtype, train=True, train_mode=not args.no_dropout,
                                                     graph_train=args.graph != "off")
    elif args.workload == "bert_large_qa":
        step, cpu_baseline, cfgd, bmodel = make_bert_large_qa(device, S, dtype)
    elif args.workload == "bert_large_qa_train":
        step, cpu_baseline, cfgd, bmodel = make_bert_large_qa(device, S, dtype, train=True, train_mode=not args.no_dropout,
                                                              graph_train=args.graph != "off")
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
    use_graph = (args.graph == "on" and not args.workload.endswith("_train")) or (
        args.graph == "auto" and world == 1 and args.workload in ("linear768", "linear768_m32", "mlp"))
    # The BERT forward workloads go through the library's GraphedSampler instead (the rank's forward + sums in the graph,
    # the S-shard group's collective eagerly after the replay).  A host that takes 4-7 ms to enqueue a forward hides behind
    # the 8.5 ms of kernels of ten samples but not behind the 2-3 ms of a one-to-three-sample shard: auto = on one rank
    # always (the step time no longer depends on how fast this box's host cores are), on several ranks when the shards
    # are that small (--strong); measured on one box, BERT-base, eager vs replay: S = 1 4.99 vs 1.99 ms, 2: 4.40 vs 2.61,
    # 3: 4.15-5.2 vs 3.06, 5: 4.45 vs 4.44, 10: 8.40-8.55 vs 8.43.  If the capture fails the step runs eagerly and says so.
    harness = getattr(step, "harness", None)
    graph_note = None
    if harness is not None:
        per_rank = -(-S // world) if _STRONG else S
        if args.graph == "on" or (args.graph == "auto" and (world == 1 or per_rank <= 4)):
            try:
                harness.graphed(True)
                graph_note = "GraphedSampler"
            except Exception as e:  # noqa: BLE001 - whatever the capture raised, the eager step is still the product path
                harness.on = False
                graph_note = f"capture failed, step ran eagerly ({type(e).__name__}: {str(e)[:160]})"
                print(f"bench.py: {graph_note}", file=sys.stderr)
        use_graph = False
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
    local_dt = dt
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
    prof_step = eager_step if use_graph else getattr(step, "eager", step)  # (a graphed training step: its eager form)
    if harness is not None:
        harness.graphed(False)  # the profiling hooks time single launches: eager calls
    auto_replays = None
    if hasattr(bmodel, "graph_replay"):
        cache = bmodel.__dict__.get("_graphs")
        auto_replays = bool(cache is not None and cache.forwards)  # did bnn.Model.__call__ replay the timed forwards?
        bmodel.graph_replay = False
    for _ in range(prof_steps):
        prof_step()
    torch.cuda.synchronize()
    lib.bf_profile_enable(0)
    prof = {}
    for kind, name in ((_C.BF_PROF_GEMM, "gemm"), (_C.BF_PROF_SAMPLE, "sample"), (_C.BF_PROF_FUSED_SMALL, "fused")):
        n, ms, work = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        _C.check(lib.bf_profile_read(kind, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work)), "bf_profile_read")
        prof[name] = (n.value, ms.value, work.value)
    # the tiled-GEMM launches one by one, in launch order: a BERT step's launch i is position i % 4 of encoder layer i // 4
    n_g = prof["gemm"][0]
    g_ms, g_work = (ctypes.c_float * max(n_g, 1))(), (ctypes.c_double * max(n_g, 1))()
    lib.bf_profile_read_launches(_C.BF_PROF_GEMM, g_ms, g_work, n_g)
    gemm_launches = [(float(g_ms[i]), float(g_work[i])) for i in range(n_g)]
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

    # every rank's GEMM / sampling time per step (a straggler shows in the N-rank line)
    by_rank = None
    if world > 1:
        mine = torch.tensor([prof["gemm"][1] / prof_steps, prof["sample"][1] / prof_steps, 1e3 * local_dt / args.steps],
                            dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        by_rank = {"gemm_ms_per_step": [round(float(t[0]), 3) for t in allr],
                   "sample_ms_per_step": [round(float(t[1]), 3) for t in allr],
                   "ms_per_step": [round(float(t[2]), 3) for t in allr]}

    if rank == 0:
        peak = PEAK_TFLOPS_F32 if dtype == "fp32" else PEAK_TFLOPS
        gn, gms, gflop = prof["gemm"]
        sn, sms, sbytes = prof["sample"]
        aliased = False
        plan = getattr(bmodel, "_plan", None)
        if plan is not None:
            # cross-layer launches (they report no byte count themselves): algorithmic bytes = mu,rho (+ Gaussian prior mu,rho)
            # read once, S samples written — ADDED to what the per-layer launches of the step reported (fp32: the pooler and
            # the classifier sample through bf_sample_logprob, whose few MB used to stand for the whole 4 GB of the plan)
            from bayeformers_amd import ops as bf_ops
            from bayeformers_amd.nn import Gaussian
            l0 = plan.layers[0]
            # a Gaussian prior is read too (16 B per scalar) unless it is the alias of the frozen posterior mean (MOPED)
            per_read = 16 if isinstance(l0.weight_prior, Gaussian) and bf_ops.prior_alias(l0.weight, l0.weight_prior) is None else 8
            aliased = isinstance(l0.weight_prior, Gaussian) and per_read == 8
            esz = 4 if dtype == "fp32" else 2
            sbytes += float(plan.scalars) * (per_read + plan.S * esz) * prof_steps
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
            if dtype == "fp32" and os.environ.get("BF_F32_GENERIC") is not None:
                # (bf_launch_gemm_nt dispatches the generic kernel under this switch — and for shapes the ring form refuses:
                # K % 32, N % 4, unaligned operands, M * N < 128 * 128; none of the workloads here has such a layer)
                kernel = "gemm_nt_f32_kernel (generic fp32 kernel: BF_F32_GENERIC is set)"
            elif dtype == "fp32":
                kernel = ("gemm256_ring5_kernel<float> (sampled-weight GEMM on v_mfma_f32_16x16x4_f32, five-slot LDS ring; mean "
                          "over the step's tiled-GEMM launches)")
            else:
                kernel = "gemm256_ring5_kernel (sampled-weight GEMM, five-slot LDS ring; mean over the step's tiled-GEMM launches)"
            bound = "mfma"
        tflops = gflop / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
        # by position (BERT forward workloads: 4 tiled launches per encoder layer, in the model's order)
        by_position = None
        per_step = len(gemm_launches) // prof_steps if prof_steps else 0
        n_enc = 4 * getattr(getattr(getattr(bmodel, "model", None), "config", None), "num_hidden_layers", 0)  # encoder launches per step
        if args.workload in ("bert_base", "bert_large_qa") and bound == "mfma" and n_enc and per_step >= n_enc \
                and len(gemm_launches) == per_step * prof_steps and all(t >= 0 for t, _ in gemm_launches):
            # (fp32: the pooler and the classifier run the tiled kernel too, after the encoder's launches — they are part of
            # roofline.frac and of no position)
            by_position = {}
            enc_ms = sum(t for i, (t, _) in enumerate(gemm_launches) if i % per_step < n_enc)
            for pos, name in enumerate(GEMM_POSITIONS):
                sel = [gemm_launches[i] for i in range(len(gemm_launches)) if i % per_step < n_enc and (i % per_step) % 4 == pos]
                t_ms, fl = sum(t for t, _ in sel), sum(w for _, w in sel)
                by_position[name] = {"frac": round(fl / (t_ms * 1e-3) / 1e12 / peak, 4), "achieved": round(fl / (t_ms * 1e-3) / 1e12, 1),
                                     "avg_launch_us": round(1e3 * t_ms / len(sel), 2), "share_of_gemm_time": round(t_ms / enc_ms, 4)}
        # the north star asks for ONE fused reparameterise + GEMM kernel; here the weights are sampled by their own launch
        # (LABBOOK.md section 4.3), so the fraction a fused kernel would be held to is flop / (GEMM time + sampling time)
        tflops_ws = gflop / ((gms + sms) * 1e-3) / 1e12 if gms + sms > 0 else 0.0
        roofline = {"bound": bound, "kernel": kernel,
                    "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(tflops / peak, 4), "frac_with_sampling": round(tflops_ws / peak, 4), "traffic": None,
                    "launches_per_step": gn // prof_steps, "avg_launch_us": round(1e3 * gms / max(gn, 1), 2),
                    "flop_per_step": gflop / prof_steps, "gemm_ms_per_step": round(gms / prof_steps, 3),
                    "by_position": by_position,
                    "fused_small_kernel": fused,
                    "sample_kernel": {"bound": "hbm", "achieved": round(sbytes / (sms * 1e-3) / 1e9, 1) if sms > 0 else 0.0,
                                      "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(sbytes / (sms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if sms > 0 else 0.0,
                                      "launches_per_step": sn // prof_steps, "ms_per_step": round(sms / prof_steps, 3),
                                      "bytes_per_step": sbytes / prof_steps, "moped_prior_aliased": aliased}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        if world == 1 and not args.no_traffic and gn > 0 and not args.workload.endswith("_train") and args.workload != "bert_base_serial":
            tr = measure_traffic(args)
            if tr is not None:
                roofline["traffic"] = round(tr["bytes_per_launch"])
                detail = {"unit": "bytes per GEMM launch (mean over the step's launches)",
                          "fabric_fetch": round(tr["fetch_bytes"]), "hbm_write": round(tr["write_bytes"]),
                          "algorithmic": alg_gemm_bytes(bmodel, cfgd, S, dtype)}
                if "hbm_read_bytes_est" in tr:
                    # FETCH_SIZE counts the L2's fabric-side reads, Infinity-Cache hits included; split by the mean fabric
                    # read latency of the launches between an HBM stream and an Infinity-Cache stream (measure_traffic)
                    detail.update({"hbm_read": round(tr["hbm_read_bytes_est"]),
                                   "infinity_cache_hit_fraction": tr["infinity_cache_hit_fraction_est"],
                                   "by_position": tr["by_position"],
                                   "hbm_read_method": "estimate: each launch's mean fabric read latency (TCC_EA0_RDREQ_LEVEL / "
                                                      "TCC_EA0_RDREQ) placed between the same launch with its operands in HBM "
                                                      "(after a 2 GiB flush) and repeated at once (cache-resident), same PMC "
                                                      "pass; rocprofv3 exposes no memory-side counter on gfx950"})
                if "fetch_unit_check" in tr:
                    detail["fetch_unit_check"] = tr["fetch_unit_check"]
                roofline["traffic_detail"] = detail
        n_ranks = dist.get_world_size() if world > 1 else 1
        per_step = S if args.strong else S * n_ranks
        total_samples = per_step * args.steps
        if args.strong:
            from bayeformers_amd.sampling import shard_span

            cfgd["samples_per_gpu"] = [shard_span(S, r, n_ranks)[1] for r in range(n_ranks)]
        if by_rank is not None:
            roofline["by_rank"] = by_rank
        if pre is not None:  # the N-rank run's own preflight: who took part, what the two message kinds cost
            cfgd["preflight"] = {k: pre[k] for k in ("backend", "ranks", "ranks_counted", "bucket_ranks_counted",
                                                     "allreduce_ms", "rccl_version") if k in pre}
        cfgd.update({"env": honoured_env(), "parallelism": f"mc-sample-shard x{n_ranks}", "last_elbo": last, "hip_graph": graph_note if harness is not None and graph_note else (
                         bool(use_graph) or ("bnn.Model.__call__ replay" if auto_replays else False) or _train_graph_note(step)),
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
