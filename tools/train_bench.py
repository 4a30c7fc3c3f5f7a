"""Training-step timing of the Bayesian path (fwd + ELBO + backward + optimiser) — row f-1 of SURVEY §8.

    python tools/train_bench.py [--workload bert_base|linear768] [--steps 10]

Prints the mean step time and, for the single layer, the per-call time of bf_linear_bwd against the time two
ideal MFMA GEMMs of the same shapes (dx and dW) would take at the forward kernel's measured rate.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bayeformers_amd as bf  # noqa: E402
import bayeformers_amd.nn as bnn  # noqa: E402
from bayeformers_amd.sampling import elbo, sample_bayesian  # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def bert(steps, S=10, B=32, L=128):
    from transformers import BertConfig, BertForSequenceClassification

    torch.manual_seed(0)
    cfg = BertConfig()
    model = BertForSequenceClassification(cfg)
    bmodel = bf.to_bayesian(model, delta=0.05, freeze=True).cuda().to(torch.bfloat16).eval()
    params = [p for p in bmodel.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-5)
    ids = torch.randint(0, cfg.vocab_size, (B, L)).cuda()
    labels = torch.randint(0, 2, (B,)).cuda()
    inputs = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

    def fwd():
        with torch.no_grad():
            raw, mean, lp, lq = sample_bayesian(bmodel, inputs, S)
            return elbo(lp, lq, torch.nn.functional.cross_entropy(mean[0].float(), labels).double(), 2105)

    def train():
        opt.zero_grad(set_to_none=True)
        raw, mean, lp, lq = sample_bayesian(bmodel, inputs, S)
        loss = elbo(lp, lq, torch.nn.functional.cross_entropy(mean[0].float(), labels).double(), 2105)
        loss.backward()
        opt.step()

    tf = timed(fwd, steps)
    tt = timed(train, steps)
    print(f"BERT-base S={S} B={B} L={L} bf16: forward+ELBO {tf * 1e3:.2f} ms, training step {tt * 1e3:.2f} ms "
          f"({S / tt:.0f} MC-samples/s trained), ratio {tt / tf:.2f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


def linear(steps, S=10, M=4096, N=768, K=768):
    layer = bnn.Linear(K, N).cuda()
    model = bnn.Model(layer)
    x = torch.randn(S * M, K, device="cuda").bfloat16().requires_grad_(True)
    gy = torch.randn(S * M, N, device="cuda").bfloat16()

    def fwd():
        with torch.no_grad(), model.monte_carlo(S):
            return model(x)

    def fb():
        x.grad = None
        with model.monte_carlo(S):
            y = model(x)
        y.backward(gy)

    tf = timed(fwd, steps)
    tb = timed(fb, steps)
    flops = 2.0 * S * M * N * K
    print(f"bnn.Linear {N}x{K}, x=[{S}x{M},{K}] bf16: fwd {tf * 1e6:.0f} us, fwd+bwd {tb * 1e6:.0f} us; "
          f"bwd alone {(tb - tf) * 1e6:.0f} us vs 2 GEMMs at 1000 TF = {2 * flops / 1e15 * 1e6:.0f} us")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="both")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    if a.workload in ("linear768", "both"):
        linear(a.steps)
        linear(a.steps, N=3072, K=768)
        linear(a.steps, N=768, K=3072)
    if a.workload in ("bert_base", "both"):
        bert(a.steps)
