"""Generate the golden fixtures from the REAL reference (yliess86/BayeFormers at /root/reference).

Runs only in the build container (the reference never travels to the GPU box):

    PYTHONPATH=/root/repo python tests/golden/make_golden.py [--skip-bert]

The reference is imported unmodified.  Its only RNG touch-point is `self.normal.sample(self.size)` in
Gaussian.sample (/root/reference/bayeformers/nn/parameters/gaussian.py:100); each Gaussian's `normal` attribute
is replaced by a stub that returns the oracle's Philox epsilon for (seed, sample, 2*layer_id + tensor_id), so the
reference and the HIP path consume identical draws.  Everything written below is DATA: inputs (or the seeds that
regenerate them, with checksums) and the reference's outputs.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import bayeformers  # noqa: E402  (the reference)
import bayeformers.nn as rbnn  # noqa: E402
from bayeformers import to_bayesian as ref_to_bayesian  # noqa: E402

from oracle import bayes_oracle as bo  # noqa: E402

SEED = 0x5EED


class PhiloxNormal:
    """Stand-in for torch.distributions.Normal(0, 1) on one reference Gaussian."""

    def __init__(self, clock, stream_id):
        self.clock, self.stream_id = clock, stream_id

    def sample(self, size):
        n = int(np.prod(tuple(size)))
        z = bo.normals(n, self.clock["seed"], self.clock["sample"], self.stream_id)
        return torch.from_numpy(z).reshape(tuple(size))


def inject(module, clock):
    """Give every reference bnn.Linear a layer_id (registration order) and Philox epsilon."""
    layers = [m for m in module.modules() if isinstance(m, rbnn.Linear)]
    for i, l in enumerate(layers):
        l.weight.normal = PhiloxNormal(clock, 2 * i)
        if isinstance(l.bias, rbnn.Gaussian):
            l.bias.normal = PhiloxNormal(clock, 2 * i + 1)
    return layers


def checksum(module):
    return float(sum(p.detach().double().abs().sum() for p in module.parameters()))


def t2n(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------- single layers
def linear_cases():
    cases = {}

    def run(name, layer, x, S, base):
        clock = {"seed": SEED, "sample": base}
        inject(layer, clock)
        ys, lps, lqs = [], [], []
        with torch.no_grad():
            for s in range(S):
                clock["sample"] = base + s
                ys.append(t2n(layer(x)))
                lps.append(float(layer.log_prior))
                lqs.append(float(layer.log_variational_posterior))
        d = {"x": t2n(x), "w_mu": t2n(layer.weight.mu), "w_rho": t2n(layer.weight.rho), "S": S, "base": base,
             "y": np.stack(ys), "log_prior": np.array(lps, np.float64), "lvp": np.array(lqs, np.float64)}
        if isinstance(layer.bias, rbnn.Gaussian):
            d["b_mu"], d["b_rho"] = t2n(layer.bias.mu), t2n(layer.bias.rho)
        if isinstance(layer.weight_prior, rbnn.Gaussian):
            d["wp_mu"], d["wp_rho"] = t2n(layer.weight_prior.mu), t2n(layer.weight_prior.rho)
            if isinstance(layer.bias, rbnn.Gaussian):
                d["bp_mu"], d["bp_rho"] = t2n(layer.bias_prior.mu), t2n(layer.bias_prior.rho)
        else:
            pr = layer.weight_prior
            d["mixture"] = np.array([float(pr.pi), float(pr.sigma1), float(pr.sigma2)], np.float64)
        for k, v in d.items():
            cases[f"{name}/{k}"] = v

    torch.manual_seed(1)
    run("mix_bias", rbnn.Linear(40, 24), torch.randn(5, 40), 3, 0)
    torch.manual_seed(2)
    run("mix_nobias_oddK", rbnn.Linear(33, 7, bias=False), torch.randn(4, 33), 2, 7)
    torch.manual_seed(3)
    run("mix_custom_prior", rbnn.Linear(16, 10, prior=rbnn.ScaledGaussianMixture(0.25, 0.5, 0.05)),
        torch.randn(3, 16), 2, 11)

    torch.manual_seed(4)
    freq = torch.nn.Linear(64, 48)
    with torch.no_grad():
        freq.weight.mul_(0.2)
        freq.weight[0, :5] = 0.0          # MOPED: delta*|w| = 0 -> log(0) = -inf -> rho := 0
        freq.weight[1, 0] = 1e-7          # below the fp32 resolution of exp(x) - 1
        freq.bias[3] = 0.0
    run("moped", rbnn.Linear.from_frequentist(freq, delta=0.05, freeze=True), torch.randn(6, 64), 3, 100)

    torch.manual_seed(5)
    edge = rbnn.Linear(24, 8)
    with torch.no_grad():
        edge.weight.rho[0, :4] = 25.0      # softplus threshold: sigma = rho
        edge.weight.rho[1, :4] = 20.0
        edge.weight.rho[2, :4] = -30.0     # tiny sigma
        edge.weight.mu[3, :4] = torch.tensor([14.0, -14.2, 5.0, -9.0])  # deep mixture tail, still finite in fp32
        edge.weight.rho[3, :4] = -12.0
    run("edge", edge, torch.randn(2, 24), 2, 3)

    torch.manual_seed(6)
    inf_edge = rbnn.Linear(8, 4)
    with torch.no_grad():
        inf_edge.weight.mu[0, 0] = 15.0    # reference fp32: exp underflow -> log(0) = -inf
        inf_edge.weight.rho[0, 0] = -12.0
    run("edge_inf", inf_edge, torch.randn(2, 8), 1, 5)
    return cases



# ---------------------------------------------------------------------------------------------- C2 at size: 768 -> 768
def linear768_cases():
    """BASELINE configs[1] at FULL size, from the real reference: bnn.Linear(768, 768) with the default
    Uniform init + mixture prior, and the MOPED variant (nn.Linear with w ~ N(0, 0.02^2), delta = 0.05, freeze=True);
    x ~ N(0, 1) of [32, 768] and [4096, 768]; S = 10.  Stored: the seeds that rebuild layer and inputs (with checksums),
    the per-sample log-probs, and a subset of the output rows (the full [10, 4096, 768] would be 126 MB)."""
    S, N, K = 10, 768, 768
    rows = {32: np.array([0, 31]), 4096: np.array([0, 256, 2049, 4095])}
    out = {"S": S, "N": N, "K": K, "rows_32": rows[32], "rows_4096": rows[4096]}

    def build(kind):
        if kind == "mixture":
            torch.manual_seed(768)
            return rbnn.Linear(K, N)
        torch.manual_seed(769)
        freq = torch.nn.Linear(K, N)
        with torch.no_grad():
            freq.weight.normal_(0.0, 0.02)
            freq.bias.normal_(0.0, 0.02)
        return rbnn.Linear.from_frequentist(freq, delta=0.05, freeze=True)

    for kind, base in (("mixture", 0), ("moped", 40)):
        layer = build(kind)
        out[f"{kind}/checksum"] = checksum(layer)
        out[f"{kind}/base"] = base
        clock = {"seed": SEED, "sample": base}
        inject(layer, clock)
        for M in (32, 4096):
            x = torch.randn(M, K, generator=torch.Generator().manual_seed(1000 + M))
            out[f"x{M}_sum"] = float(x.double().abs().sum())
            ys, lps, lqs = [], [], []
            with torch.no_grad():
                for s in range(S):
                    clock["sample"] = base + s
                    ys.append(t2n(layer(x))[rows[M]])
                    lps.append(float(layer.log_prior))
                    lqs.append(float(layer.log_variational_posterior))
            out[f"{kind}/y{M}"] = np.stack(ys)
            out[f"{kind}/log_prior"] = np.array(lps, np.float64)  # the same for both M (same weights, same eps)
            out[f"{kind}/lvp"] = np.array(lqs, np.float64)
    return out

# ---------------------------------------------------------------------------------------------- C1: MLP
class MLP(torch.nn.Module):
    """The 784-512-512-10 MLP of /root/reference/examples/mlp_mnist.py:16-26 (architecture only)."""

    def __init__(self, in_features, hidden, n_classes):
        super().__init__()
        self.mlp = torch.nn.Sequential(
            torch.nn.Linear(in_features, hidden), torch.nn.ReLU(),
            torch.nn.Linear(hidden, hidden), torch.nn.ReLU(),
            torch.nn.Linear(hidden, n_classes), torch.nn.LogSoftmax(dim=1))

    def forward(self, input):
        return self.mlp(input)


def mlp_case():
    S, B, NB = 5, 128, 469
    torch.manual_seed(0)
    model = MLP(784, 512, 10)
    bmodel = ref_to_bayesian(model, delta=0.05)
    csum = checksum(bmodel)  # before any forward: the log-prob Parameters are still 0
    torch.manual_seed(123)
    x = torch.rand(B, 784)
    labels = torch.randint(0, 10, (B,))
    clock = {"seed": SEED, "sample": 0}
    inject(bmodel, clock)
    pred = torch.zeros(S, B, 10)
    lp = torch.zeros(S)
    lq = torch.zeros(S)
    with torch.no_grad():
        for s in range(S):  # the sample loop of examples/mlp_mnist.py:97-100
            clock["sample"] = s
            pred[s] = bmodel(x)
            lp[s] = bmodel.log_prior()
            lq[s] = bmodel.log_variational_posterior()
        nll = torch.nn.functional.nll_loss(pred.mean(0), labels, reduction="sum")  # mlp_mnist.py:103-107
        loss = (lq.mean() - lp.mean()) / NB + nll
    return {"S": S, "B": B, "n_batches": NB, "model_seed": 0, "input_seed": 123, "delta": 0.05,
            "checksum": csum, "x_sum": float(x.double().sum()), "labels": t2n(labels),
            "pred": t2n(pred), "log_prior": t2n(lp).astype(np.float64), "lvp": t2n(lq).astype(np.float64),
            "nll": float(nll), "loss": float(loss)}


# ---------------------------------------------------------------------------------------------- C3: BERT
HIDDEN_AT = [(0, 1), (3, 17), (7, 40), (12, 63), (16, 64), (21, 90), (27, 126), (31, 127)]   # (sequence, token)
HIDDEN_AT_TINY = [(0, 1), (1, 7), (2, 8), (3, 15)]


def bert_case(tiny, samples=None):
    from transformers import BertConfig, BertForSequenceClassification

    if tiny:
        cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                         vocab_size=1000, max_position_embeddings=64)
        S, B, L = 3, 4, 16
    else:
        cfg = BertConfig()
        S, B, L = 10, 32, 128
    if samples is not None:
        S = samples  # BASELINE config 4: S = 64 global sample indices (8 per GPU x 8 GPUs on the HIP side)
    torch.manual_seed(0)
    model = BertForSequenceClassification(cfg).eval()
    bmodel = ref_to_bayesian(model, delta=0.05, freeze=True).eval()
    csum = checksum(bmodel)  # before any forward
    torch.manual_seed(321)
    ids = torch.randint(0, cfg.vocab_size, (B, L))
    mask = torch.ones(B, L, dtype=torch.long)
    labels = torch.randint(0, 2, (B,))
    clock = {"seed": SEED, "sample": 0}
    layers = inject(bmodel, clock)
    logits = torch.zeros(S, B, 2)
    lp = torch.zeros(S, dtype=torch.float64)
    lq = torch.zeros(S, dtype=torch.float64)
    # the reference's LAST-LAYER hidden states at 8 (sequence, token) positions spread over the batch — one per two 256-row
    # bands of the [B L, hidden] activation the GEMMs see — so that rows other than [CLS] are pinned to the reference too
    # (the seq-cls logits read token 0 of the last layer only).  Not stored for the S = 64 fixture (size).
    hidden_at = HIDDEN_AT_TINY if tiny else HIDDEN_AT
    keep_hidden = samples is None
    hidden = torch.zeros(S, len(hidden_at), cfg.hidden_size)
    t0 = time.time()
    with torch.no_grad():
        for s in range(S):  # sample_bayesian, examples/bert_glue.py:63-66
            clock["sample"] = s
            out = bmodel(input_ids=ids, attention_mask=mask, labels=labels, output_hidden_states=keep_hidden)
            logits[s] = out[1]
            if keep_hidden:
                last = out.hidden_states[-1]
                for i, (b, t) in enumerate(hidden_at):
                    hidden[s, i] = last[b, t]
            lp[s] = float(bmodel.log_prior())
            lq[s] = float(bmodel.log_variational_posterior())
        nll = torch.nn.functional.cross_entropy(logits.mean(0), labels)  # bert_glue.py:234
    print(f"  bert tiny={tiny}: {len(layers)} layers, {time.time() - t0:.1f}s for {S} samples", flush=True)
    out = {"S": S, "B": B, "L": L, "model_seed": 0, "input_seed": 321, "delta": 0.05, "n_layers": len(layers),
           "checksum": csum, "ids_sum": int(ids.sum()), "labels": t2n(labels), "logits": t2n(logits),
           "log_prior": t2n(lp), "lvp": t2n(lq), "nll": float(nll)}
    if keep_hidden:
        out["hidden_at"] = np.array(hidden_at, dtype=np.int64)
        out["hidden"] = t2n(hidden)
    return out


def grad_cases():
    """Gradients of the reference's autograd graph (likelihood path only: its log-probs are detached)."""
    out = {}

    def run(name, layer, x, S, base):
        clock = {"seed": SEED, "sample": base}
        inject(layer, clock)
        x = x.clone().requires_grad_(True)
        g = torch.randn(S, x.shape[0], layer.out_features)
        for p_ in layer.parameters():
            p_.grad = None
        loss = 0.0
        for s in range(S):
            clock["sample"] = base + s
            loss = loss + (layer(x) * g[s]).sum()
        loss.backward()
        d = {"x": t2n(x), "g": t2n(g), "S": S, "base": base, "w_mu": t2n(layer.weight.mu), "w_rho": t2n(layer.weight.rho),
             "dx": t2n(x.grad), "dw_rho": t2n(layer.weight.rho.grad)}
        if layer.weight.mu.grad is not None:
            d["dw_mu"] = t2n(layer.weight.mu.grad)
        if isinstance(layer.bias, rbnn.Gaussian):
            d["b_mu"], d["b_rho"] = t2n(layer.bias.mu), t2n(layer.bias.rho)
            d["db_rho"] = t2n(layer.bias.rho.grad)
            if layer.bias.mu.grad is not None:
                d["db_mu"] = t2n(layer.bias.mu.grad)
        for k, v in d.items():
            out[f"{name}/{k}"] = v

    torch.manual_seed(11)
    run("mix_bias", rbnn.Linear(40, 24), torch.randn(5, 40), 3, 0)
    torch.manual_seed(12)
    run("nobias_oddK", rbnn.Linear(33, 7, bias=False), torch.randn(4, 33), 2, 7)
    torch.manual_seed(13)
    freq = torch.nn.Linear(64, 48)
    run("moped_frozen", rbnn.Linear.from_frequentist(freq, delta=0.05, freeze=True), torch.randn(6, 64), 3, 100)
    torch.manual_seed(14)
    big = rbnn.Linear(128, 192)
    with torch.no_grad():
        big.weight.rho[0, :8] = 25.0  # softplus threshold branch of the backward
    run("big", big, torch.randn(96, 128), 2, 9)
    return out


def bert_large_qa_case():
    """BASELINE config 5: to_bayesian(BERT-large QA), S=10, seq=384, batch=16 (examples/bert_squad.py:190-212,:222)."""
    from transformers import BertConfig, BertForQuestionAnswering

    cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    S, B, L = 10, 16, 384
    torch.manual_seed(0)
    model = BertForQuestionAnswering(cfg).eval()
    bmodel = ref_to_bayesian(model, delta=0.05, freeze=True).eval()
    csum = checksum(bmodel)
    torch.manual_seed(654)
    ids = torch.randint(0, cfg.vocab_size, (B, L))
    mask = torch.ones(B, L, dtype=torch.long)
    clock = {"seed": SEED, "sample": 0}
    layers = inject(bmodel, clock)
    start = torch.zeros(S, B, L)
    end = torch.zeros(S, B, L)
    lp = torch.zeros(S, dtype=torch.float64)
    lq = torch.zeros(S, dtype=torch.float64)
    t0 = time.time()
    with torch.no_grad():
        for s in range(S):  # sample_bayesian, examples/bert_squad.py:198-203
            clock["sample"] = s
            out = bmodel(input_ids=ids, attention_mask=mask)
            start[s], end[s] = out[0], out[1]
            lp[s] = float(bmodel.log_prior())
            lq[s] = float(bmodel.log_variational_posterior())
            print(f"    sample {s}: {time.time() - t0:.0f}s", flush=True)
    # round 6: samples 3 and 6 of BOTH heads at four sequences (in the uneven 8-rank split of this configuration — 2, 2, 1, 1, 1,
    # 1, 1, 1 — sample 3 closes rank 1's shard and sample 6 is all of rank 4's: per-sample logits of the middle of the step are
    # now pinned too, not only through the means)
    seqs = [0, 5, 10, 15]
    mid = {f"{head}_s{k}": t2n(t[k][seqs]) for k in (3, 6) for head, t in (("start", start), ("end", end))}
    return {"S": S, "B": B, "L": L, "model_seed": 0, "input_seed": 654, "delta": 0.05, "n_layers": len(layers),
            "checksum": csum, "ids_sum": int(ids.sum()), "start_mean": t2n(start.mean(0)), "end_mean": t2n(end.mean(0)),
            "start_s0": t2n(start[0]), "end_s9": t2n(end[9]), "log_prior": t2n(lp), "lvp": t2n(lq),
            "mid_seqs": np.array(seqs), **mid}


def checkpoint_case():
    """What /root/reference/examples/bert_glue.py:303-309 saves — `b_model.state_dict()` after forwards — for a small
    converted model, in both prior variants (default scale mixture; MOPED, delta given), together with the
    reference's forward of that model for S Philox samples.  The HIP path must load these state dicts as they are
    (duplicated shared-prior keys, the two per-layer log-prob scalars included) and reproduce the outputs."""
    out = {}
    for name, kw in (("mixture", {}), ("moped", {"delta": 0.07, "freeze": True})):
        torch.manual_seed(21)
        net = torch.nn.Sequential(torch.nn.Linear(48, 96), torch.nn.Tanh(), torch.nn.Linear(96, 32), torch.nn.Tanh(),
                                  torch.nn.Linear(32, 6, bias=False))
        torch.manual_seed(22)
        bmodel = ref_to_bayesian(net, **kw)
        with torch.no_grad():  # a "trained" posterior: move rho (and mu where it is free) away from the init
            g = torch.Generator().manual_seed(23)
            for n_, p_ in bmodel.named_parameters():
                if n_.endswith("rho") and "prior" not in n_:
                    p_.add_(0.3 * torch.randn(p_.shape, generator=g))
                if n_.endswith("mu") and "prior" not in n_ and p_.requires_grad:
                    p_.add_(0.05 * torch.randn(p_.shape, generator=g))
        S, B, base = 4, 24, 40
        torch.manual_seed(24)
        x = torch.randn(B, 48)
        clock = {"seed": SEED, "sample": base}
        inject(bmodel, clock)
        ys, lps, lqs = [], [], []
        with torch.no_grad():
            for s_ in range(S):
                clock["sample"] = base + s_
                ys.append(t2n(bmodel(x)))
                lps.append(float(bmodel.log_prior()))
                lqs.append(float(bmodel.log_variational_posterior()))
        sd = bmodel.state_dict()  # taken after the forward of sample base + S - 1, as a training script would
        out[f"{name}/keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[f"{name}/sd/{k}"] = t2n(v)
        out[f"{name}/x"], out[f"{name}/y"] = t2n(x), np.stack(ys)
        out[f"{name}/log_prior"], out[f"{name}/lvp"] = np.array(lps, np.float64), np.array(lqs, np.float64)
        out[f"{name}/S"], out[f"{name}/base"] = S, base
        out[f"{name}/requires_grad"] = np.array(sorted(n_ for n_, p_ in bmodel.named_parameters() if p_.requires_grad))
    return out


def conversion_case():
    """to_bayesian / from_frequentist numerics and state-dict layout (bayeformers/__init__.py:19-63)."""
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Linear(12, 9), torch.nn.Tanh(), torch.nn.Linear(9, 4, bias=False))
    torch.manual_seed(8)
    moped = ref_to_bayesian(net, delta=0.1, freeze=True)
    torch.manual_seed(8)
    plain = ref_to_bayesian(net)
    d = {"moped_keys": np.array(sorted(moped.state_dict().keys())),
         "plain_keys": np.array(sorted(plain.state_dict().keys())),
         "moped_requires_grad": np.array(sorted(n for n, p in moped.named_parameters() if p.requires_grad)),
         "plain_requires_grad": np.array(sorted(n for n, p in plain.named_parameters() if p.requires_grad))}
    for k, v in moped.state_dict().items():
        d["moped/" + k] = t2n(v)
    for k, v in plain.state_dict().items():
        d["plain/" + k] = t2n(v)
    return d


# ---------------------------------------------------------------------------------------------- training step (tiny BERT)
def bert_train_case():
    """One training step of the reference on the tiny BERT of bert_case(True): the sample loop of
    examples/bert_glue.py:63-66 WITH gradients, nll = CrossEntropy(mean logits) and
    loss = (lvp - log_prior) / n_batches + nll (bert_glue.py:234-235), loss.backward() (bert_glue.py:239).
    Dropout is off (eval mode) so that the step is a function of the Philox epsilon only.  Stored: the gradient of every
    trainable tensor as (sum, sum |g|, max |g|) and, in full, the rho gradients of one layer of each kind."""
    from transformers import BertConfig, BertForSequenceClassification

    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                     vocab_size=1000, max_position_embeddings=64)
    S, B, L, NB = 3, 4, 16, 2105
    torch.manual_seed(0)
    model = BertForSequenceClassification(cfg).eval()
    bmodel = ref_to_bayesian(model, delta=0.05, freeze=True).eval()
    csum = checksum(bmodel)
    torch.manual_seed(321)
    ids = torch.randint(0, cfg.vocab_size, (B, L))
    mask = torch.ones(B, L, dtype=torch.long)
    labels = torch.randint(0, 2, (B,))
    clock = {"seed": SEED, "sample": 0}
    inject(bmodel, clock)
    logits = torch.zeros(S, B, 2)
    lp = torch.zeros(S, B)
    lq = torch.zeros(S, B)
    for s in range(S):
        clock["sample"] = s
        logits[s] = bmodel(input_ids=ids, attention_mask=mask, labels=labels)[1]
        lp[s] = bmodel.log_prior()
        lq[s] = bmodel.log_variational_posterior()
    nll = torch.nn.functional.cross_entropy(logits.mean(0).view(-1, 2), labels.view(-1))
    loss = (lq.mean() - lp.mean()) / NB + nll
    loss.backward()
    out = {"S": S, "B": B, "L": L, "n_batches": NB, "model_seed": 0, "input_seed": 321, "delta": 0.05, "checksum": csum,
           "ids_sum": int(ids.sum()), "labels": t2n(labels), "logits": t2n(logits), "nll": float(nll.detach()), "loss": float(loss.detach())}
    full = ("bert.encoder.layer.0.attention.self.query", "bert.encoder.layer.1.attention.self.value",
            "bert.encoder.layer.0.attention.output.dense", "bert.encoder.layer.1.intermediate.dense",
            "bert.encoder.layer.0.output.dense", "bert.pooler.dense", "classifier")
    names = []
    for name, p_ in bmodel.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad.detach().double()
        names.append(name)
        out[f"stat/{name}"] = np.array([float(g.sum()), float(g.abs().sum()), float(g.abs().max())], np.float64)
        if name.endswith(".rho") and "prior" not in name and name.rsplit(".", 2)[0].replace("model.", "", 1) in full:
            out[f"grad/{name}"] = t2n(p_.grad)
    out["names"] = np.array(names)
    print(f"  bert train: {len(names)} tensors with a gradient, {sum(k.startswith('grad/') for k in out)} stored in full")
    return out



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-bert", action="store_true")
    ap.add_argument("--only-grads", action="store_true")
    ap.add_argument("--only-bert-large", action="store_true")
    ap.add_argument("--only-c3", action="store_true", help="BERT tiny + BASELINE config 3: BERT-base, S = 10 (about a minute of CPU)")
    ap.add_argument("--only-c4", action="store_true", help="BASELINE config 4: BERT-base, S = 64 (about 6 minutes of CPU)")
    ap.add_argument("--only-checkpoint", action="store_true")
    ap.add_argument("--only-bert-train", action="store_true")
    ap.add_argument("--only-linear768", action="store_true", help="BASELINE config 2 at full size (a few seconds of CPU)")
    args = ap.parse_args()
    torch.set_num_threads(8)
    if args.only_linear768:
        np.savez_compressed(os.path.join(HERE, "linear768_c2.npz"), **linear768_cases())
        return
    if args.only_c3:
        np.savez_compressed(os.path.join(HERE, "bert_tiny.npz"), **bert_case(True))
        np.savez_compressed(os.path.join(HERE, "bert_c3.npz"), **bert_case(False))
        return
    if args.only_c4:
        np.savez_compressed(os.path.join(HERE, "bert_c4.npz"), **bert_case(False, samples=64))
        return
    if args.only_checkpoint:
        np.savez_compressed(os.path.join(HERE, "checkpoint.npz"), **checkpoint_case())
        return
    if args.only_bert_train:
        np.savez_compressed(os.path.join(HERE, "bert_tiny_train.npz"), **bert_train_case())
        return
    if args.only_bert_large:
        np.savez_compressed(os.path.join(os.environ.get("BF_GOLDEN_OUT", HERE), "bert_large_qa_c5.npz"), **bert_large_qa_case())
        return
    if args.only_grads:
        np.savez_compressed(os.path.join(HERE, "linear_grads.npz"), **grad_cases())
        return
    print("eps KAT"); np.savez_compressed(os.path.join(HERE, "eps_kat.npz"), seed=SEED,
                                         z_s0_str0=bo.normals(64, SEED, 0, 0), z_s9_str5_off3=bo.normals(64, SEED, 9, 5, 3))
    print("linear cases"); np.savez_compressed(os.path.join(HERE, "linear_cases.npz"), **linear_cases())
    print("linear 768 C2"); np.savez_compressed(os.path.join(HERE, "linear768_c2.npz"), **linear768_cases())
    print("grad cases"); np.savez_compressed(os.path.join(HERE, "linear_grads.npz"), **grad_cases())
    print("conversion"); np.savez_compressed(os.path.join(HERE, "conversion.npz"), **conversion_case())
    print("checkpoint"); np.savez_compressed(os.path.join(HERE, "checkpoint.npz"), **checkpoint_case())
    print("mlp C1"); np.savez_compressed(os.path.join(HERE, "mlp_c1.npz"), **mlp_case())
    print("bert tiny"); np.savez_compressed(os.path.join(HERE, "bert_tiny.npz"), **bert_case(True))
    print("bert tiny training step"); np.savez_compressed(os.path.join(HERE, "bert_tiny_train.npz"), **bert_train_case())
    if not args.skip_bert:
        print("bert base C3 (about a minute of CPU)"); np.savez_compressed(os.path.join(HERE, "bert_c3.npz"), **bert_case(False))
        print("bert base C4, S = 64 (about 6 minutes of CPU)")
        np.savez_compressed(os.path.join(HERE, "bert_c4.npz"), **bert_case(False, samples=64))


if __name__ == "__main__":
    main()
