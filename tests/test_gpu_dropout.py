"""Training-mode dropout inside the fused kernels (the reference trains in .train(),
/root/reference/examples/bert_glue.py:221, with HuggingFace's p = 0.1 on the attention probabilities and on every dense
output in front of a residual + LayerNorm).  The masks are the build's own Philox contract (csrc/bf_philox.h), like
epsilon: each test builds the mask from the ORACLE's restatement of that contract, applies it with plain torch ops
(fp64), and compares outputs and gradients with the kernels."""
import numpy as np
import pytest
import torch

import bayeformers_amd as bf
from bayeformers_amd import ops
from bayeformers_amd import random as bfr
from oracle import bayes_oracle as bo

pytestmark = pytest.mark.gpu
SEED = 0x5EED


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,T,p,masked", [(3, 2, 128, 0.1, False), (2, 3, 128, 0.25, True), (2, 2, 256, 0.1, False),
                                            (2, 2, 384, 0.1, True)])
def test_attention_dropout_matches_torch_with_the_same_mask(dtype, B, H, T, p, masked):
    """softmax -> dropout -> V, forward and backward (one tile: the single kernel; longer sequences: the dq and dk / dv
    kernels, which find their keep bits per (query tile, key tile)): kernels vs fp64 torch with the oracle's mask."""
    D, call, site = 64, 5, 9
    g = torch.Generator(device="cuda").manual_seed(B * 100 + T)
    qkv = [torch.randn(B, T, H * D, device="cuda", generator=g).to(dtype).requires_grad_(True) for _ in range(3)]
    q, k, v = (t.view(B, T, H, D).transpose(1, 2) for t in qkv)  # [B, H, T, D] views, as the attention hook gets them
    key_mask = None
    if masked:
        key_mask = torch.zeros(B, T, device="cuda")
        key_mask[0, T - 17:] = float("-inf")
        key_mask[1, :5] = float("-inf")
    drop = ops.Dropout(p, SEED, call, site)
    scale = D ** -0.5
    keep = torch.from_numpy(bo.attention_keep_mask(B, H, T, p, SEED, call, site)).cuda().double()
    assert abs(float(keep.mean()) - (1 - p)) < 0.01

    # reference (fp64 autograd on the same 16-bit inputs)
    ref_in = [t.detach().double().requires_grad_(True) for t in qkv]
    rq, rk, rv = (t.view(B, T, H, D).transpose(1, 2) for t in ref_in)
    s = rq @ rk.transpose(-1, -2) * scale
    if key_mask is not None:
        s = s + key_mask[:, None, None, :].double()
    pr = torch.softmax(s, dim=-1) * keep * bo.dropout_keep_scale(p)
    ref = (pr @ rv).transpose(1, 2)  # [B, T, H, D]

    out = ops.AttentionFn.apply(q, k, v, key_mask, None, scale, drop)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert (out.double() - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-3
    # a kept / dropped pattern that is not the mask would show as O(1) errors: check a second call number differs
    with torch.no_grad():
        other = ops.attention_forward(q, k, v, key_mask, scale, None, drop=ops.Dropout(p, SEED, call + 1, site))
    assert (other.double() - ref).abs().max().item() > 20 * tol * ref.abs().max().item()
    go = torch.randn(B, T, H, D, device="cuda", generator=g).to(dtype)
    out.backward(go)
    ref.backward(go.double())
    for got, want, name in zip(qkv, ref_in, "qkv"):
        gtol = (2.0 ** -6 if dtype == torch.bfloat16 else 2.0 ** -9) * want.grad.abs().max().item()
        assert (got.grad.double() - want.grad).abs().max().item() <= gtol, name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,N,p", [(37, 768, 0.1), (64, 1024, 0.3), (5, 200, 0.1), (9, 3072, 0.1)])
def test_add_layernorm_dropout_matches_torch_with_the_same_mask(dtype, rows, N, p):
    """LayerNorm(dropout(x) + residual): forward, and the backward that regenerates the mask (dx = dz o keep / (1 - p),
    dresidual = dz), vs fp64 torch with the oracle's mask: group = 8 consecutive features, index row * (N / 8) + n / 8."""
    call, site = 3, 4
    g = torch.Generator(device="cuda").manual_seed(rows * 7 + N)
    x = torch.randn(rows, N, device="cuda", generator=g).to(dtype).requires_grad_(True)
    res = torch.randn(rows, N, device="cuda", generator=g).to(dtype).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(N, device="cuda", generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(N, device="cuda", generator=g)).requires_grad_(True)
    drop = ops.Dropout(p, SEED, call, site)
    keep = torch.from_numpy(bo.dropout_keep(0, rows * (N // 8), p, SEED, call, site)).reshape(rows, N).cuda().double()
    rx, rr, rg, rb = (t.detach().double().requires_grad_(True) for t in (x, res, gamma, beta))
    z = rx * keep * bo.dropout_keep_scale(p) + rr
    ref = torch.nn.functional.layer_norm(z, (N,), rg, rb, 1e-12)
    out = ops.AddLayerNormFn.apply(x, res, gamma, beta, 1e-12, drop)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert (out.double() - ref).abs().max().item() <= tol * ref.abs().max().item()
    go = torch.randn(rows, N, device="cuda", generator=g).to(dtype)
    out.backward(go)
    ref.backward(go.double())
    for got, want, name in ((x, rx, "x"), (res, rr, "residual"), (gamma, rg, "gamma"), (beta, rb, "beta")):
        gt = (2.0 ** -6 if dtype == torch.bfloat16 else 2e-5) * want.grad.abs().max().item()
        assert (got.grad.double() - want.grad).abs().max().item() <= gt, name
    # exactly the dropped features carry no gradient
    assert torch.equal(x.grad == 0, keep == 0) or float(((x.grad == 0) != (keep == 0)).double().mean()) < 1e-3


def test_train_mode_bert_takes_the_fused_paths_and_repeats_its_masks():
    """A converted tiny BERT in .train() (HF dropout 0.1): attention and the Output blocks run the kernels with dropout (no
    framework fallback), two forwards from the same RNG state agree bit for bit (masks are a function of seed / call /
    site), a later forward draws new masks, and backward runs through the regenerated masks."""
    from transformers import BertConfig, BertForSequenceClassification

    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=100,
                     max_position_embeddings=128)
    model = bf.to_bayesian(BertForSequenceClassification(cfg), delta=0.05, freeze=True).cuda()
    bf.fuse_activations(model); bf.fuse_residual_layernorm(model); bf.fuse_shared_inputs(model)
    assert bf.fuse_attention(model)
    bf.fuse_embeddings(model)
    model = model.to(torch.bfloat16).train()
    ids = torch.randint(0, 100, (4, 128), device="cuda")
    inputs = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    calls = {"attn": 0, "ln": 0}
    orig_attn, orig_ln = ops.AttentionFn.forward, ops.AddLayerNormFn.forward

    def count_attn(ctx, *a):
        calls["attn"] += a[6] is not None and a[6].p > 0
        return orig_attn(ctx, *a)

    def count_ln(ctx, *a):
        calls["ln"] += len(a) > 5 and a[5] is not None and a[5].p > 0
        return orig_ln(ctx, *a)

    ops.AttentionFn.forward, ops.AddLayerNormFn.forward = staticmethod(count_attn), staticmethod(count_ln)
    try:
        from bayeformers_amd.sampling import sample_bayesian

        def run():
            bf.manual_seed(SEED)
            torch.manual_seed(1)  # the embedding block's dropout is the framework's
            raw, mean, lp, lq = sample_bayesian(model, inputs, 3)
            return mean[0]

        a = run()
        assert calls == {"attn": 2, "ln": 4}, calls  # both layers' attention, SelfOutput + Output of both layers
        b = run()
        assert torch.equal(a, b)
        c = sample_bayesian(model, inputs, 3)[1][0]  # next sample indices AND next dropout call
        assert not torch.equal(a, c)
        a.float().sum().backward()
        grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
        assert grads and all(torch.isfinite(g).all() for g in grads)
    finally:
        ops.AttentionFn.forward, ops.AddLayerNormFn.forward = orig_attn, orig_ln


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_a_shard_draws_the_masks_of_its_global_samples(dtype):
    """the in-kernel dropout masks are a function of the GLOBAL Monte-Carlo sample (like epsilon), not of where a
    sample lies in the rank's batch: the slabs of samples [2, 4) run as a shard of their own — Dropout(origin=(2, 2)), what
    bnn.Model.monte_carlo(span=(2, 4)) hands the kernels — come out bit for bit as in the unsharded run of all four, forward
    and backward, for the residual + LayerNorm block and for the attention probabilities."""
    S, rows_per_sample, N, p = 4, 96, 256, 0.2
    g = torch.Generator(device="cuda").manual_seed(11)
    x, res, go = (torch.randn(S * rows_per_sample, N, device="cuda", generator=g).to(dtype) for _ in range(3))
    gamma, beta = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    full = ops.Dropout(p, SEED, 7, 3)
    shard = ops.Dropout(p, SEED, 7, 3, origin=(2, 2))
    lo = 2 * rows_per_sample
    y_full = ops.add_layernorm(x, res, gamma, beta, 1e-12, full)
    y_shard = ops.add_layernorm(x[lo:].contiguous(), res[lo:].contiguous(), gamma, beta, 1e-12, shard)
    assert torch.equal(y_full[lo:], y_shard)
    assert not torch.equal(y_full[:lo], y_shard)       # (and not the masks of samples 0, 1)
    b_full = ops.add_layernorm_backward(x, res, gamma, go, 1e-12, full)
    b_shard = ops.add_layernorm_backward(x[lo:].contiguous(), res[lo:].contiguous(), gamma, go[lo:].contiguous(), 1e-12, shard)
    assert torch.equal(b_full[0][lo:], b_shard[0]) and torch.equal(b_full[3][lo:], b_shard[3])     # dz, dx
    if dtype == torch.float32:
        return
    # attention probabilities: sequences of samples 2, 3 (Bs sequences per sample)
    Bs, H, T, D = 3, 2, 128, 64
    q, k, v = (torch.randn(S * Bs, T, H * D, device="cuda", generator=g).to(dtype).view(S * Bs, T, H, D).transpose(1, 2) for _ in range(3))
    o_full = ops.attention_forward(q, k, v, None, 0.125, None, drop=full)
    o_shard = ops.attention_forward(q[2 * Bs:], k[2 * Bs:], v[2 * Bs:], None, 0.125, None, drop=shard)
    assert torch.equal(o_full[2 * Bs:], o_shard) and not torch.equal(o_full[:2 * Bs], o_shard)


def test_sharded_train_mode_forward_equals_the_unsharded_one():
    """The same through the model: a converted BERT in .train() (embedding dropout off: that one is the framework's RNG) run as
    two shards of two samples gives, sample by sample, the logits of the unsharded four-sample forward."""
    from transformers import BertConfig, BertForSequenceClassification

    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=300,
                     max_position_embeddings=128)
    torch.manual_seed(0)
    bmodel = bf.to_bayesian(BertForSequenceClassification(cfg), delta=0.05, freeze=True).cuda().to(torch.bfloat16)
    bf.fuse_activations(bmodel), bf.fuse_residual_layernorm(bmodel), bf.fuse_shared_inputs(bmodel)
    bf.fuse_attention(bmodel), bf.fuse_embeddings(bmodel)
    bmodel.train()
    bmodel.model.bert.embeddings.dropout.p = 0.0
    bmodel.model.dropout.p = 0.0                         # the pooled-output dropout is the framework's too
    ids = torch.randint(0, 300, (3, 128), generator=torch.Generator().manual_seed(1)).cuda()
    S = 4

    def run(start, count):
        bf.manual_seed(SEED)
        with torch.no_grad(), bmodel.monte_carlo(count, span=(start, S)):
            out = bmodel(input_ids=ids.repeat(count, 1), attention_mask=torch.ones_like(ids).repeat(count, 1))
        return out.logits.float().view(count, 3, -1)

    whole = run(0, S)
    assert not torch.equal(whole[0], whole[2])
    for start in (0, 2):
        part = run(start, 2)
        assert torch.equal(part, whole[start:start + 2]), start
