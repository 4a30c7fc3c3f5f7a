"""The S-sharded training step (bayeformers_amd.training) on CPU: gloo, world sizes 2 and 3.  The HIP forward is replaced
by a differentiable stub whose per-sample weights are a pure function of the GLOBAL sample index — the property the
Philox contract gives the real kernels — so the ranks' summed gradients must equal the single-process gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bayeformers_amd.nn as bnn
from bayeformers_amd import random as bfr
from bayeformers_amd.training import GradientBuckets, clip_gradients, grad_norm, training_step


class NoisyNet(bnn.Model):
    """y_s = x (W * (1 + 0.3 sin(global sample index + k))) + b: a 'sampled' weight per Monte-Carlo sample."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.w = torch.nn.Parameter(torch.randn(5, 3, generator=g))
        self.b = torch.nn.Parameter(torch.randn(3, generator=g))
        self.unused = torch.nn.Parameter(torch.ones(2))  # never receives a gradient: its bucket is sent by finish()

    def forward(self, x):
        ctx = bfr.STATE.ctx
        S, base = ctx.S, ctx.sample_base
        B = x.shape[0] // S
        idx = base + torch.arange(S, dtype=torch.float32)
        scale = 1.0 + 0.3 * torch.sin(idx[:, None, None] + torch.arange(3, dtype=torch.float32)[None, None, :])
        y = torch.einsum("sbi,sio->sbo", x.view(S, B, 5), self.w[None] * scale) + self.b
        self._lp_buf = torch.stack([-(idx.double() ** 2), 2.0 * idx.double()], dim=1)[None]
        return y.reshape(S * B, 3)


def _run(S, steps, group_used, own_buckets=True):
    torch.manual_seed(0)
    x = torch.randn(4, 5)
    labels = torch.tensor([0, 2, 1, 1])
    model = NoisyNet()
    params = list(model.parameters())
    opt = torch.optim.SGD(params, lr=0.5)
    # small buckets: w alone, b + unused together; own_buckets=False: training_step has to see to the reduction itself
    buckets = GradientBuckets(params, bucket_bytes=300) if own_buckets else None
    bfr.manual_seed(11)
    grads = None
    for _ in range(steps):
        loss = training_step(model, x, S, lambda mean: torch.nn.functional.cross_entropy(mean[0], labels), opt, n_batches=7,
                             buckets=buckets, max_grad_norm=0.05)
        if grads is None:
            grads = [p.grad.clone() for p in (model.w, model.b)]
    return float(loss), [g.numpy() for g in grads], [p.detach().numpy().copy() for p in (model.w, model.b)]


def _worker(rank, world, port, q, S, steps, own_buckets=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _run(S, steps, True, own_buckets))
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_and_clip_single_process():
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2)),
          torch.nn.Parameter(torch.randn(6).to(torch.bfloat16))]
    buckets = GradientBuckets(ps, bucket_bytes=600)
    # one bucket per dtype and size limit (600 bytes): fp32 [2x2, 5, 7x3] = 120 bytes together, bf16 [6]
    assert sorted(len(b[1]) for b in buckets.buckets) == [1, 3]
    buckets.zero()
    loss = (ps[0] ** 2).sum() + (3 * ps[1]).sum() + ps[2].sum() + (ps[3].float() * 2).sum()
    loss.backward()
    buckets.finish()
    for p in ps:
        assert p.grad.data_ptr() == buckets._views[p].data_ptr()
    np.testing.assert_allclose(ps[0].grad.numpy(), 2 * ps[0].detach().numpy(), rtol=1e-6)
    ref = [p.grad.clone() for p in ps]
    tn = torch.sqrt(sum((g.float() ** 2).sum() for g in ref))
    assert float(grad_norm(buckets.flats())) == pytest.approx(float(tn), rel=1e-3)  # the padding is zero; a bf16 bucket's norm is bf16
    # not a fused optimizer: gradients scaled in place, exactly as torch.nn.utils.clip_grad_norm_ scales them
    opt = torch.optim.SGD(ps, lr=0.1)
    total = clip_gradients(opt, buckets.flats(), 0.5)
    assert float(total) == pytest.approx(float(tn), rel=1e-3)
    for p, g in zip(ps, ref):
        np.testing.assert_allclose(p.grad.float().numpy(), (g.float() * min(1.0, 0.5 / (float(tn) + 1e-6))).numpy(), rtol=1e-2)
    # a second step starts from nothing again; parameters without a gradient read zero
    buckets.zero()
    assert all(p.grad is None for p in ps)
    (ps[1] * 2).sum().backward()
    buckets.finish()
    assert ps[1].grad.data_ptr() == buckets._views[ps[1]].data_ptr() and float(ps[1].grad.sum()) == 10.0
    assert float(ps[0].grad.abs().sum()) == 0.0 and float(ps[2].grad.abs().sum()) == 0.0


def test_clip_through_a_fused_optimizer_matches_scaled_gradients():
    """A fused torch optimizer takes the clipping factor as its `grad_scale` input: the update equals the one from
    gradients scaled by torch.nn.utils.clip_grad_norm_."""
    def run(fused_path):
        torch.manual_seed(2)
        ps = [torch.nn.Parameter(torch.randn(9, 4)), torch.nn.Parameter(torch.randn(11))]
        opt = torch.optim.AdamW(ps, lr=0.05, eps=1e-8, weight_decay=0.0, fused=True)
        for k in range(3):
            opt.zero_grad(set_to_none=True)
            ((ps[0] ** 2).sum() * (k + 1) + (ps[1] ** 3).sum()).backward()
            if fused_path:
                clip_gradients(opt, [p.grad for p in ps], 0.7)
            else:
                torch.nn.utils.clip_grad_norm_(ps, 0.7)
            opt.step()
        return [p.detach().clone() for p in ps]

    for a, b in zip(run(True), run(False)):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("world,S,own_buckets", [(2, 6, True), (3, 10, True), (2, 5, False)])
def test_sharded_training_step_matches_single_process(world, S, own_buckets):
    """Two optimisation steps (ELBO of the mean logits, backward, clip at 0.05, SGD) on `world` ranks that each
    backpropagate their own slice of the S samples: the first step's summed gradients and the parameters after both steps
    equal the single-process run's on every rank (S = 10 over 3 ranks: shards of 4, 3, 3).  own_buckets=False: the caller
    passes no GradientBuckets — training_step must still reduce the gradients over the ranks (it builds the buckets of the
    optimizer itself), or the ranks would step apart."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, S, 2, own_buckets)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    loss, grads, params = _run(S, 2, False)
    for rank, l, g, w in res:
        assert l == pytest.approx(loss, rel=1e-6)
        for a, b in zip(g, grads):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7)
        for a, b in zip(w, params):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7)


def test_clipping_factor_does_not_outlive_its_step():
    """clip_gradients hands a fused optimizer its factor through `grad_scale`; training_step removes it after the update,
    so a later step without clipping is not divided by a stale factor."""
    torch.manual_seed(0)
    x = torch.randn(4, 5)
    labels = torch.tensor([0, 2, 1, 1])

    def nll(mean):
        return torch.nn.functional.cross_entropy(mean[0], labels)

    def run(clip_first):
        model = NoisyNet()
        opt = torch.optim.AdamW(list(model.parameters()), lr=0.05, weight_decay=0.0, fused=True)
        bfr.manual_seed(11)
        if clip_first:
            training_step(model, x, 3, nll, opt, n_batches=7, max_grad_norm=0.01)
            assert not hasattr(opt, "grad_scale") and not hasattr(opt, "found_inf")
            return None
        return model, opt

    run(True)
    # two optimizers from the same start: one clipped step then an unclipped one == the same two steps where the second
    # never saw a grad_scale attribute (checked above); here: an unclipped step right after a clipped one changes the
    # parameters exactly as torch's own unclipped update would
    model, opt = run(False)
    training_step(model, x, 3, nll, opt, n_batches=7, max_grad_norm=0.01)
    before = [p.detach().clone() for p in model.parameters()]
    state = {k: {n: (v.clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in opt.state.items()}
    training_step(model, x, 3, nll, opt, n_batches=7, max_grad_norm=None)
    after = [p.detach().clone() for p in model.parameters()]
    # reference: restore, recompute the same gradients, plain optimizer.step()
    with torch.no_grad():
        for p, b in zip(model.parameters(), before):
            p.copy_(b)
    for k, st in state.items():
        for n, v in st.items():
            if torch.is_tensor(v):
                opt.state[k][n].copy_(v)
            else:
                opt.state[k][n] = v
    bfr.manual_seed(11, next_sample=3)
    from bayeformers_amd.sampling import elbo, sample_bayesian
    opt.zero_grad(set_to_none=True)
    _, mean, lp, lq = sample_bayesian(model, x, 3)
    elbo(lp, lq, nll(mean).double(), 7).backward()
    opt.step()
    for a, p in zip(after, model.parameters()):
        np.testing.assert_allclose(a.numpy(), p.detach().numpy(), rtol=1e-6, atol=1e-8)


def test_gradient_written_in_place_by_its_producer_needs_no_copy():
    """The Bayesian layers' backward kernels write their parameter gradients straight into the bucket slot
    (ops.linear_backward: `param._bf_grad_sink`) and hand autograd None.  Same protocol here with a stand-in autograd
    function: the slot holds the gradient, the bucket closes when the in-place and the hooked gradients have all arrived,
    the hook that autograd may still fire with None is ignored, and nothing is written outside a step."""
    class InPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            ctx.w = w
            return x @ w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            sink = getattr(ctx.w, "_bf_grad_sink", None)
            slot = sink.slot(ctx.w) if sink is not None else None
            if slot is not None:
                torch.matmul(x.t(), g, out=slot)
                sink.arrived(ctx.w)
                return g @ w.t(), None
            return g @ w.t(), x.t() @ g

    torch.manual_seed(5)
    w = torch.nn.Parameter(torch.randn(4, 3))
    b = torch.nn.Parameter(torch.randn(3))
    x = torch.randn(6, 4)
    buckets = GradientBuckets([w, b], bucket_bytes=1 << 20)  # one bucket holding both
    assert buckets.slot(w) is None  # no step open: the producer must take the ordinary path
    (InPlace.apply(x, w) + b).sum().backward()
    ref_w, ref_b = w.grad.clone(), b.grad.clone()
    for _ in range(2):  # two steps: the state of the first must not leak into the second
        buckets.zero()
        assert buckets.slot(w) is buckets._views[w]
        (InPlace.apply(x, w) + b).sum().backward()
        buckets.finish()
        assert w.grad.data_ptr() == buckets._views[w].data_ptr() and b.grad.data_ptr() == buckets._views[b].data_ptr()
        np.testing.assert_allclose(w.grad.numpy(), ref_w.numpy(), rtol=1e-6)
        np.testing.assert_allclose(b.grad.numpy(), ref_b.numpy(), rtol=1e-6)
    buckets.remove()
    assert not hasattr(w, "_bf_grad_sink")


def test_second_gradient_for_a_sunk_parameter_is_never_dropped():
    """with the opt-in KL gradient a mu / rho gets two gradients per backward.  A producer that has written the
    first one into the bucket slot (and closed the bucket) must not see the second one silently ignored: with
    set_kl_gradient(True) the slot is not offered at all (autograd sums, the hook copies the sum); a second defined
    gradient for an arrived parameter raises."""
    import bayeformers_amd as bf

    class InPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            ctx.w = w
            return x @ w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            sink = getattr(ctx.w, "_bf_grad_sink", None)
            slot = sink.slot(ctx.w) if sink is not None else None
            if slot is not None:
                torch.matmul(x.t(), g, out=slot)
                sink.arrived(ctx.w)
                return g @ w.t(), None
            return g @ w.t(), x.t() @ g

    torch.manual_seed(7)
    w = torch.nn.Parameter(torch.randn(4, 3))
    x = torch.randn(6, 4)
    loss = lambda: InPlace.apply(x, w).sum() + (w * w).sum()   # the second term plays the KL gradient
    loss().backward()
    ref = w.grad.clone()
    buckets = GradientBuckets([w], bucket_bytes=1 << 20)
    buckets.zero()
    with pytest.raises(RuntimeError, match="received\\s+another one|already in its bucket slot"):
        loss().backward()
    buckets.finish()
    bf.set_kl_gradient(True)
    try:
        buckets.zero()
        assert buckets.slot(w) is None
        loss().backward()
        buckets.finish()
        np.testing.assert_allclose(w.grad.numpy(), ref.numpy(), rtol=1e-6)
    finally:
        bf.set_kl_gradient(False)
        buckets.remove()


def test_buckets_settle_on_the_parameters_that_receive_gradients():
    """requires_grad says too much: the Gaussian priors of a converted model are Parameters no gradient ever reaches (the
    reference's optimizer skips them because .grad stays None).  After the first step the buckets cover only what backward
    produced; the others keep .grad = None and are neither zero-filled, nor sent, nor stepped."""
    torch.manual_seed(3)
    w = torch.nn.Parameter(torch.randn(8, 4))
    prior = torch.nn.Parameter(torch.randn(8, 4))     # never in the graph
    b = torch.nn.Parameter(torch.randn(4))
    buckets = GradientBuckets([w, prior, b], bucket_bytes=1 << 20)
    assert buckets.flats()[0].numel() == 32 + 32 + 4
    for step in range(3):
        buckets.zero()
        (torch.ones(2, 8) @ w + b).sum().backward()
        buckets.finish()
        assert prior.grad is None
        assert w.grad.data_ptr() == buckets._views[w].data_ptr() and float(w.grad.sum()) == 2.0 * 32
        if step >= 1:
            assert {id(p) for p in buckets.params} == {id(w), id(b)}
            assert sum(f.numel() for f in buckets.flats()) == 32 + 4
    # a parameter that joins the graph later is reported, not silently left out of the reduction
    buckets.zero()
    with pytest.raises(RuntimeError, match="had no gradient in the first step"):
        (torch.ones(2, 8) @ (w + prior) + b).sum().backward()


def test_dropout_group_numbering_refuses_an_unsplittable_shard():
    """a shard that does not start at global sample 0 numbers its dropout groups from
    first_sample x (groups per sample); a tensor that cannot be cut into the shard's samples used to fall back to 0 —
    the masks of samples 0.., correlated across ranks — without a word."""
    from bayeformers_amd import _C
    from bayeformers_amd.ops import Dropout

    d0 = Dropout(0.1, 1, 0, 1, origin=(0, 3))
    assert d0.first_group(7, 96) == 0                     # the first shard numbers from 0 whatever the tensor looks like
    d = Dropout(0.1, 1, 0, 1, origin=(4, 2))              # global samples 4 and 5
    assert d.first_group(2 * 160, 96) == 4 * 160 * 96
    with pytest.raises(_C.BayeFormersAMDError, match="cannot be split"):
        d.first_group(2 * 160 + 1, 96)


def test_release_training_buffers_forgets_the_deferred_table():
    from bayeformers_amd import training

    class M:   # what release_training_buffers needs of a model
        pass

    m = M()
    mgr = m.__dict__["_pgrad"] = training.DeferredParamGrads()
    mgr.table, mgr.armed = {"sig": ()}, True
    training.release_training_buffers(m)
    assert mgr.table is None and not mgr.armed
    training.release_training_buffers(M())   # nothing to release: no error


def test_deferred_param_grads_decides_before_the_backward_pass():
    """training.DeferredParamGrads without a GPU: the observe / arm / fall-back logic and the table it builds (host side only —
    nothing is launched).  A step defers only when its forward showed exactly the layers, shapes and parameter addresses the table
    was built for, each layer once, with no gradient standing on its parameters."""
    import bayeformers_amd as bf
    from bayeformers_amd import training

    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8, bias=False))
    model = bf.to_bayesian(net)
    l0, l1 = model.fused_children()
    mgr = training.DeferredParamGrads()
    S, M, cdt = 3, 256, torch.bfloat16

    def forward(layers, m=M):
        mgr.begin(model)
        assert l0._bf_pg_defer is mgr and l1._bf_pg_defer is mgr
        for l in layers:
            mgr.note_forward(l, S, m, cdt)
        mgr.decide()

    forward([l0, l1])
    assert not mgr.armed and mgr.table is not None            # first step: observed, table built for the next one
    t = mgr.table
    assert t["n"] == 3 and len(t["layers"]) == 2               # two weights and one bias
    assert t["layers"][id(l1)][5] is None                      # no bias buffer for the bias-less layer
    assert t["layers"][id(l0)][4].shape[0] == S and tuple(t["layers"][id(l0)][4].shape[2:]) == (32, 16)
    assert {id(p) for p, _ in t["grads"]} == {id(l0.weight.rho), id(l0.weight.mu), id(l0.bias.rho), id(l0.bias.mu),
                                               id(l1.weight.rho), id(l1.weight.mu)}
    assert all(v.data_ptr() % 16 == 0 for _, v in t["grads"])  # every slot of the flat buffer on a 16-byte boundary
    forward([l0, l1])
    assert mgr.armed and mgr.table is t                         # same shapes: this step defers, on the same table
    assert mgr.keep_buffer(l0, S, M, cdt, 1, 0) is t["layers"][id(l0)][4] and mgr.grad_view(l0.weight.rho).shape == (32, 16)
    with pytest.raises(RuntimeError, match="does not match"):
        mgr.keep_buffer(l0, S, M + 1, cdt, 1, 0)                # a backward that contradicts the armed forward: never half deferred
    mgr.note_forward(l0, S, M, cdt)                             # (a block recomputed during backward: not observed any more)
    forward([l0, l1], m=512)
    assert not mgr.armed and mgr.table is not t                 # another batch shape: per-layer path now, re-armed for next time
    forward([l0, l0, l1], m=512)
    assert not mgr.armed and mgr.table is None                  # a layer used twice in one forward is never deferred
    forward([l0, l1], m=512)
    forward([l0, l1], m=512)
    assert mgr.armed
    l0.weight.rho.grad = torch.zeros_like(l0.weight.rho)        # a gradient already standing (accumulation): not this step
    forward([l0, l1], m=512)
    assert not mgr.armed
    l0.weight.rho.grad = None
    training.release_training_buffers(model)                     # (no manager on the model itself: nothing to do)
    mgr.release()
    assert mgr.table is None
