"""The reference's training step on S-sharded ranks.

Reference: /root/reference/examples/bert_glue.py:227-241 — per batch: `sample_bayesian` (S forwards), the NLL of the
MEAN logits, `loss = (lvp - log_prior) / n_batches + nll`, `loss.backward()`, `clip_grad_norm_(parameters, 1)`,
`optimizer.step()`.  The reference runs it in one process; its only multi-GPU construct (DataParallel over the batch,
examples/bert_squad.py:245) loses the log-probs.

Here every rank of a torch.distributed group backpropagates ITS Monte-Carlo samples (`sampling.sample_bayesian` keeps
the local part of the all-reduced means in the autograd graph), so a rank's parameter gradients are the contribution of
its samples and their SUM over the ranks is the gradient of the single-process step — for any number of ranks, because
the epsilon of a sample depends on its global index only.  `GradientBuckets` all-reduces them (RCCL over xGMI on MI355X):
the gradients live as views of a few flat buffers; a buffer goes on the wire, asynchronously, as soon as backward has
produced its last gradient, so the collectives run under the remaining backward GEMMs.  Bucket size: xGMI is
point-to-point (7 links of ~153 GB/s per GPU) and a ring all-reduce is bound per link, so a few large messages
(default 128 MiB: BERT-base's 342 MB of d rho in three) beat DDP's 25 MiB default; the last bucket — the first
layers' gradients — is the only one that cannot hide under backward.
"""
import weakref
from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from . import random as bfr
from .nn.model import Model
from .sampling import elbo, sample_bayesian


_NO_DEFERRED = __import__("os").environ.get("BF_NO_DEFERRED_PGRAD") is not None  # developer A/B: per-layer weight reductions

_BUCKETS_OF: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()  # optimizer -> the buckets training_step built for it


class GradientBuckets:
    """Flat gradient buffers over the parameters of a model THAT RECEIVE A GRADIENT, summed over the ranks of `group` while
    backward runs.  `zero()` before the forward, `finish()` after `backward()`; afterwards the parameters' `.grad` are
    views of the buffers (the optimizer updates from them directly).

    Which parameters: `requires_grad` says too much — the Gaussian priors of a converted model are nn.Parameters like the
    reference's (/root/reference/bayeformers/nn/parameters/gaussian.py:51-52), and no gradient ever reaches them (the
    log-probs are detached, layers/linear.py:99-102); the reference's optimizer skips them because their .grad stays None.
    So the buckets are laid out after the FIRST step, over what that step's backward actually produced (the first step
    itself runs on buckets over everything; the rest keep .grad = None, exactly as without buckets).  Measured on BERT-base:
    with the priors inside, every step zero-filled 370 slots, all-reduced 1 GB instead of 0.39 GB and ran the fused AdamW
    over three times the elements — the whole +2.4 ms the bucket path used to cost (profiles/r4g_*).

    How a gradient reaches its slot: the Bayesian layers' own backward kernels (ops.linear_backward: all fp32 mu / rho
    gradients, 342 MB of a BERT-base step) find it through `param._bf_grad_sink` and WRITE THERE — no copy, autograd is
    handed None.  Any other parameter (LayerNorms, embeddings) comes through a tensor hook that copies it into the slot at
    once (autograd keeps adopting the produced tensor as .grad until finish() points .grad at the slot).  A bucket goes on
    the wire, asynchronously, when its last gradient has arrived, i.e. under the remaining backward GEMMs.  Buckets hold one
    dtype each; slots are packed, odd-sized parameters last, so that all but those start on a 256-byte boundary."""

    ALIGN = 256  # bytes

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional["dist.ProcessGroup"] = None,
                 bucket_bytes: int = 128 << 20):
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.candidates: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self._works = []
        self._arrived = set()   # parameters whose gradient is in its slot this step
        self.active = False     # between zero() and finish(): the slots are live destinations
        self._settled = False   # True once the layout covers exactly the parameters that receive gradients
        self._relayout = None   # parameter list to lay out at the next zero()
        self._hooks = {p: p.register_hook(self._make_hook(p)) for p in self.candidates}
        self._layout(self.candidates)

    # ------------------------------------------------------------------------------------------------ layout
    def _layout(self, params) -> None:
        for p in getattr(self, "params", []):
            if getattr(p, "_bf_grad_sink", None) is self:
                del p._bf_grad_sink
        self.params = list(params)
        self.buckets = []   # [flat tensor, [params], pending count, launched]
        self._bucket_of = {}
        self._views = {}
        # backward produces gradients roughly in reverse order of use: per dtype, buckets are filled in reverse
        # registration order, so the first bucket to go on the wire holds the last layers' gradients
        by_kind = {}
        for p in reversed(self.params):
            by_kind.setdefault((p.dtype, p.device), []).append(p)
        for ps in by_kind.values():
            cur, cur_bytes = [], 0
            for p in ps:
                nb = p.numel() * p.element_size()
                if cur and cur_bytes + nb > self.bucket_bytes:
                    self._close(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nb
            if cur:
                self._close(cur)
        for p in self.params:
            p._bf_grad_sink = self

    def _close(self, ps):
        # parameters whose size keeps the next one 256-byte aligned first, the odd-sized ones (a 2-element classifier bias)
        # last: the slots are packed without padding
        q = self.ALIGN // ps[0].element_size()
        ps = [p for p in ps if p.numel() % q == 0] + [p for p in ps if p.numel() % q != 0]
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        for p in ps:
            self._views[p] = flat[off:off + p.numel()].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += p.numel()
        self.buckets.append([flat, ps, len(ps), False])

    # ------------------------------------------------------------------------------------------------ one step
    def zero(self) -> None:
        """Start a step: gradients are None (autograd adopts the tensors backward produces, no accumulation kernels)."""
        if self._relayout is not None:
            self._layout(self._relayout)
            self._relayout, self._settled = None, True
        self._works = []
        self._arrived = set()
        self.active = True
        for b in self.buckets:
            b[2], b[3] = len(b[1]), False
        for p in self.candidates:
            p.grad = None

    def slot(self, p):
        """The standing destination of p's gradient while a step is open (None otherwise): ops.linear_backward writes it.
        Not with the opt-in KL gradient (set_kl_gradient): then a mu / rho receives TWO gradients per backward — the layer's
        and the KL term's — which autograd must sum before anything is sent, so the producer takes the ordinary path and
        the hook below copies the sum."""
        return self._views.get(p) if self.active and not bfr.STATE.kl_gradient else None

    def arrived(self, p) -> None:
        """p's gradient is in its slot (written there by the kernel that produced it, or copied by the hook)."""
        i = self._bucket_of[p]
        if self.buckets[i][3] or p in self._arrived:
            raise RuntimeError("GradientBuckets: a parameter received a second gradient in one backward() (a module "
                               "used twice?) — every trainable parameter must be used once per step")
        self._arrived.add(p)
        self.buckets[i][2] -= 1
        if self.buckets[i][2] == 0:
            self._launch(i)

    def _make_hook(self, p):
        def hook(grad):
            if grad is None or not self.active:
                return None  # (a gradient its kernel wrote in place reaches autograd as None)
            if p in self._arrived:
                # its slot is complete — possibly on the wire already: a further defined gradient can be neither added
                # nor dropped
                raise RuntimeError("GradientBuckets: a parameter whose gradient was already in its bucket slot received "
                                   "another one in the same backward(); every gradient of a parameter must reach it "
                                   "through one path per step")
            view = self._views.get(p)
            if view is None:
                raise RuntimeError("GradientBuckets: a parameter that had no gradient in the first step received one now; "
                                   "build new GradientBuckets for the changed model")
            view.copy_(grad)  # straight into the slot: no reference kept, autograd still adopts `grad` without a clone
            self.arrived(p)
            return None
        return hook

    def _launch(self, i: int) -> None:
        b = self.buckets[i]
        if b[3]:
            return
        b[3] = True
        for p in b[1]:
            if p not in self._arrived:  # no gradient this step: the slot must read zero on the wire
                self._views[p].zero_()
        if self.distributed:
            self._works.append(dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """After backward(): send what has not been sent (buckets holding a parameter without a gradient this step), wait,
        and point .grad at the slots.  After the first step the layout shrinks to the parameters that got a gradient."""
        for i in range(len(self.buckets)):
            self._launch(i)
        for w in self._works:
            w.wait()
        self._works = []
        self.active = False
        got = self._arrived
        for p in self.params:
            # the reduced gradient; a parameter no gradient reached keeps .grad = None (the optimizer skips it, as it
            # does without buckets) until the layout drops it
            p.grad = self._views[p] if (self._settled or p in got) else None
        if not self._settled:
            if len(got) < len(self.params):
                self._relayout = [p for p in self.params if p in got]
            else:
                self._settled = True
        self._arrived = set()

    def flats(self) -> List[Tensor]:
        return [b[0] for b in self.buckets]

    def remove(self) -> None:
        for h in self._hooks.values():
            h.remove()
        self._hooks = {}
        self.active = False
        for p in self.params:
            if getattr(p, "_bf_grad_sink", None) is self:
                del p._bf_grad_sink


class DeferredParamGrads:
    """One launch for the weight gradients of ALL Bayesian linear layers of a single-process training step.

    Every layer's backward (bf_linear_bwd) ends with a reduction over the Monte-Carlo samples — dmu = sum_s dW_s,
    drho = (sum_s dW_s o eps_s) o softplus'(rho), eps regenerated — that reads the layer's S (x split-K) fp32 gradients of
    the sampled weights: 73 launches of ~9 waves per CU in a BERT-base step (1.5 ms, 2.9 TB/s).  Here the layers leave their
    dW_s in standing buffers instead (`d_dw_keep`: 4.6 GB for BERT-base at S = 10, 0.016 of an MI355X's memory) and ONE
    table-driven launch over all of them (bf_param_grad_table: 83 k workgroups, every CU's loads in flight) runs after the
    backward pass; the gradients land in flat buffers of this object and become the parameters' `.grad`, like
    GradientBuckets' slots.  Same arithmetic per element, in the same order: the gradients are those of the per-layer path
    bit for bit.

    `training_step` uses it when nothing else has a claim on how the gradients travel: one process, no gradient buckets,
    no opt-in KL gradient.  The first step of a model (or of a new batch shape) runs the per-layer path and only OBSERVES
    (layer, samples, rows, dtype) of every layer's forward; the buffers and the table are built from that, and a step
    defers when its forward showed exactly the shapes the table was built for — decided before the backward pass starts,
    so a step is never half deferred.  The biases ride in the same launch (73 more launches of 6 us gone): their per-sample
    gradients — the column sums of dy, [S][N] — are written into standing buffers by whichever kernel forms them; the sums a
    producer of dy left with it (attention backward: query / key / value) are copied there by one multi-tensor copy."""

    def __deepcopy__(self, memo):
        return DeferredParamGrads()

    def __reduce__(self):
        return (DeferredParamGrads, ())

    def release(self) -> None:
        """Give the standing buffers back (4.6 GB for BERT-base at S = 10): the next training step observes and builds again.
        `release_training_buffers(model)` calls it."""
        self.table, self.armed, self._last, self._copies = None, False, None, ([], [])

    def __init__(self):
        self.table = None      # {id(layer): (layer, S, M, cdt, dw buffer)} + device blob, built from an observed step
        self.armed = False     # the running step's backward defers
        self._seen = []        # (layer, S, M, cdt) of the running step's forwards
        self._open = False     # between begin() and decide(): forwards are being observed
        self._last = None      # (seed, sample_base, counter snapshot, S) of the backward calls of the running step
        self._copies = ([], [])  # (sources, destinations): column sums their producer left elsewhere, gathered before the launch

    # -- called by training_step ---------------------------------------------------------------------------------
    def begin(self, model) -> None:
        from .nn.layers.linear import Linear

        for l in model.fused_children():
            if isinstance(l, Linear) and l.__dict__.get("_bf_pg_defer") is not self:
                l._bf_pg_defer = self
        self._seen, self._open, self.armed, self._last, self._copies = [], True, False, None, ([], [])

    def decide(self) -> None:
        """After the step's forward, before its backward."""
        self._open = False
        seen = self._seen
        ids = [id(l) for l, *_ in seen]
        usable = bool(seen) and len(set(ids)) == len(ids) and not bfr.STATE.kl_gradient
        if usable and self.table is not None and self._signature(seen) == self.table["sig"]:
            self.armed = all(p.grad is None for p, _ in self.table["grads"])
            return
        self.armed = False
        self.table = self._build(seen) if usable else None

    def finish(self) -> None:
        """After the step's backward: the one reduction launch, and the gradients handed to the parameters."""
        if not self.armed or self._last is None:
            self.armed = False
            return
        from . import _C, ops

        seed, base, counter, S = self._last
        t = self.table
        if self._copies[0]:
            torch._foreach_copy_(self._copies[1], self._copies[0])
        self._copies = ([], [])
        with bfr.counter_override(counter):
            _C.check(_C.lib().bf_param_grad_table(t["blob"].data_ptr(), t["n"], t["blocks"], S, seed, base & 0xFFFFFFFF,
                                                  ops._stream_ptr()), "bf_param_grad_table")
        for p, view in t["grads"]:
            p.grad = view
        self.armed, self._last = False, None

    # -- called by the layers ------------------------------------------------------------------------------------
    def note_forward(self, layer, S, M, cdt) -> None:
        if self._open:
            self._seen.append((layer, int(S), int(M), cdt))

    def keep_buffer(self, layer, S, M, cdt, seed, sample_base):
        if not self.armed:
            return None
        e = self.table["layers"].get(id(layer))
        if e is None or e[1:4] != (S, M, cdt):  # cannot happen after decide(); a deferred step must not be left half done
            raise RuntimeError("DeferredParamGrads: a layer's backward does not match the forward this step was armed for")
        self._last = (seed, sample_base, getattr(bfr.STATE, "override_snapshot", None), S)
        return e[4]

    def keep_bias_buffer(self, layer, given_colsum):
        """The standing [S][N] buffer the layer's bias column sums go to; `given_colsum`: they exist already (left by the kernel
        that produced dy) — gathered into the buffer before the launch."""
        db = self.table["layers"][id(layer)][5]
        if db is not None and given_colsum is not None:
            self._copies[0].append(given_colsum)
            self._copies[1].append(db)
        return db

    def grad_view(self, p):
        return self.table["view"][id(p)]

    # -- internals -----------------------------------------------------------------------------------------------
    @staticmethod
    def _signature(seen):
        from .nn.parameters.gaussian import Gaussian

        return tuple((id(l), S, M, cdt) + tuple(v for g in (l.weight, l.bias) if isinstance(g, Gaussian)
                                                for v in (g.mu.data_ptr(), g.rho.data_ptr(), g.mu.requires_grad))
                     for l, S, M, cdt in seen)

    def _build(self, seen):
        import ctypes

        from . import _C, ops

        lib = _C.lib()
        dev = seen[0][0].weight.rho.device
        from .nn.parameters.gaussian import Gaussian

        def tensors_of(l):
            return [l.weight] + ([l.bias] if isinstance(l.bias, Gaussian) else [])

        def aligned(n):  # every tensor's slots start on a 16-byte boundary of the flat buffer
            return (n + 3) // 4 * 4

        n_el = sum(aligned(g.rho.numel()) * (2 if g.mu.requires_grad else 1) for l, *_ in seen for g in tensors_of(l))
        flat = torch.zeros(n_el, dtype=torch.float32, device=dev)
        n_entries = sum(len(tensors_of(l)) for l, *_ in seen)
        arr = (_C.bf_pgrad_t * n_entries)()
        layers, view, grads, off, i = {}, {}, [], 0, 0
        for l, S, M, cdt in seen:
            N, K = l.out_features, l.in_features
            sp = lib.bf_linear_bwd_splits(S, M, N, K, ops._TORCH2BF[cdt])
            dw = torch.empty((S, sp, N, K), dtype=torch.float32, device=dev)
            db = torch.empty((S, N), dtype=torch.float32, device=dev) if isinstance(l.bias, Gaussian) else None
            layers[id(l)] = (l, S, M, cdt, dw, db)
            for k, (g, buf, splits) in enumerate(zip(tensors_of(l), (dw, db), (sp, 1))):
                rho, mu = g.rho, g.mu
                view[id(rho)] = flat[off:off + rho.numel()].view_as(rho)
                off += aligned(rho.numel())
                grads.append((rho, view[id(rho)]))
                if mu.requires_grad:
                    view[id(mu)] = flat[off:off + mu.numel()].view_as(mu)
                    off += aligned(mu.numel())
                    grads.append((mu, view[id(mu)]))
                e = arr[i]
                e.d_dw, e.d_rho, e.d_drho = buf.data_ptr(), rho.data_ptr(), view[id(rho)].data_ptr()
                e.d_dmu = view[id(mu)].data_ptr() if mu.requires_grad else None
                e.n, e.stream_id, e.splits = rho.numel(), 2 * l.layer_id + k, splits
                i += 1
        blocks = ctypes.c_uint32()
        nbytes = lib.bf_param_grad_table_bytes(arr, n_entries, ctypes.byref(blocks))
        blob = torch.empty(nbytes, dtype=torch.uint8)
        _C.check(lib.bf_param_grad_table_build(arr, n_entries, blob.data_ptr(), nbytes), "bf_param_grad_table_build")
        return {"sig": self._signature(seen), "layers": layers, "view": view, "grads": grads, "flat": flat,
                "blob": blob.to(dev), "n": n_entries, "blocks": blocks.value}


def release_training_buffers(model: Model) -> None:
    """Free what `training_step` keeps on a model between steps (the per-sample weight-gradient buffers of DeferredParamGrads) —
    for a long evaluation phase after training on the same GPU."""
    mgr = model.__dict__.get("_pgrad")
    if mgr is not None:
        mgr.release()


def grad_norm(tensors: List[Tensor]) -> Tensor:
    """2-norm of all `tensors` together (0-d fp32), as torch.nn.utils.clip_grad_norm_ computes it, in one multi-tensor
    launch per dtype instead of one reduction per tensor."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    # (one stack + one cast per dtype: a .float() per 0-d norm is a launch per gradient tensor)
    norms = [torch.stack(torch._foreach_norm(ts)).float() for ts in by_dtype.values()]
    return torch.linalg.vector_norm(torch.cat(norms) if len(norms) > 1 else norms[0])


def clip_gradients(optimizer: torch.optim.Optimizer, tensors: List[Tensor], max_norm: float) -> Tensor:
    """clip_grad_norm_(parameters, max_norm) (examples/bert_glue.py:240) for the step that follows: gradients are scaled
    by min(1, max_norm / (|g| + 1e-6)).  A fused torch optimizer applies the factor inside its update kernel (its
    `grad_scale` input, the hook torch.amp's GradScaler uses: gradients are divided by it) — no extra pass over the
    gradients; any other optimizer gets them scaled in place.  Returns the total norm."""
    total = grad_norm(tensors)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    if all(g.get("fused") for g in optimizer.param_groups):
        optimizer.grad_scale = (1.0 / coef).to(torch.float32)
        optimizer.found_inf = torch.zeros((), dtype=torch.float32, device=total.device)
    else:
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dt, ts in by_dtype.items():
            torch._foreach_mul_(ts, coef.to(dt))
    return total


def training_step(model: Model, inputs, samples: int, nll_fn: Callable, optimizer: torch.optim.Optimizer, n_batches: int,
                  buckets: Optional[GradientBuckets] = None, max_grad_norm: Optional[float] = 1.0,
                  group: Optional["dist.ProcessGroup"] = None, select: Optional[Callable] = None) -> Tensor:
    """One optimisation step as in examples/bert_glue.py:227-241, S-sharded over `group` when torch.distributed is up.

    nll_fn(mean_outputs) -> scalar negative log-likelihood of the MEAN outputs (a tuple, as sample_bayesian returns).
    Every rank ends the step with the same parameters.  Returns the (detached) ELBO loss."""
    if buckets is None:
        # S-sharded ranks each hold the gradient of THEIR samples: without a reduction they would step apart.  The
        # buckets of an optimizer are built once and kept with it.
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if sharded:
            buckets = _BUCKETS_OF.get(optimizer)
            if buckets is None or buckets.group is not group:
                if buckets is not None:
                    buckets.remove()
                buckets = _BUCKETS_OF[optimizer] = GradientBuckets(
                    [p for g in optimizer.param_groups for p in g["params"]], group=group)
    if buckets is not None:
        buckets.zero()
    else:
        optimizer.zero_grad(set_to_none=True)
    # one process, no buckets: the weight gradients of all Bayesian linears are reduced by ONE launch after the backward pass
    deferred = None
    if buckets is None and not _NO_DEFERRED and not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        deferred = model.__dict__.get("_pgrad")
        if deferred is None:
            deferred = model.__dict__["_pgrad"] = DeferredParamGrads()
        deferred.begin(model)
    _, mean, lp, lq = sample_bayesian(model, inputs, samples, select=select, group=group)
    loss = elbo(lp, lq, nll_fn(mean).double(), n_batches)
    if deferred is not None:
        deferred.decide()
    loss.backward()
    if deferred is not None:
        deferred.finish()
    if buckets is not None:
        buckets.finish()
    if max_grad_norm is not None:
        grads = buckets.flats() if buckets is not None else [p.grad for g in optimizer.param_groups for p in g["params"]
                                                             if p.grad is not None]
        clip_gradients(optimizer, grads, max_grad_norm)
    try:
        optimizer.step()
    finally:
        # the clipping factor belongs to THIS step (torch.amp's GradScaler removes the two attributes the same way): a
        # later step without clipping must not divide its gradients by a stale one
        for name in ("grad_scale", "found_inf"):
            if hasattr(optimizer, name):
                delattr(optimizer, name)
    return loss.detach()


class GraphedTrainingStep:
    """`training_step` for ONE batch signature on one process, replayed from a HIP graph: forward of the S samples, ELBO,
    backward, gradient clipping and the optimizer update are captured once and replayed — the 681 launches of a BERT-base
    step cost the host one `graph.replay()` instead of 19-22 ms of Python and launch calls (profiles/r5s_*), which is what
    bounds the step once a rank's shard is one or two samples.

        opt = torch.optim.AdamW(params, lr=2e-5, fused=True, capturable=True)
        step = GraphedTrainingStep(bmodel.train(), inputs, samples=10, nll_fn=nll, optimizer=opt, n_batches=2105)
        for batch in loader:
            loss = step(batch)          # same shapes / dtypes: copied into the captured buffers

    Step k is the k-th eager `training_step` bit for bit: the Monte-Carlo sample counter AND the dropout call counter live in
    device memory while the object exists (`random.use_device_counter`), the captured step copies and advances both, so every
    replay draws fresh epsilon and fresh masks; torch's own dropout (the embedding block) advances through its graph-safe
    generator.  The first `eager_steps` calls run the ordinary eager step (they ARE training steps: optimizer state, sampling
    plan, workspaces and tile schedules come into being there); the next call captures and replays.
    Needs: one process (S-sharded ranks keep the eager step: their bucketed gradient all-reduce runs under backward), an
    optimizer built with `capturable=True`; its learning rate is turned into a device tensor (a scheduler's `fill_` is seen by
    the replays).  What a capture bakes in besides the shapes — seed, compute dtype, sampling plan, parameter addresses — is
    re-checked before every replay; a difference captures again (graphs.still_valid).  The returned loss is the graph's buffer."""

    def __init__(self, model: Model, inputs, samples: int, nll_fn: Callable, optimizer: torch.optim.Optimizer, n_batches: int,
                 max_grad_norm: Optional[float] = 1.0, select: Optional[Callable] = None, eager_steps: int = 2) -> None:
        from . import graphs
        from .sampling import GraphedSampler

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedTrainingStep: S-sharded ranks run the eager training_step (bucketed all-reduce under backward)")
        if not all(g.get("capturable") for g in optimizer.param_groups):
            raise RuntimeError("GraphedTrainingStep: build the optimizer with capturable=True (its step count must live on the GPU)")
        leaves = GraphedSampler._leaves(inputs)
        tensors = [v for v in leaves if isinstance(v, Tensor)]
        if not tensors or not all(t.is_cuda for t in tensors):
            raise RuntimeError("GraphedTrainingStep: the inputs must be tensors on the GPU the model runs on")
        self.model, self.samples, self.nll_fn, self.optimizer = model, int(samples), nll_fn, optimizer
        self.n_batches, self.max_grad_norm, self.select = n_batches, max_grad_norm, select
        self.device = tensors[0].device
        self._signature = GraphedSampler._sig(inputs)
        self._inputs = GraphedSampler._map(inputs, lambda v: v.clone())
        for g in optimizer.param_groups:  # a python float would be baked into the captured update
            if not isinstance(g["lr"], Tensor):
                g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=self.device)
        self.graph = self._loss = None
        self.steps = self.captures = 0
        self._eager_steps = max(1, int(eager_steps))
        graphs.acquire_counter(self.device)
        self._open = True

    def load(self, inputs) -> None:
        from .sampling import GraphedSampler

        if GraphedSampler._sig(inputs) != self._signature:
            raise ValueError("GraphedTrainingStep: the batch differs from the captured one in structure, shape, dtype or device")
        for dst, src in zip(GraphedSampler._leaves(self._inputs), GraphedSampler._leaves(inputs)):
            if isinstance(src, Tensor):
                dst.copy_(src)

    def _eager(self) -> Tensor:
        return training_step(self.model, self._inputs, self.samples, self.nll_fn, self.optimizer, self.n_batches,
                             max_grad_norm=self.max_grad_norm, select=self.select)

    def _capture(self) -> None:
        from . import graphs, sampling

        self.graph = self._loss = None
        sampling._REPEAT_CACHE.clear()  # the S-fold repeat of the inputs must be a launch OF the graph, not a kept tensor
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss = self._eager()
        sampling._REPEAT_CACHE.clear()  # (what the capture put there lives in the graph's memory pool)
        self.graph, self._loss = graph, loss
        self._baked = graphs.baked_state(self.model)
        self.captures += 1

    def __call__(self, inputs=None) -> Tensor:
        from . import graphs

        if not self._open:
            raise RuntimeError("GraphedTrainingStep: closed")
        if inputs is not None:
            self.load(inputs)
        if self.graph is None and self.steps < self._eager_steps:
            self.steps += 1
            return self._eager()
        if self.graph is None or not graphs.still_valid(self.model, self._baked):
            self._capture()   # (a capture executes nothing: the replay below is this call's step)
        self.graph.replay()
        self.steps += 1
        return self._loss

    def close(self) -> None:
        from . import graphs

        self.graph = self._loss = None
        if self._open:
            self._open = False
            graphs.release_counter()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
