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

from .nn.model import Model
from .sampling import elbo, sample_bayesian


_BUCKETS_OF: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()  # optimizer -> the buckets training_step built for it


class GradientBuckets:
    """Flat gradient buffers over the trainable parameters of a model, summed over the ranks of `group` while backward
    runs.  `zero()` before the forward, `finish()` after `backward()`; the parameters' `.grad` are views of the buffers
    (the optimizer updates from them directly).  Buckets hold one dtype each (the fp32 mu / rho masters apart from a
    16-bit model's embeddings and LayerNorms); the slots are packed, odd-sized parameters last, so that all but those
    start on a 256-byte boundary and the multi-tensor optimizer kernels keep their vectorised path.  What it costs: one
    torch.cat per bucket (the gradients backward produced -> their slots), i.e. one extra read and write of the gradients per
    step — use it where there is something to all-reduce (world > 1)."""

    ALIGN = 256  # bytes

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional["dist.ProcessGroup"] = None,
                 bucket_bytes: int = 128 << 20):
        self.group = group
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
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
                if cur and cur_bytes + nb > bucket_bytes:
                    self._close(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nb
            if cur:
                self._close(cur)
        self._works = []
        # Two ways a gradient reaches its slot.  The Bayesian layers' own backward kernels (ops.linear_backward: all of the
        # fp32 mu / rho gradients, 342 MB of a BERT-base step) find the slot through `param._bf_grad_sink` and WRITE THERE:
        # no copy, autograd sees None.  Every other parameter (LayerNorms, embeddings) comes through a tensor hook (which
        # sees the gradient before it is accumulated) and is copied into its slot when its bucket closes.
        self._hooks = [p.register_hook(self._make_hook(p)) for p in self.params]
        self._got = {}
        self._direct = set()   # parameters whose gradient was written straight into the slot this step
        self.active = False    # between zero() and finish(): the slots are live destinations
        for p in self.params:
            p._bf_grad_sink = self

    def _close(self, ps):
        # parameters whose size keeps the next one 256-byte aligned first, the odd-sized ones (a 2-element classifier bias)
        # last: the slots are packed without padding, so that ONE torch.cat fills a bucket
        q = self.ALIGN // ps[0].element_size()
        ps = [p for p in ps if p.numel() % q == 0] + [p for p in ps if p.numel() % q != 0]
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        for p in ps:
            self._views[p] = flat[off:off + p.numel()].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += p.numel()
        self.buckets.append([flat, ps, len(ps), False])

    def slot(self, p):
        """The standing destination of p's gradient while a step is open (None otherwise): ops.linear_backward writes it."""
        return self._views.get(p) if self.active else None

    def arrived(self, p) -> None:
        """p's gradient has been written into its slot by the kernel that produced it."""
        i = self._bucket_of[p]
        if self.buckets[i][3] or p in self._got or p in self._direct:
            raise RuntimeError("GradientBuckets: a parameter received a second gradient in one backward() (a module "
                               "used twice?) — every trainable parameter must be used once per step")
        self._direct.add(p)
        self.buckets[i][2] -= 1
        if self.buckets[i][2] == 0:
            self._launch(i)

    def zero(self) -> None:
        """Start a step: gradients are None (autograd adopts the tensors backward produces, no accumulation kernels);
        a bucket goes on the wire when its last gradient has arrived."""
        self._works = []
        self._got = {}
        self._direct = set()
        self.active = True
        for b in self.buckets:
            b[2], b[3] = len(b[1]), False
        for p in self.params:
            p.grad = None

    def _launch(self, i: int) -> None:
        b = self.buckets[i]
        if b[3]:
            return
        b[3] = True
        have = [p for p in b[1] if p in self._got]
        direct = [p for p in b[1] if p in self._direct]
        if len(direct) == len(b[1]):
            pass  # every gradient of the bucket was written in place by its kernel: nothing to copy
        elif len(have) == len(b[1]) and all(self._got[p].dtype == b[0].dtype for p in have):
            # one batched copy of the bucket's gradients into its flat buffer
            torch.cat([self._got[p].reshape(-1) for p in b[1]], out=b[0])
        else:
            missing = [p for p in b[1] if p not in self._got and p not in self._direct]
            for p in missing:  # no gradient this step: the slot must read zero
                self._views[p].zero_()
            if have:
                torch._foreach_copy_([self._views[p] for p in have], [self._got[p] for p in have])
        if self.distributed:
            self._works.append(dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, p):
        def hook(grad):
            if grad is None or p in self._direct:
                return None  # written in place by its kernel (ops.linear_backward): autograd was handed no gradient
            i = self._bucket_of[p]
            if self.buckets[i][3] or p in self._got:
                raise RuntimeError("GradientBuckets: a parameter received a second gradient in one backward() (a module "
                                   "used twice?) — every trainable parameter must be used once per step")
            self._got[p] = grad
            self.buckets[i][2] -= 1
            if self.buckets[i][2] == 0:
                self._launch(i)
            return None
        return hook

    def finish(self) -> None:
        """After backward(): send what has not been sent (parameters without a gradient this step) and wait."""
        for i in range(len(self.buckets)):
            self._launch(i)
        for w in self._works:
            w.wait()
        self._works = []
        self._got = {}
        self._direct = set()
        self.active = False
        for p in self.params:  # the optimizer (and the clipping) read the reduced gradients from the flat buffers
            p.grad = self._views[p]

    def flats(self) -> List[Tensor]:
        return [b[0] for b in self.buckets]

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.active = False
        for p in self.params:
            if getattr(p, "_bf_grad_sink", None) is self:
                del p._bf_grad_sink


def grad_norm(tensors: List[Tensor]) -> Tensor:
    """2-norm of all `tensors` together (0-d fp32), as torch.nn.utils.clip_grad_norm_ computes it, in one multi-tensor
    launch per dtype instead of one reduction per tensor."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    norms = [n.float() for ts in by_dtype.values() for n in torch._foreach_norm(ts)]
    return torch.linalg.vector_norm(torch.stack(norms))


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
    _, mean, lp, lq = sample_bayesian(model, inputs, samples, select=select, group=group)
    loss = elbo(lp, lq, nll_fn(mean).double(), n_batches)
    loss.backward()
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
