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
from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .nn.model import Model
from .sampling import elbo, sample_bayesian


class GradientBuckets:
    """Flat gradient buffers over the trainable parameters of a model, summed over the ranks of `group` while backward
    runs.  `zero()` before the forward, `finish()` after `backward()`; the parameters' `.grad` are views of the buffers
    (the optimizer updates from them directly)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional["dist.ProcessGroup"] = None,
                 bucket_bytes: int = 128 << 20):
        self.group = group
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        # backward produces gradients roughly in reverse order of use: buckets are filled in reverse registration order
        self.buckets = []   # [flat tensor, [params], pending count, launched]
        self._bucket_of = {}
        self._views = {}
        cur, cur_bytes, key = [], 0, None
        for p in reversed(self.params):
            k = (p.dtype, p.device)
            nb = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nb > bucket_bytes):
                self._close(cur)
                cur, cur_bytes = [], 0
            key = k
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._close(cur)
        self._works = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._arrived) for p in self.params]

    def _close(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        for p in ps:
            self._views[p] = flat[off:off + p.numel()].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += p.numel()
        self.buckets.append([flat, ps, len(ps), False])

    def zero(self) -> None:
        """Clear the buffers and (re-)attach the views as the parameters' gradients."""
        self._works = []
        for b in self.buckets:
            b[0].zero_()
            b[2], b[3] = len(b[1]), False
        for p in self.params:
            if p.grad is not self._views[p]:
                p.grad = self._views[p]

    def _launch(self, i: int) -> None:
        b = self.buckets[i]
        if b[3]:
            return
        b[3] = True
        if self.distributed:
            self._works.append(dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _arrived(self, p) -> None:
        if p.grad is not self._views[p]:  # autograd replaced the view (first gradient of a parameter whose grad was None)
            self._views[p].copy_(p.grad)
            p.grad = self._views[p]
        i = self._bucket_of[p]
        self.buckets[i][2] -= 1
        if self.buckets[i][2] == 0:
            self._launch(i)

    def finish(self) -> None:
        """After backward(): send what has not been sent (parameters without a gradient this step) and wait."""
        for i in range(len(self.buckets)):
            self._launch(i)
        for w in self._works:
            w.wait()
        self._works = []

    def flats(self) -> List[Tensor]:
        return [b[0] for b in self.buckets]

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def clip_grad_norm_(buckets: GradientBuckets, max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) on the flat buffers: the same total 2-norm and scaling
    (examples/bert_glue.py:240 clips at 1), in a few launches instead of one per parameter."""
    flats = buckets.flats()
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(f.float()) for f in flats]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for f in flats:
        f.mul_(coef.to(f.dtype))
    return total


def training_step(model: Model, inputs, samples: int, nll_fn: Callable, optimizer: torch.optim.Optimizer, n_batches: int,
                  buckets: Optional[GradientBuckets] = None, max_grad_norm: Optional[float] = 1.0,
                  group: Optional["dist.ProcessGroup"] = None, select: Optional[Callable] = None) -> Tensor:
    """One optimisation step as in examples/bert_glue.py:227-241, S-sharded over `group` when torch.distributed is up.

    nll_fn(mean_outputs) -> scalar negative log-likelihood of the MEAN outputs (a tuple, as sample_bayesian returns).
    Every rank ends the step with the same parameters.  Returns the (detached) ELBO loss."""
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
            clip_grad_norm_(buckets, max_grad_norm)
    elif max_grad_norm is not None:
        torch.nn.utils.clip_grad_norm_([p for g in optimizer.param_groups for p in g["params"]], max_grad_norm)
    optimizer.step()
    return loss.detach()
