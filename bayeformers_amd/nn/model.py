# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.model

Wrapper that turns any module containing Bayesian children into a Bayesian model — same surface as
/root/reference/bayeformers/nn/model.py (is_module_bayesian :16-28, Model :31-89).

Additions for the MI355X path (all optional; with S = 1 the behaviour is the reference's):
  * `with model.monte_carlo(S): out = model(**inputs_repeated_S_times)` runs S Monte-Carlo samples in ONE forward:
    every bnn.Linear sees [S*B, ...] (sample-major), multiplies slab s by W_s and leaves per-sample log-probs;
  * the per-layer {log_prior, log_q} land in one [L, S, 2] float64 buffer, so `log_prior()` is one reduction
    instead of 74 scalar adds; `log_prior_samples()` / `log_variational_posterior_samples()` return the [S] values;
  * all layers of one forward share the same reserved Monte-Carlo sample indices (bayeformers_amd.random).
"""
import contextlib
import itertools
import warnings
from typing import Any, Iterator, List, Optional

import torch
from torch import Tensor
from torch.nn import Module

from .. import ops
from .. import random as bfr
from ..graphs import GraphCache, auto_forward
from ..plan import SamplePlan
from .layers.base import KernelLayer
from .layers.linear import Linear


def is_module_bayesian(module: Module) -> bool:
    """A module is Bayesian if it exposes log_prior and log_variational_posterior (model.py:16-28)."""
    log_prior = hasattr(module, "log_prior")
    log_variational_posterior = hasattr(module, "log_variational_posterior")
    return log_prior and log_variational_posterior


_TOKENS = itertools.count(1)


class _ForwardContext:
    """What the Bayesian layers of one Model.forward share: the reserved sample indices, their log-prob slots and
    (when every layer is plannable) the cross-layer sampling plan."""

    def __init__(self, sample_base: int, S: int, slots: dict, plan=None, lp_buf=None, shard_start: int = 0, dropping: bool = True,
                 drop_sites=None):
        self.sample_base, self.S, self._slots = sample_base, S, slots
        self.plan, self.lp_buf = plan, lp_buf
        self.token = next(_TOKENS)
        self.shared_out = {}  # id(first layer of a stacked run) -> (input identity, [L, S, M, N] outputs)
        # device-counter mode: the counter value this forward's kernels added — what `replay` needs, i.e. only a forward
        # that records gradients (the copy is a kernel launch)
        self.counter = bfr.counter_snapshot(torch.is_grad_enabled())
        if self.counter is not None:  # ... and the layers' autograd nodes share it (counter_snapshot's per-forward cache)
            self._counter_snap = (bfr.STATE.device_counter, bfr.STATE.counter_moves, self.counter)
        self.graph_tasks = set()  # ids of the backward passes that reached this forward's outputs (bfr.remember_context)
        # the `call` of every dropout applied inside this forward; `shard_start` (the first global sample of this process's
        # shard within the step) is where the kernels start numbering their dropout groups (ops.Dropout.first_group): a
        # sample's masks are a function of its GLOBAL index, so S-sharded training draws the single-process masks
        self.drop_call = bfr.reserve_dropout_call()
        self.drop_counter = bfr.reserve_dropout_counter(needed=dropping)  # device-counter mode: this forward's copy of it
        self.shard_start = int(shard_start)
        self.drop_sites = drop_sites  # the owning model's [next free dropout site number] (random.dropout_site)

    @contextlib.contextmanager
    def replay(self):
        """Run layer forwards again under this (finished) forward's sample indices: the recomputation of a checkpointed
        block during backward (bayeformers_amd.random.recompute_context).  Sampled weights still resident in the plan's
        arenas are reused; overwritten ones are drawn again from the same counters."""
        prev = bfr.STATE.ctx
        bfr.STATE.ctx = self
        self._replaying = True
        try:
            with bfr.counter_override(self.counter):
                yield self
        finally:
            self._replaying = False
            bfr.STATE.ctx = prev
            if self.plan is not None and prev is None:
                self.plan.pending.clear()  # the log-probs of these samples were reduced when the forward ended

    def slot(self, layer) -> Optional[Tensor]:
        return self._slots.get(id(layer))


class _KLFn(torch.autograd.Function):
    """[S, 2] per-sample {log_prior, log_q} of the model as a differentiable function of every layer's mu and rho
    (opt-in, bayeformers_amd.set_kl_gradient).  Backward = one bf_kl_grad launch per parameter tensor."""

    @staticmethod
    def forward(ctx, lp, model, S, seed, base, counter, *params):
        ctx.model, ctx.S, ctx.seed, ctx.base, ctx.counter = model, S, seed, base, counter
        return lp.clone()

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        from .parameters.gaussian import Gaussian

        g = g.to(torch.float64).contiguous()
        grads = []
        with bfr.counter_override(ctx.counter):
            for layer in ctx.model.fused_children():
                pairs = [(layer.weight, layer.weight_prior, 2 * layer.layer_id)]
                if isinstance(layer.bias, Gaussian):
                    pairs.append((layer.bias, layer.bias_prior, 2 * layer.layer_id + 1))
                for gauss, prior, sid in pairs:
                    dmu, drho = ops.kl_grad(gauss, prior, sid, ctx.S, ctx.seed, ctx.base, g, gauss.mu.requires_grad)
                    grads += [dmu, drho if gauss.rho.requires_grad else None]
        return (None, None, None, None, None, None) + tuple(grads)


class Model(Module):
    """Wrapper gathering log_prior and log_variational_posterior from Bayesian children.

    Attributes:
        model (Optional[nn.Module]): wrapped module (None when subclassed with its own forward)
    """

    def __init__(self, model: Optional[Module] = None) -> None:
        super(Model, self).__init__()
        self.model = model
        self._mc_samples = 1
        self._mc_span = (0, 1)  # (first global sample index of this process, indices every step consumes)
        self._fused: Optional[List[KernelLayer]] = None
        self._lp_buf: Optional[Tensor] = None
        self._last_base = None
        self._plan: Optional[SamplePlan] = None
        self.cross_layer_sampling = True  # one sampling launch per ~96 MB group of layers instead of one per layer
        # evaluation forwards (eval mode, no_grad, one sample per call: the reference's own caller loop,
        # /root/reference/examples/bert_glue.py:63-66) are replayed from a HIP graph from their third call on (graphs.py)
        self.graph_replay = True
        self._graphs = GraphCache()

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, *args, **kwargs) -> Any:
        """Forward of the wrapped module (model.py:53-57)."""
        if self.model is not None:
            return self.model.forward(*args, **kwargs)
        raise NotImplementedError("Forward pass not implemented yet")

    def __call__(self, *args, **kwargs):
        if bfr.STATE.ctx is not None:  # nested bnn.Model: the outer forward owns the sample indices
            return super(Model, self).__call__(*args, **kwargs)
        out = auto_forward(self, args, kwargs)  # an evaluation forward seen before: replayed from its HIP graph
        if out is not None:
            return out
        return self._eager_call(*args, **kwargs)

    def train(self, mode: bool = True):
        cache = self.__dict__.get("_graphs")
        if cache is not None:
            cache.refused.clear()  # signatures that were kept eager because of a child in training mode get another look
        return super(Model, self).train(mode)

    def _eager_call(self, *args, **kwargs):
        S = self._mc_samples
        layers = self.fused_children()
        if layers:
            # a kernel of an earlier forward may have found a prior's baked constants different from the tensors they were read
            # from (an in-place edit through .data: no version counter moved).  That forward's log-prior is NaN; from this
            # one on every cached copy is dropped and the priors are described again.
            ops.refresh_stale_epoch()
        ops._COLSUM_OFFERS.clear()  # (column sums a backward pass offered and nobody took)
        slots = {}
        if layers:
            dev = layers[0].weight.mu.device
            buf = self._lp_buf
            if buf is None or buf.shape[0] != len(layers) or buf.shape[1] != S or buf.device != dev:
                with torch.inference_mode(False):  # (kept across forwards, written in place by forwards of any mode)
                    buf = self._lp_buf = torch.zeros((len(layers), S, 2), dtype=torch.float64, device=dev)
            slots = {id(l): buf[i] for i, l in enumerate(layers)}
        plan = None
        linears = [l for l in layers if isinstance(l, Linear)]
        if linears and self.cross_layer_sampling and SamplePlan.plannable(linears):
            # layers seen with <= 64 rows per sample run the single fused kernel instead (Linear.forward marks them)
            planned = [(i, l) for i, l in enumerate(layers) if isinstance(l, Linear) and not l._small_m]
            if planned:
                cdt = bfr.get_compute_dtype()
                pl = [l for _, l in planned]
                shared = tuple(sorted({id(l._shared_input): l._shared_input for l in pl
                                       if l._shared_input is not None}.values(), key=lambda t: t[0].layer_id))
                key = SamplePlan.make_key(pl, S, cdt, shared)
                if (self._plan is None or self._plan.key != key or not self._plan.alias_valid()
                        or self._plan.epoch != bfr.STATE.stale_epoch):
                    self._plan = SamplePlan(pl, S, cdt, layers[0].weight.mu.device, index=[i for i, _ in planned],
                                            shared=shared)
                plan = self._plan
        if plan is None:
            self._plan = None  # (a plan of an earlier forward that no layer would use now: its arenas are released)
        # every rank reserves the GLOBAL sample indices of the step and runs its own contiguous slice of them
        start, total = self._mc_span
        base = bfr.reserve_samples(total) + start
        self._last_base, self._last_seed, self._last_S = base, bfr.STATE.seed, S
        self._last_counter = bfr.counter_snapshot() if bfr.STATE.kl_gradient else None
        ctx = bfr.STATE.ctx = _ForwardContext(base, S, slots, plan, self._lp_buf, shard_start=start, dropping=self.training,
                                              drop_sites=self.__dict__.setdefault("_drop_sites", [1]))
        out = None
        try:
            out = super(Model, self).__call__(*args, **kwargs)
            return out
        finally:
            if plan is not None:
                plan.finish(self._lp_buf)  # one log-prob reduction for all the groups this forward sampled
            bfr.STATE.ctx = None
            # what a finished forward keeps is what a checkpointed block's recomputation needs (sample indices, slots,
            # plan, counter): NOT the stacked query/key/value outputs and their inputs — a replay re-keys on data_ptr and
            # computes them again anyway, and holding them would pin ~3 GB of a BERT-base step until the next forward
            ctx.shared_out = {}
            if torch.is_grad_enabled() and out is not None:
                bfr.remember_context(ctx, out)
            bfr.commit_samples(total)

    @contextlib.contextmanager
    def monte_carlo(self, samples: int, shard=(0, 1), span=None):
        """Run forwards with `samples` Monte-Carlo samples folded into the batch axis (inputs repeated S times,
        sample-major: `x.repeat(S, 1, ...)`).  shard = (rank, world): this process runs `samples` of the
        `samples * world` global sample indices of each step (S-sharding over GPUs, sampling.sample_bayesian).
        span = (start, total) says it directly — this process runs the global indices [start, start + samples) of the
        `total` every step consumes — for shards of unequal size (S = 10 over 8 GPUs: sampling.shard_span)."""
        prev = (self._mc_samples, self._mc_span)
        S = int(samples)
        if span is None:
            span = (int(shard[0]) * S, S * int(shard[1]))
        start, total = int(span[0]), int(span[1])
        if S < 1 or start < 0 or start + S > total:
            raise ValueError(f"monte_carlo: samples={S} at offset {start} do not fit in the step's {total} sample indices")
        self._mc_samples, self._mc_span = S, (start, total)
        self._mc_harness = getattr(self, "_mc_harness", 0) + 1  # (inside a harness: no transparent graph replay, graphs.py)
        try:
            yield self
        finally:
            self._mc_samples, self._mc_span = prev
            self._mc_harness -= 1

    def fused_children(self) -> List[KernelLayer]:
        """The kernel-backed children (bnn.Linear, bnn.Embedding) in registration order; their layer_id (Philox
        stream) is that order."""
        if self._fused is None:
            self._fused = [m for m in self.modules() if isinstance(m, KernelLayer)]
            for i, l in enumerate(self._fused):
                l.layer_id = i
        return self._fused

    def refresh(self) -> None:
        """Re-scan the children after the module tree was edited."""
        self._fused, self._lp_buf, self._plan = None, None, None
        self.__dict__.pop("_children_kept", None)
        cache = self.__dict__.get("_graphs")
        if cache is not None:
            cache.close()  # captured forwards hold the old layers' buffers

    # ------------------------------------------------------------------------------------------ log-probs
    @property
    def bayesian_children(self) -> Iterator[Module]:
        """All Bayesian children (duck-typed, model.py:59-68)."""
        children = filter(is_module_bayesian, self.modules())
        children = [c for c in children if c != self]
        return children

    def _only_fused(self) -> bool:
        return self._children()[1]

    def _children(self):
        """(Bayesian children, all of them kernel-backed?) — walked once and kept: the walk over a converted BERT-base's ~600
        modules is 1.5 ms, and the reference's caller loop asks twice per sample (log_prior(), log_variational_posterior():
        /root/reference/examples/bert_glue.py:65-66).  Like `fused_children()` the result stands until `refresh()`; a kept
        child that is no longer registered where it was found (a layer swapped out) makes it stale at once."""
        kept = self.__dict__.get("_children_kept")
        if kept is not None and all(parent._modules.get(name) is child for parent, name, child in kept[2]):
            return kept[0], kept[1]
        children = list(self.bayesian_children)
        where = []
        wanted = {id(c) for c in children}
        for parent in self.modules():
            for name, child in parent._modules.items():
                if id(child) in wanted:
                    where.append((parent, name, child))
        only = all(isinstance(c, KernelLayer) for c in children)
        self.__dict__["_children_kept"] = (children, only, where)
        return children, only

    def _sum(self, index: int, name: str):
        children, only_fused = self._children()
        if not len(children):
            warnings.warn("No Bayesian Child is present in this model")
        if len(children) and self._lp_buf is not None and only_fused:
            # one reduction over the [L, S, 2] buffer the kernels wrote: sum over layers, mean over samples
            if bfr.STATE.kl_gradient and torch.is_grad_enabled():
                return self.log_prob_samples()[:, index].mean().to(torch.float32)
            return self._lp_buf[:, :, index].sum(0).mean().to(torch.float32)
        value = 0.0
        for child in children:
            value += getattr(child, name)
        return value

    def log_prior(self) -> Tensor:
        """Sum of the children's log_prior (model.py:70-78); mean over the S samples of the last forward."""
        return self._sum(0, "log_prior")

    def log_variational_posterior(self) -> Tensor:
        """Sum of the children's log_variational_posterior (model.py:81-89); mean over the S samples."""
        return self._sum(1, "log_variational_posterior")

    def _kl_params(self):
        from .parameters.gaussian import Gaussian

        params = []
        for layer in self.fused_children():
            params += [layer.weight.mu, layer.weight.rho]
            if isinstance(layer.bias, Gaussian):
                params += [layer.bias.mu, layer.bias.rho]
        return params

    def log_prob_samples(self) -> Tensor:
        """[S, 2] float64: per-sample {log_prior, log_variational_posterior} summed over the fused layers.

        Detached, as in the reference, unless bayeformers_amd.set_kl_gradient(True): then the result carries the
        Bayes-by-Backprop gradient w.r.t. every layer's mu and rho."""
        if self._lp_buf is None:
            raise RuntimeError("no forward has run yet")
        lp = self._lp_buf.sum(0)
        if bfr.STATE.kl_gradient and torch.is_grad_enabled() and self._only_fused():
            lp = _KLFn.apply(lp, self, self._last_S, self._last_seed, self._last_base, self._last_counter,
                             *self._kl_params())
        return lp

    def log_prior_samples(self) -> Tensor:
        return self.log_prob_samples()[:, 0]

    def log_variational_posterior_samples(self) -> Tensor:
        return self.log_prob_samples()[:, 1]
