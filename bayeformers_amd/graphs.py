"""HIP-graph replay of whole forwards: the state shared by `sampling.GraphedSampler` (an S-sample step) and
`GraphedForward` (ONE `bnn.Model` forward — what the reference's own caller loop runs S times,
/root/reference/examples/bert_glue.py:63-66).

A forward of a converted BERT-base is ~1.9 ms of kernels behind 4-5 ms of Python and launch calls: the unchanged reference
loop `for s in range(S): model(**inputs); model.log_prior(); ...` is bound by the host.  `bnn.Model.__call__` therefore replays
an evaluation forward (eval mode, no gradient recorded, one Monte-Carlo sample per call, CUDA tensor inputs) from a HIP graph
once it has seen the same call signature twice: the Monte-Carlo sample counter moves to device memory
(`random.use_device_counter`), so replay k draws the epsilon the k-th eager forward would have drawn, and the model's
`log_prior()` / `log_variational_posterior()` read the buffer the replayed kernels wrote.  Outputs are handed out as COPIES
(a loop that keeps `outputs.logits` of every call must not find the last call in all of them).

What a capture bakes in — the Philox seed (a kernel argument), the compute dtype, the sampling plan (which priors alias their
frozen means, where the sampled weights live, every parameter's address) — is compared with the current state before every
replay (`baked_state`); a difference captures again.
"""
import os
from typing import Any, Optional

import torch
from torch import Tensor

from . import random as bfr

_USERS = {"n": 0, "moved": False}  # open graph holders; whether the first of them moved the sample counter to the device


def acquire_counter(device) -> None:
    """A graph is about to bake the device counter's ADDRESS into its launches: the counter must live on the device until the
    last holder has released it."""
    if _USERS["n"] == 0:
        _USERS["moved"] = bfr.STATE.device_counter is None
    _USERS["n"] += 1
    bfr.use_device_counter(True, device=device)


def release_counter() -> None:
    """When the last holder lets go and the holders had moved the counter to the device, it moves back to the host (advanced
    by what the replays consumed)."""
    _USERS["n"] -= 1
    if _USERS["n"] == 0 and _USERS["moved"]:
        try:
            bfr.use_device_counter(False)
        except Exception:  # interpreter shutdown: the HIP runtime may be gone already
            pass


def plan_key(model):
    """The key `bnn.Model.__call__` would build its sampling plan under NOW (addresses of every planned parameter, sample
    count, compute dtype, stacked runs), or None when the model has no plan."""
    from .plan import SamplePlan

    plan = getattr(model, "_plan", None)
    if plan is None:
        return None
    return SamplePlan.make_key(plan.layers, plan.S, bfr.get_compute_dtype(), plan.shared)


def baked_state(model):
    """The host state a captured forward of `model` bakes into its launches."""
    plan = getattr(model, "_plan", None)
    return bfr.STATE.seed, bfr.get_compute_dtype(), plan, (plan.key if plan is not None else None), bfr.STATE.stale_epoch


def still_valid(model, baked) -> bool:
    from . import ops

    ops.refresh_stale_epoch()  # (a replay whose kernels found a stale prior bumped the library's counter)
    seed, cdt, plan, key, epoch = baked
    now = baked_state(model)
    if not (now[0] == seed and now[1] == cdt and now[2] is plan and now[4] == epoch):
        return False
    # the plan's key as it would be built now: a parameter whose storage moved (`p.data = ...`, `.to()`) voids the capture
    return plan is None or (plan.alias_valid() and plan_key(model) == key)


class GraphCache:
    """What a bnn.Model keeps of captured graphs: the GraphedSamplers of `sample_bayesian(graph=True)` and the GraphedForwards
    of its own `__call__`.  A cache only — a copied or pickled model starts with an empty one (captured graphs can be neither
    copied nor pickled)."""

    def __init__(self):
        self.samplers = []   # [(key, GraphedSampler)], most recent last
        self.forwards = []   # [(signature, GraphedForward)], most recent last
        self.seen = {}       # signature -> eager calls seen so far (a forward is captured at its third call)
        self.refused = set()  # signatures whose capture failed or whose outputs cannot be copied: always eager
        self.evictions = 0    # captured forwards closed to make room: a loop over more than KEEP signatures would capture for ever

    def __deepcopy__(self, memo):
        return GraphCache()

    def __reduce__(self):
        return (GraphCache, ())

    def close(self) -> None:
        for _, s in self.samplers + self.forwards:
            s.close()
        self.samplers, self.forwards, self.seen = [], [], {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SCALARS = (type(None), bool, int, float, str)


def call_signature(args, kwargs):
    """(signature, device) of a call whose tensors are all on one CUDA device and whose other arguments are plain scalars;
    (None, None) for anything else (nested containers, CPU tensors, objects): such calls run eagerly."""
    sig, device = [], None
    for name, v in list(enumerate(args)) + sorted(kwargs.items()):
        if isinstance(v, Tensor):
            if not v.is_cuda or v.requires_grad or (device is not None and v.device != device):
                return None, None
            device = v.device
            sig.append((name, tuple(v.shape), v.dtype, tuple(v.stride())))
        elif isinstance(v, _SCALARS):
            sig.append((name, v))
        else:
            return None, None
    if device is None:
        return None, None
    sig.append(("device", device.index))
    return tuple(sig), device


def _map_tensors(obj: Any, fn, depth: int = 0):
    """`obj` with fn applied to every tensor; raises TypeError on a leaf that is neither a tensor nor a plain scalar."""
    if isinstance(obj, Tensor):
        return fn(obj)
    if isinstance(obj, _SCALARS):
        return obj
    if depth > 6:
        raise TypeError("output nested too deeply")
    if isinstance(obj, dict):  # HF ModelOutput is an OrderedDict subclass built from keyword arguments
        items = {k: _map_tensors(v, fn, depth + 1) for k, v in obj.items()}
        return type(obj)(**items) if type(obj) is not dict else items
    if isinstance(obj, (tuple, list)):
        vals = [_map_tensors(v, fn, depth + 1) for v in obj]
        if hasattr(obj, "_fields"):  # namedtuple
            return type(obj)(*vals)
        return type(obj)(vals)
    raise TypeError(f"cannot copy an output of type {type(obj).__name__}")


class GraphedForward:
    """ONE evaluation forward of a bnn.Model (one signature) as a HIP graph: replay k = eager forward k, bit for bit."""

    def __init__(self, model, args, kwargs, device, warmup: int = 1) -> None:
        self.model, self.device = model, device
        # (the captured input buffers are written on every later call, possibly outside the torch.inference_mode() this call may
        # be running under: they must be ordinary tensors, not inference tensors)
        with torch.inference_mode(False):
            self._args = tuple(a.clone() if isinstance(a, Tensor) else a for a in args)
            self._kwargs = {k: (v.clone() if isinstance(v, Tensor) else v) for k, v in kwargs.items()}
        self.graph = self._static = None
        self.captures = 0
        self._warmup = max(1, int(warmup))
        acquire_counter(device)
        self._open = True
        try:
            self._capture()
        except BaseException:
            self.close()
            raise

    def _capture(self) -> None:
        self.graph = self._static = None
        model = self.model
        total = model._mc_span[1]
        with torch.inference_mode(False), torch.no_grad(), torch.cuda.device(self.device):
            for _ in range(self._warmup):  # plans, workspaces and tile schedules are built outside the capture
                model._eager_call(*self._args, **self._kwargs)
            # the warm-up forwards consumed sample indices (and dropout call numbers) the caller never saw: hand them back
            bfr.STATE.device_counter.sub_(self._warmup * total)
            bfr.STATE.device_drop_counter.sub_(self._warmup)
            bfr.STATE.counter_moves += 1
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model._eager_call(*self._args, **self._kwargs)
            _map_tensors(out, lambda t: t)  # refuse (TypeError) what __call__ could not hand out as copies
            self.graph, self._static = graph, out
        self._layers = list(model.fused_children())
        self._lp_views = [l._lp_view for l in self._layers]
        self._baked = baked_state(model)
        self.captures += 1

    def __call__(self, args, kwargs):
        if self.graph is None:
            raise RuntimeError("GraphedForward: closed")
        if not still_valid(self.model, self._baked):
            self._capture()
        for dst, src in zip(self._args, args):
            if isinstance(src, Tensor):
                dst.copy_(src)
        for k, src in kwargs.items():
            if isinstance(src, Tensor):
                self._kwargs[k].copy_(src)
        self.graph.replay()
        for l, v in zip(self._layers, self._lp_views):  # what Linear.forward leaves behind: the layer's lazy log-prob scalars
            l._lp_view, l._lp_dirty = v, True
        return _map_tensors(self._static, lambda t: t.clone())

    def close(self) -> None:
        self.graph = self._static = None
        if self._open:
            self._open = False
            release_counter()


AUTO_AFTER = 2   # eager calls of one signature before it is captured (a one-off call never pays for a capture)
KEEP = 2         # captured signatures kept per model
MAX_EVICTIONS = 4  # ... and how often one of them may be closed for another before new signatures stay eager
DISABLED = os.environ.get("BF_NO_AUTO_GRAPH") is not None


def auto_forward(model, args, kwargs) -> Optional[Any]:
    """`bnn.Model.__call__`'s replay path: the output of this call from a HIP graph, or None when the call is to run eagerly
    (not an evaluation forward of the kind described in the module docstring, or a signature not seen often enough yet)."""
    if (DISABLED or not getattr(model, "graph_replay", True) or torch.is_grad_enabled() or model.training
            or model._mc_samples != 1 or model._mc_span != (0, 1) or getattr(model, "_mc_harness", 0)
            or not torch.cuda.is_available()
            or torch.cuda.is_current_stream_capturing()):
        return None
    sig, device = call_signature(args, kwargs)
    if sig is None:
        return None
    cache = model.__dict__.get("_graphs")
    if cache is None:
        cache = model.__dict__["_graphs"] = GraphCache()
    if sig in cache.refused:
        return None
    for i, (k, fw) in enumerate(cache.forwards):
        if k == sig and fw.graph is not None:
            cache.forwards.append(cache.forwards.pop(i))
            return fw(args, kwargs)
    n = cache.seen.get(sig, 0)
    if n < AUTO_AFTER:
        if len(cache.seen) > 16:
            cache.seen.clear()
        cache.seen[sig] = n + 1
        return None
    if any(m.training for m in model.modules()):  # a child switched to train() on its own: dropout masks are per call
        cache.refused.add(sig)  # (not asked again — the walk costs 0.5 ms — until the model's mode is set anew: Model.train)
        return None
    if cache.evictions >= MAX_EVICTIONS:  # a loop that cycles through more signatures than are kept: capturing does not pay
        return None
    while len(cache.forwards) >= KEEP:
        cache.forwards.pop(0)[1].close()
        cache.evictions += 1
        if cache.evictions == MAX_EVICTIONS:
            import warnings

            warnings.warn(f"bayeformers_amd: this model's evaluation forwards come in more than {KEEP} batch signatures; they keep "
                          "running eagerly from now on (bucket the batch shapes, or use sampling.GraphedSampler per shape)")
    try:
        fw = GraphedForward(model, args, kwargs, device)
    except Exception as e:  # noqa: BLE001 - whatever the capture raised, the eager forward is still the product path
        import warnings

        cache.refused.add(sig)
        warnings.warn(f"bayeformers_amd: this forward cannot be replayed from a HIP graph and keeps running eagerly "
                      f"({type(e).__name__}: {str(e)[:200]})")
        return None
    cache.forwards.append((sig, fw))
    return fw(args, kwargs)
