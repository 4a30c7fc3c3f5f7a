"""Counter-based RNG state of the Monte-Carlo path.

The reference draws eps from torch's global generator in layer-execution order
(/root/reference/bayeformers/nn/parameters/gaussian.py:100).  Here eps is a pure function of
(seed, sample index, 2*layer_id + tensor_id, element index) — csrc/bf_philox.h — so the only state is the seed
and the next unused Monte-Carlo sample index.  Every forward of a bnn.Model (or of a bare bnn.Linear) reserves S
consecutive sample indices; all layers inside one forward share them, which is what makes results independent
of kernel tiling, of S-batching and of how samples are sharded over GPUs.
"""
import os

import torch

DEFAULT_SEED = 0x5EED

_DTYPES = {
    "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, torch.bfloat16: torch.bfloat16,
    "fp16": torch.float16, "float16": torch.float16, "half": torch.float16, torch.float16: torch.float16,
    "fp32": torch.float32, "float32": torch.float32, torch.float32: torch.float32,
}


class _State:
    """Process-wide (NOT thread-local): autograd runs backward nodes on its own thread, which must see the same seed,
    device counter and running forward as the thread that ran the forward — like the reference's use of torch's
    global generator.  One process drives one GPU (one rank per device), so there is nothing to keep apart."""

    def __init__(self):
        self.seed = DEFAULT_SEED
        self.next_sample = 0
        self.ctx = None  # (sample_base, S) while a bnn.Model forward is running
        self.live_ctxs = []  # finished grad-enabled bnn.Model forwards whose graphs may still run backward (recompute_context)
        self.compute_dtype = torch.bfloat16
        self.next_layer_id = 0
        self.device_counter = None  # 1-element int32 device tensor when the sample counter lives on the GPU
        self.device_drop_counter = None  # ... and, with it, the dropout `call` counter (1-element int32)
        self.kl_gradient = False    # opt-in Bayes-by-Backprop gradient of the KL terms (the reference has none)
        self.next_dropout_call = 0  # dropout contract (csrc/bf_philox.h): one `call` number per forward
        self.next_dropout_site = 1  # ... and one `site` number per module that applies a dropout
        self.stale_epoch = 0        # the library's stale-prior counter as of the running forward (ops.refresh_stale_epoch)
        self.counter_moves = 0      # host-side count of the writes to device_counter (counter_snapshot's cache key)


STATE = _State()


def manual_seed(seed: int, next_sample: int = 0) -> None:
    """Seed the Philox key and rewind the Monte-Carlo sample counter."""
    STATE.seed = int(seed) & (2 ** 64 - 1)
    STATE.next_sample = int(next_sample) & 0xFFFFFFFF
    STATE.next_dropout_call = 0
    if STATE.device_counter is not None:
        v = STATE.next_sample if STATE.next_sample < 2 ** 31 else STATE.next_sample - 2 ** 32
        STATE.device_counter.fill_(v)
        STATE.device_drop_counter.zero_()
        STATE.counter_moves += 1


def get_state():
    return STATE.seed, STATE.next_sample


def reserve_samples(n: int) -> int:
    """Reserve n consecutive MC sample indices; returns the first (host-side part of it).

    With a device-resident counter (use_device_counter) the host part is always 0: the kernels add the counter
    themselves and commit_samples() advances it after the forward has been enqueued."""
    if STATE.device_counter is not None:
        return 0
    base = STATE.next_sample
    STATE.next_sample = (base + int(n)) & 0xFFFFFFFF
    return base


def commit_samples(n: int) -> None:
    """Call after the kernels of a forward are enqueued: moves a device-resident counter past the n indices used."""
    if STATE.device_counter is not None:
        STATE.device_counter.add_(int(n))
        STATE.counter_moves += 1


def use_device_counter(enable: bool = True, device="cuda") -> None:
    """Keep the Monte-Carlo sample counter in device memory (bf_set_sample_counter).

    Needed to capture a step in a HIP graph (torch.cuda.graph): a graph bakes kernel arguments, so a host-side
    sample index would make every replay draw the same epsilon; with the counter on the device the captured
    `counter += S` gives each replay fresh samples — replay k of a graph behaves like the k-th eager step."""
    from . import _C

    if enable:
        if STATE.device_counter is None:
            v = STATE.next_sample if STATE.next_sample < 2 ** 31 else STATE.next_sample - 2 ** 32
            c = STATE.next_dropout_call if STATE.next_dropout_call < 2 ** 31 else STATE.next_dropout_call - 2 ** 32
            # (ordinary tensors even when the caller runs under torch.inference_mode(): they are updated in place for the rest of
            # the process, in whatever mode later forwards run)
            with torch.inference_mode(False):
                STATE.device_counter = torch.full((1,), v, dtype=torch.int32, device=device)
                # the dropout `call` numbers move with it: the device counter takes over where the host's stands, the host part
                # every later forward hands the kernels is 0
                STATE.device_drop_counter = torch.full((1,), c, dtype=torch.int32, device=device)
        # the library keeps one pointer per HIP device, selected by the CURRENT device: make it the counter's
        with torch.cuda.device(STATE.device_counter.device):
            _C.check(_C.lib().bf_set_sample_counter(STATE.device_counter.data_ptr()), "bf_set_sample_counter")
    else:
        if STATE.device_counter is not None:
            STATE.next_sample = int(STATE.device_counter.item()) & 0xFFFFFFFF
            STATE.next_dropout_call = int(STATE.device_drop_counter.item()) & 0xFFFFFFFF
            with torch.cuda.device(STATE.device_counter.device):
                _C.check(_C.lib().bf_set_sample_counter(None), "bf_set_sample_counter")
        STATE.device_counter = STATE.device_drop_counter = None


def set_compute_dtype(dtype) -> None:
    """MFMA operand precision of the sampled-weight GEMM: 'bf16' (default), 'fp16', or 'fp32' (exact, 1/16 rate)."""
    STATE.compute_dtype = _DTYPES[dtype]


def get_compute_dtype() -> torch.dtype:
    return STATE.compute_dtype


def _graph_task_id() -> int:
    try:
        return torch._C._current_graph_task_id()
    except AttributeError:  # pragma: no cover - private API of the installed torch
        return -1


LIVE_CONTEXTS = 4  # finished forwards remembered for the recomputation of their checkpointed blocks


def remember_context(ctx, output) -> None:
    """Called at the end of a grad-enabled bnn.Model forward: keeps its (slimmed) context for `recompute_context` and
    binds it to the autograd graph it produced — a hook on the grad_fn of every output tensor stamps the context with the
    id of the backward pass that reaches it, so a checkpointed block recomputed during THAT backward finds the forward it
    belongs to even when other forwards (a second loss, an evaluation pass, another model) ran in between."""
    def tensors(o, depth=0):
        if isinstance(o, torch.Tensor):
            yield o
        elif depth >= 6:
            return
        elif isinstance(o, dict):
            for v in o.values():
                yield from tensors(v, depth + 1)
        elif isinstance(o, (tuple, list)):
            for v in o:
                yield from tensors(v, depth + 1)
        elif hasattr(o, "__dict__") and not isinstance(o, (type, torch.nn.Module)):  # dataclasses, plain result objects
            for v in vars(o).values():
                yield from tensors(v, depth + 1)

    def stamp(*_):
        tid = _graph_task_id()
        if tid != -1:
            ctx.graph_tasks.add(tid)

    seen = set()
    for t in tensors(output):
        fn = t.grad_fn
        if fn is not None and id(fn) not in seen:
            seen.add(id(fn))
            fn.register_prehook(stamp)
    STATE.live_ctxs.append(ctx)
    del STATE.live_ctxs[:-LIVE_CONTEXTS]


def recompute_context():
    """The context of the bnn.Model forward a Bayesian layer belongs to when it is called with no forward running WHILE
    autograd executes a backward pass: that is the recomputation of a checkpointed block (torch.utils.checkpoint), which
    must draw the epsilon of the forward it repeats — the same sample indices, the same (device) counter value — not fresh
    ones.  The forward is the one whose outputs this backward pass has reached (`remember_context`); if that is not
    unique (two forwards of checkpointed models in one backward, e.g. loss1 + loss2) the call raises instead of guessing."""
    tid = _graph_task_id()
    if tid == -1 or not STATE.live_ctxs:
        return None
    hit = [c for c in STATE.live_ctxs if tid in c.graph_tasks]
    if len(hit) == 1:
        return hit[0]
    if not hit:
        # the loss was built on something the output walk did not reach (an output type it does not know): a forward no
        # backward pass has claimed yet can still own this one — finished ones (claimed by EARLIER backward passes) cannot.
        # One such candidate is the answer (the usual forward / backward / forward / backward loop, from its second step
        # on too); several are ambiguous.
        free = [c for c in STATE.live_ctxs if not c.graph_tasks]
        if len(free) == 1:
            free[0].graph_tasks.add(tid)  # claimed: the rest of this backward finds it at once, later ones pass it over
            return free[0]
    raise RuntimeError(
        "bayeformers_amd: a Bayesian layer is being recomputed during backward (a checkpointed block) and "
        f"{len(hit) or len(STATE.live_ctxs)} finished bnn.Model forwards could own it; run backward() after each "
        "grad-enabled forward of a checkpointed model (evaluation passes belong under torch.no_grad())")


def reserve_dropout_call() -> int:
    """A fresh `call` number of the dropout contract (csrc/bf_philox.h): every bnn.Model forward takes one, so that all the
    dropouts of that forward — and their regeneration in backward or in a recomputed checkpointed block — share it.
    Device-counter mode: the host part is always 0, the number lives in `reserve_dropout_counter()`'s copy."""
    if STATE.device_drop_counter is not None:
        return 0
    c = STATE.next_dropout_call
    STATE.next_dropout_call = (c + 1) & 0xFFFFFFFF
    return c


def reserve_dropout_counter(needed: bool = True):
    """Device-counter mode: this forward's COPY of the device-resident call counter (what its dropout kernels, their
    backward and a recomputed block add to `call`), taken before the counter moves on by one; None in host mode — and for a
    forward of a model that is not in training mode (`needed` False: no dropout will ask), which then leaves the counter
    where it is, exactly as the host counter only matters to forwards that drop."""
    if STATE.device_drop_counter is None:
        return None
    if not needed:
        STATE.device_drop_counter.add_(1)
        return None
    snap = STATE.device_drop_counter.clone()
    STATE.device_drop_counter.add_(1)
    return snap


def dropout_counter():
    """The call-counter copy for a dropout applied now (see dropout_call): the running forward's, the original forward's in
    a recomputed block; a fresh one outside any forward in device-counter mode; None in host mode."""
    ctx = STATE.ctx if STATE.ctx is not None else recompute_context()
    if ctx is None:
        return reserve_dropout_counter()
    if ctx.drop_counter is None and STATE.device_drop_counter is not None and ctx is STATE.ctx:
        # a forward that was not expected to drop (the bnn.Model in eval mode, a child switched to train() on its own): the
        # counter has moved past this forward's number by exactly one — no other forward can have begun since
        ctx.drop_counter = STATE.device_drop_counter.clone().sub_(1)
    return ctx.drop_counter


def dropout_call() -> int:
    """The `call` number for a dropout applied now: the running bnn.Model forward's, the original forward's when a
    checkpointed block is being recomputed during backward, a fresh one outside any forward."""
    ctx = STATE.ctx if STATE.ctx is not None else recompute_context()
    return ctx.drop_call if ctx is not None else reserve_dropout_call()


def dropout_origin():
    """(first global sample of this process's shard within the step, samples in the shard) of the running — or, in a
    recomputed block, the original — bnn.Model forward: what makes a sample's dropout masks independent of the sharding."""
    ctx = STATE.ctx if STATE.ctx is not None else recompute_context()
    return (ctx.shard_start, ctx.S) if ctx is not None else (0, 1)


def dropout_site(module) -> int:
    """The `site` number of a module: assigned at its first dropout, in execution order, from the counter of the bnn.Model whose
    forward is running — a model's masks are a function of (seed, call, its own modules), not of what other models this process
    ran before (round 6: the counter used to be process-wide).  Outside any bnn.Model forward: a process-wide counter."""
    site = getattr(module, "_bf_drop_site", None)
    if site is None:
        ctx = STATE.ctx if STATE.ctx is not None else recompute_context()
        counter = getattr(ctx, "drop_sites", None) if ctx is not None else None
        if counter is not None:
            site = module._bf_drop_site = counter[0]
            counter[0] += 1
        else:
            site = module._bf_drop_site = STATE.next_dropout_site
            STATE.next_dropout_site += 1
    return site


def new_layer_id() -> int:
    i = STATE.next_layer_id
    STATE.next_layer_id += 1
    return i


def set_kl_gradient(enable: bool = True) -> None:
    """Make `Model.log_prior()` / `log_variational_posterior()` differentiable w.r.t. mu and rho (bf_kl_grad).

    Off by default: the reference stores its log-probs with `.data =` (layers/linear.py:99-102), so its ELBO only
    trains the likelihood term; switching this on gives the gradient the ELBO formula implies (Blundell et al.)."""
    STATE.kl_gradient = bool(enable)


_AB_OLD_SNAPSHOT = os.environ.get("BF_AB_OLD_SNAPSHOT", "0") == "1"


def counter_snapshot(needed: bool = True):
    """In device-counter mode: a copy of the counter as the forward saw it (backward regenerates the same eps).
    `needed` False — no gradient will be asked of this forward — gives None: the copy is a 4 us kernel, and every
    Bayesian layer of a BERT-base forward used to launch one.  Inside a bnn.Model forward the layers share ONE copy for
    as long as nothing moved the counter (commit_samples / manual_seed count their moves in STATE.counter_moves)."""
    counter = STATE.device_counter
    if _AB_OLD_SNAPSHOT:  # tools/r5r_ab.sh: one copy per call, as before round 5
        return counter.clone() if counter is not None else None
    if counter is None or not needed:
        return None
    fwd = STATE.ctx
    if fwd is None:
        return counter.clone()
    if getattr(fwd, "_replaying", False) and fwd.counter is not None:
        # a checkpointed block recomputed during backward (_ForwardContext.replay): the autograd nodes built NOW are the
        # ones that run backward under use_reentrant=True, and the live counter has moved past this forward — they must
        # regenerate epsilon from the value the forward itself saw
        return fwd.counter
    snap = getattr(fwd, "_counter_snap", None)
    if snap is None or snap[0] is not counter or snap[1] != STATE.counter_moves:
        snap = fwd._counter_snap = (counter, STATE.counter_moves, counter.clone())
    return snap[2]


class counter_override:
    """Temporarily point the kernels at a saved counter value (used by backward passes, which run on autograd's own
    thread).  Restores exactly the pointer that was set before, read back from the library."""

    def __init__(self, snapshot):
        self.snapshot = snapshot
        self.prev = None

    def __enter__(self):
        self.outer = getattr(STATE, "override_snapshot", None)
        STATE.override_snapshot = self.snapshot  # (what a deferred reduction of this backward pass must run under too)
        if self.snapshot is not None:
            from . import _C
            lib = _C.lib()
            with torch.cuda.device(self.snapshot.device):
                self.prev = lib.bf_get_sample_counter()
                _C.check(lib.bf_set_sample_counter(self.snapshot.data_ptr()), "bf_set_sample_counter")

    def __exit__(self, *exc):
        STATE.override_snapshot = self.outer
        if self.snapshot is not None:
            from . import _C
            with torch.cuda.device(self.snapshot.device):
                _C.check(_C.lib().bf_set_sample_counter(self.prev), "bf_set_sample_counter")
