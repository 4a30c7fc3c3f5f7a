"""Cross-layer sampling plan of a bnn.Model.

The reference samples every layer's weights inside that layer's forward
(/root/reference/bayeformers/nn/layers/linear.py:97).  Epsilon here is a pure function of
(seed, sample, layer, element), so W_s does not depend on the activations and the sampling + log-prob work of
many layers can be done by ONE launch (bf_sample_logprob_table) ahead of the GEMMs that consume it:

  * the model's bnn.Linear layers are cut, in registration order, into groups whose sampled weights
    (S x n x 2 B) total about GROUP_BYTES; group g lives in arena g % R of a ring of R arenas;
  * when a layer runs whose group is not sampled yet, that group AND as many of the following ones as the ring
    holds are sampled by one launch — everything that read the arenas being overwritten was enqueued before that
    point.  The ring is as long as ARENA_BYTES allows: on a 288 GB MI355X BERT-base (1.7 GB of bf16 samples at
    S = 10) and BERT-large (6 GB) fit whole, i.e. one sampling launch per forward.  Large launches matter: the
    VALU-bound kernel runs at 0.65 T eps/s in 96 MB launches and at 1.0 T eps/s over the whole model;
  * each block leaves one [S][2] fp64 row of partial log-prob sums; ONE bf_reduce_logprob launch at the end of the
    forward turns them into the per-layer {log_prior, log_q}[S] the Model sums — instead of 2 launches per layer.

If a layer runs while its arena holds another group (unusual execution order) the group is simply sampled again:
same counters, same values.
"""
import ctypes
from typing import List, Optional

import torch

from . import _C
from . import ops
from . import random as bfr

import os

GROUP_BYTES = 96 << 20
# Sampled weights held at once: group g lives in arena g % R, R = as many arenas as fit in ARENA_BYTES.  MI355X has
# 288 GB: BERT-base (S = 10: 1.7 GB of bf16 weights) and BERT-large (6 GB) fit whole, so ONE launch samples the model.
ARENA_BYTES = int(os.environ.get("BF_PLAN_ARENA_BYTES", str(16 << 30)))


def _no_plan():
    return None


class SamplePlan:
    # a cache of a bnn.Model (arenas of sampled weights, a device table of raw pointers): a copied or pickled model gets
    # none and builds its own at its first forward
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_plan, ())

    def __init__(self, layers, S: int, cdt: torch.dtype, device: torch.device, index=None, shared=()):
        """layers: the planned bnn.Linear modules; index[i] = row of layers[i] in the model's [L, S, 2] log-prob
        buffer (layers that take the single-kernel small-M path are left out, so rows may have gaps).
        shared: tuples of layers that read the same activations (query/key/value): when such a tuple is a run of
        consecutive planned layers of one shape its sampled weights are laid out back to back ([L][S][N][K], then
        [L][S][N] biases) so that ONE bf_gemm_nt_layers launch can multiply them all (self.stacked)."""
        from .nn.parameters.base import NoneParameter

        self.S, self.cdt, self.device = S, cdt, device
        self.epoch = bfr.STATE.stale_epoch  # the table bakes prior constants in: void once a kernel reported a stale one
        self.layers = layers
        self.index = list(index) if index is not None else list(range(len(layers)))
        self.shared = tuple(shared)
        self.key = self.make_key(layers, S, cdt, shared)
        lib = _C.lib()
        esz = 4 if cdt == torch.float32 else 2

        # runs of layers that share their input and can be stacked: consecutive in the plan, one shape, all with bias
        pos = {id(l): i for i, l in enumerate(layers)}
        run_of = {}
        for tup in shared:
            idx = [pos.get(id(l)) for l in tup]
            l0 = tup[0]
            ok = (None not in idx and idx == list(range(idx[0], idx[0] + len(tup))) and
                  all(self.index[i] == self.index[idx[0]] + k for k, i in enumerate(idx)) and
                  all(l.out_features == l0.out_features and l.in_features == l0.in_features and
                      not isinstance(l.bias, NoneParameter) for l in tup) and
                  (S * l0.weight.mu.numel() * esz) % 256 == 0 and (S * l0.out_features * 4) % 256 == 0)
            if ok:
                for l in tup:
                    run_of[id(l)] = tuple(tup)

        # groups of consecutive layers (a stackable run is never split)
        self.group_of, groups, cur, cur_bytes = {}, [], [], 0
        prev_row = None
        for l, row in zip(layers, self.index):
            b = S * l.weight.mu.numel() * esz
            run = run_of.get(id(l))
            inside_run = run is not None and run[0] is not l
            if cur and not inside_run and (cur_bytes + b > GROUP_BYTES or row != prev_row + 1):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(l)
            cur_bytes += b
            prev_row = row
        if cur:
            groups.append(cur)
        self.groups = groups

        # arena layout: per group, [W_s of each layer][b_s of each layer], 256-byte aligned slices
        def align(x):
            return (x + 255) // 256 * 256

        self.slices = {}
        self.stacked = {}  # id(first layer of a run) -> (run, W [L,S,N,K] view, b [L,S,N] view); filled below
        stacked_off = {}
        group_bytes = []
        for gi, g in enumerate(groups):
            off = 0
            for l in g:
                self.group_of[id(l)] = gi
                run = run_of.get(id(l))
                if run is not None:
                    if run[0] is l:  # lay the whole run out: all W back to back, then all b
                        wb, bb = S * l.weight.mu.numel() * esz, S * l.out_features * 4
                        for k, m in enumerate(run):
                            self.slices[id(m)] = (gi, off + k * wb, off + len(run) * wb + k * bb)
                        stacked_off[id(l)] = (gi, off, off + len(run) * wb)
                        off += len(run) * (wb + bb)
                    continue
                wb = align(S * l.weight.mu.numel() * esz)
                has_bias = not isinstance(l.bias, NoneParameter)
                bb = align(S * l.out_features * 4) if has_bias else 0
                self.slices[id(l)] = (gi, off, off + wb if has_bias else None)
                off += wb + bb
            group_bytes.append(off)
        arena_bytes = max(group_bytes)
        # ring of arenas: group g lives in arena g % R; a launch samples as many consecutive groups as the ring holds
        self.arenas = [torch.empty(arena_bytes, dtype=torch.uint8, device=device) for _ in range(max(1, min(len(groups), ARENA_BYTES // max(arena_bytes, 1))))]
        self.arena_owner = [None] * len(self.arenas)

        self.pending = set()  # groups sampled in the running forward whose block partials are not reduced yet
        self._kinds_of = {}   # (first group, last group) -> set of prior kinds of a launch over those groups

        # table: entries in layer order (weight, then bias)
        n_entries = sum(1 + (self.slices[id(l)][2] is not None) for l in layers)
        arr = (_C.bf_tensor_t * n_entries)()
        # The table bakes in which Gaussian priors are aliases of their posterior's frozen mean (ops.prior_alias: the kernel
        # then reads neither the prior's mu nor its rho).  That was decided on the tensors' contents: `alias_watch` keeps
        # their version counters, and Model rebuilds the plan when one of them moved (alias_valid).
        self.alias_watch = []
        self.views = {}
        first_entry_of_layer = []
        e = 0
        for li, l in enumerate(layers):
            gi, woff, boff = self.slices[id(l)]
            arena = self.arenas[gi % len(self.arenas)]
            first_entry_of_layer.append(e)
            N, K = l.out_features, l.in_features
            if not ops.fill_tensor(arr[e], l.weight, l.weight_prior, 2 * l.layer_id):
                raise _C.BayeFormersAMDError("SamplePlan: user-defined priors are not plannable")
            self._watch(arr[e], l.weight, l.weight_prior)
            wv = arena[woff:woff + S * N * K * esz].view(cdt).view(S, N, K)
            arr[e].d_sample_out, arr[e].out_dtype = wv.data_ptr(), ops._TORCH2BF[cdt]
            e += 1
            bv = None
            if boff is not None:
                if not ops.fill_tensor(arr[e], l.bias, l.bias_prior, 2 * l.layer_id + 1):
                    raise _C.BayeFormersAMDError("SamplePlan: user-defined priors are not plannable")
                self._watch(arr[e], l.bias, l.bias_prior)
                bv = arena[boff:boff + S * N * 4].view(torch.float32).view(S, N)
                arr[e].d_sample_out, arr[e].out_dtype = bv.data_ptr(), _C.BF_DT_F32
                e += 1
            self.views[id(l)] = (wv, bv)
        for tup in {id(r[0]): r for r in run_of.values()}.values():
            gi, woff, boff = stacked_off[id(tup[0])]
            arena = self.arenas[gi % len(self.arenas)]
            L, N, K = len(tup), tup[0].out_features, tup[0].in_features
            self.stacked[id(tup[0])] = (tup, arena[woff:woff + L * S * N * K * esz].view(cdt).view(L, S, N, K),
                                        arena[boff:boff + L * S * N * 4].view(torch.float32).view(L, S, N))
        total = ctypes.c_uint32()
        nbytes = lib.bf_sample_table_bytes(arr, n_entries, ctypes.byref(total))
        blob = torch.empty(nbytes, dtype=torch.uint8, pin_memory=False)
        begin = torch.empty(n_entries + 1, dtype=torch.int32)
        kinds = torch.empty(n_entries, dtype=torch.int32)  # the tensors' effective prior kinds
        _C.check(lib.bf_sample_table_build(arr, n_entries, blob.data_ptr(), nbytes, begin.data_ptr(), kinds.data_ptr()),
                 "bf_sample_table_build")
        self.entry_kinds = kinds.tolist()
        self.first_entry_of_layer = first_entry_of_layer + [n_entries]
        self.n_entries, self.total_blocks = n_entries, total.value
        self.blob = blob.to(device)
        begin_l = begin.tolist()
        layer_rows = [begin_l[fe] for fe in first_entry_of_layer] + [begin_l[-1]]
        self.layer_rows = torch.tensor(layer_rows, dtype=torch.int32, device=device)
        self.layer_rows_host = layer_rows
        self.partials = torch.empty((self.total_blocks, S, 2), dtype=torch.float64, device=device)
        # per group: (first plan-layer index, number of layers, first block, end block, first row of the model buffer)
        self.group_span = []
        li = 0
        for g in groups:
            self.group_span.append((li, len(g), layer_rows[li], layer_rows[li + len(g)], self.index[li]))
            li += len(g)
        self.scalars = sum(l.weight.mu.numel() + (l.out_features if self.slices[id(l)][2] is not None else 0) for l in layers)

    def _watch(self, entry, gaussian, prior) -> None:
        if entry.prior.kind == _C.BF_PRIOR_GAUSSIAN and entry.prior.pi == 1.0:
            for t in (gaussian.mu, prior.mu, prior.rho):
                self.alias_watch.append((t, t._version))

    def alias_valid(self) -> bool:
        """False once a tensor behind an aliased prior was edited in place since the table was built."""
        for t, v in self.alias_watch:
            if t._version != v:
                return False
        return True

    @staticmethod
    def make_key(layers, S, cdt, shared=()):
        from .nn.parameters.gaussian import Gaussian

        key = [S, cdt, len(layers), tuple(tuple(l.layer_id for l in t) for t in shared)]
        for l in layers:
            key.append(l.layer_id)
            for g in (l.weight, l.bias, l.weight_prior, l.bias_prior):
                if isinstance(g, Gaussian):
                    key.append(g.mu.data_ptr())
                    key.append(g.rho.data_ptr())
                else:
                    key.append(id(g))
                    c = getattr(g, "constants", None)
                    if c is not None:
                        key.append(c())
        return tuple(key)

    @staticmethod
    def plannable(layers) -> bool:
        from .nn.parameters.base import NoneParameter
        from .nn.parameters.gaussian import Gaussian, ScaledGaussianMixture

        if not layers:
            return False
        ok = (Gaussian, ScaledGaussianMixture, NoneParameter)
        dev = layers[0].weight.mu.device
        return dev.type == "cuda" and all(
            isinstance(l.weight_prior, ok) and isinstance(l.bias_prior, ok) and l.weight.mu.device == dev and
            l.compute_dtype is None for l in layers)

    def _sample_groups(self, first: int, last: int, token, seed: int, sample_base: int):
        """ONE launch over the (contiguous) blocks of groups first..last; they must map to distinct arenas."""
        b0, b1 = self.group_span[first][2], self.group_span[last][3]
        # the set of prior kinds among the launched tensors: a uniform launch runs the kernel compiled for its kind alone
        mask = self._kinds_of.get((first, last))
        if mask is None:
            l0 = self.group_span[first][0]
            l1 = self.group_span[last][0] + self.group_span[last][1]
            mask = 0
            for k in self.entry_kinds[self.first_entry_of_layer[l0]:self.first_entry_of_layer[l1]]:
                mask |= 1 << k
            if os.environ.get("BF_NO_UNIFORM_TABLE") is not None:
                mask = 0  # developer A/B: always the kernel that takes every prior kind
            self._kinds_of[(first, last)] = mask
        _C.check(_C.lib().bf_sample_logprob_table(self.blob.data_ptr(), self.n_entries, b0, b1, self.S, seed,
                                                  sample_base & 0xFFFFFFFF, self.partials.data_ptr(), mask,
                                                  ops._stream_ptr()), "bf_sample_logprob_table")
        for gi in range(first, last + 1):
            self.pending.add(gi)
            self.arena_owner[gi % len(self.arenas)] = (gi, token)

    def ensure(self, layer, token, seed: int, sample_base: int, lp_buf: torch.Tensor):
        """Make sure `layer`'s group has been sampled for the forward identified by `token`; returns (W_s, b_s)."""
        gi = self.group_of[id(layer)]
        if self.arena_owner[gi % len(self.arenas)] != (gi, token):
            # this group and as many of the following ones as the arena ring holds, in one launch: everything that
            # read the arenas being overwritten was enqueued before this point
            last = min(len(self.groups), gi + len(self.arenas)) - 1
            while last > gi and self.arena_owner[last % len(self.arenas)] == (last, token):
                last -= 1
            self._sample_groups(gi, last, token, seed, sample_base)
        return self.views[id(layer)]

    def finish(self, lp_buf: torch.Tensor) -> None:
        """Reduce the block partials of every group sampled since the last call into the model's [L, S, 2] buffer:
        one bf_reduce_logprob launch per run of groups whose layers are consecutive rows (normally one per forward)."""
        if not self.pending:
            return
        lib = _C.lib()
        stream = ops._stream_ptr()
        run = None  # (first plan-layer, number of layers, first row)
        for gi in sorted(self.pending) + [None]:
            span = self.group_span[gi] if gi is not None else None
            if run is not None and span is not None and span[0] == run[0] + run[1] and span[4] == run[2] + run[1]:
                run = (run[0], run[1] + span[1], run[2])
                continue
            if run is not None:
                _C.check(lib.bf_reduce_logprob(self.partials.data_ptr(), self.layer_rows.data_ptr() + 4 * run[0], run[1],
                                               self.S, lp_buf.data_ptr() + run[2] * self.S * 2 * 8, stream),
                         "bf_reduce_logprob")
            run = (span[0], span[1], span[4]) if span is not None else None
        self.pending.clear()
