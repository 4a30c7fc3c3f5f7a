# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.layers.base — plumbing shared by the kernel-backed Bayesian layers."""
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import Module

from ... import random as bfr

_LOGPROB_NAMES = ("log_prior", "log_variational_posterior")


class KernelLayer(Module):
    """A Bayesian layer whose forward runs in the HIP kernels and leaves {log_prior, log_variational_posterior} per
    Monte-Carlo sample in a [S, 2] float64 slot.  The two reference attributes (0-d fp32 Parameters, state-dict
    compatible) are refreshed lazily from that slot on access."""

    def _init_kernel_layer(self) -> None:
        self.register_parameter("log_prior", nn.Parameter(torch.tensor(0.), requires_grad=False))
        self.register_parameter("log_variational_posterior", nn.Parameter(torch.tensor(0.), requires_grad=False))
        self.layer_id = bfr.new_layer_id()
        self.compute_dtype = None
        self._lp_own = None    # [S, 2] float64 buffer when the layer is used outside a bnn.Model
        self._lp_view = None   # where the last forward wrote {log_prior, log_q} per sample
        self._lp_dirty = False

    def __getattr__(self, name):
        if name in _LOGPROB_NAMES and self.__dict__.get("_lp_dirty", False):
            self._sync_logprobs()
        return super(KernelLayer, self).__getattr__(name)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        """state_dict() reads `_parameters` directly: refresh the two log-prob scalars first, so that a checkpoint
        holds the values of the last forward like the reference's does (layers/linear.py:99-102 assign them eagerly;
        /root/reference/examples/bert_glue.py:303-309 saves them)."""
        if self.__dict__.get("_lp_dirty", False):
            self._sync_logprobs()
        super(KernelLayer, self)._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """A checkpoint's log-prob scalars replace those of an earlier forward: drop the pending lazy refresh, or the next
        attribute access would overwrite the loaded values with that forward's (the reference keeps what it loaded until
        the next forward, layers/linear.py:99-102)."""
        if any((prefix + n) in state_dict for n in _LOGPROB_NAMES):
            self.__dict__["_lp_dirty"] = False
        super(KernelLayer, self)._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _sync_logprobs(self) -> None:
        self.__dict__["_lp_dirty"] = False
        v = self._lp_view.mean(0).to(torch.float32)
        self._parameters["log_prior"].data = v[0]
        self._parameters["log_variational_posterior"].data = v[1]

    @property
    def log_prob_samples(self) -> Optional[Tensor]:
        """[S, 2] float64 {log_prior, log_variational_posterior} of each sample of the last forward."""
        return self._lp_view

    def _begin(self, device):
        """(ctx, sample_base, S, slot) of this forward: shared with the enclosing bnn.Model, or a private one."""
        ctx = bfr.STATE.ctx
        if ctx is not None:
            base, S, slot = ctx.sample_base, ctx.S, ctx.slot(self)
        else:
            from ... import ops

            ops.refresh_stale_epoch()   # a bare layer call (no bnn.Model around, which does this for its layers)
            base, S, slot = bfr.reserve_samples(1), 1, None
        if slot is None:
            if self._lp_own is None or self._lp_own.shape[0] != S or self._lp_own.device != device:
                self._lp_own = torch.zeros((S, 2), dtype=torch.float64, device=device)
            slot = self._lp_own
        return ctx, base, S, slot

    def _end(self, ctx, slot) -> None:
        if ctx is None:
            bfr.commit_samples(1)
        self._lp_view = slot
        self._lp_dirty = True
