# -*- coding: utf-8 -*-
"""The parameter protocol of the Bayesian layers.

API parity with /root/reference/bayeformers/nn/parameters/base.py: `parameter(size)` (:17-32), the abstract
`Parameter` module with `sample()` / `log_prob(input)` (:35-52) and `NoneParameter`, the stand-in for a missing bias
(:55-69).
"""
from typing import Optional

import torch
from torch import Size, Tensor, nn


def parameter(size: Size, dtype: Optional[torch.dtype] = torch.float32) -> nn.Parameter:
    """A zero-filled float32 nn.Parameter.  `dtype` is accepted and ignored, as in the reference (base.py:32):
    mu and rho are fp32 masters whatever precision the matrix cores run in."""
    del dtype
    return nn.Parameter(torch.zeros(size, dtype=torch.float32))


class Parameter(nn.Module):
    """What a layer needs from a posterior or a prior: a draw and a summed log-density."""

    def sample(self) -> Tensor:
        raise NotImplementedError(f"{type(self).__name__}.sample is not defined")

    def log_prob(self, input: Tensor) -> Tensor:
        raise NotImplementedError(f"{type(self).__name__}.log_prob is not defined")

    def _apply(self, fn, recurse=True):
        """`.cuda()` / `.to(device)` move the tensors; `.to(torch.bfloat16)` / `.half()` do NOT change their
        precision: the HIP kernels read fp32 masters and pick the MFMA operand type themselves."""

        def device_only(t):
            moved = fn(t)
            if t.is_floating_point() and moved.dtype != t.dtype:
                moved = t.to(device=moved.device)
            return moved

        return super()._apply(device_only, recurse)


class NoneParameter(Parameter):
    """Absent parameter (a layer built with bias=False): nothing to draw, contributes 0 to every log-density."""

    def sample(self) -> None:
        return None

    def log_prob(self, input: Tensor) -> float:
        return 0.0
