# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.parameters.base

Parameter protocol of the Bayesian layers — same surface as the reference's
/root/reference/bayeformers/nn/parameters/base.py (parameter :17-32, Parameter :35-52, NoneParameter :55-69).
"""
from typing import Optional

import torch
import torch.nn as nn
from torch import Size, Tensor
from torch.nn import Module


def parameter(size: Size, dtype: Optional[torch.dtype] = torch.float32) -> nn.Parameter:
    """Zero fp32 parameter of the given size.  Like the reference (base.py:32) the dtype argument is ignored:
    the variational parameters are fp32 masters whatever precision the matmul runs in."""
    return nn.Parameter(torch.zeros(size, dtype=torch.float32))


class Parameter(Module):
    """Base class of Bayesian parameters: ``sample()`` and ``log_prob(input)`` (base.py:48-52)."""

    def __init__(self) -> None:
        super(Parameter, self).__init__()

    def sample(self) -> Tensor:
        raise NotImplementedError("Sample not implemented yet")

    def log_prob(self, input: Tensor) -> Tensor:
        raise NotImplementedError("Log_prob not implemented yet")

    def _apply(self, fn, recurse=True):
        # Device moves are honoured, precision changes are not: `model.to(torch.bfloat16)` / `.half()` must leave
        # mu/rho (and the prior's constants) in fp32 — the HIP kernels read fp32 masters and choose the MFMA
        # operand precision themselves.
        def keep_fp32(t):
            out = fn(t)
            if t.is_floating_point() and out.dtype != t.dtype:
                out = t.to(device=out.device)
            return out

        return super(Parameter, self)._apply(keep_fp32, recurse)


class NoneParameter(Parameter):
    """Proxy for an absent bias: ``sample() -> None``, ``log_prob() -> 0.0`` (base.py:55-69)."""

    def __init__(self) -> None:
        super(NoneParameter, self).__init__()

    def sample(self) -> Tensor:
        return None

    def log_prob(self, input: Tensor) -> Tensor:
        return 0.0
