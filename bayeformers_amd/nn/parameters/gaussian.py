# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.parameters.gaussian

Gaussian variational parameter and the scale-mixture prior — same surface and state-dict keys as
/root/reference/bayeformers/nn/parameters/gaussian.py (Gaussian :22-116, ScaledGaussianMixture :119-171,
default :175-177).  Inside bnn.Linear.forward none of the methods below run: the layer hands (mu, rho, prior)
to the fused HIP kernel.  `sample()` is the same kernel for one tensor; `log_prob(input)` of an ARBITRARY input
is protocol surface only (not on the hot path) and is evaluated with device tensor ops in stable form.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Size, Tensor

from ... import random as _bfr
from .base import Parameter, parameter
from .initializations import DEFAULT_UNIFORM, Initialization

LOG_SQRT_2PI = float(np.log(np.sqrt(2 * np.pi)))


def _frozen_scalar(value: float) -> nn.Parameter:
    """0-d fp32 constant kept as a frozen nn.Parameter, so that it appears in the state dict under the reference's
    key and follows the module across devices."""
    return nn.Parameter(torch.tensor(float(value), dtype=torch.float32), requires_grad=False)


class Gaussian(Parameter):
    """Mean-field Gaussian parametrised by mu and rho, sigma = softplus(rho) = log(1 + exp(rho)).

        eps ~ N(0, 1)      (Philox counter, csrc/bf_philox.h)
        W   = mu + eps * sigma

    Attributes / state-dict keys (as the reference): mu, rho, zero, one."""

    def __init__(self, size: Size, initialization: Optional[Initialization] = DEFAULT_UNIFORM,
                 dtype: Optional[torch.dtype] = torch.float32) -> None:
        super(Gaussian, self).__init__()
        self.size = size
        self.dtype = dtype
        self.initialization = initialization
        self.mu, self.rho = parameter(size), parameter(size)
        for name, value in (("zero", 0.0), ("one", 1.0)):  # the reference's N(0, 1) constants: state-dict keys only
            self.register_parameter(name, _frozen_scalar(value))

        # Philox stream of a stand-alone sample(); a bnn.Linear uses 2*layer_id + {0,1} instead.
        self.stream_id = 2 * _bfr.new_layer_id()
        self.reset_parameters()

    def reset_parameters(self) -> None:
        """Reset mu and rho with the initialization callback (gaussian.py:74-79)."""
        self.mu, self.rho = self.initialization(self.mu, self.rho)

    @property
    def sigma(self) -> Tensor:
        """softplus(rho) (gaussian.py:81-88)."""
        return F.softplus(self.rho)

    def sample(self) -> Tensor:
        """One reparameterised draw W = mu + eps * sigma (gaussian.py:90-101) from the fused HIP kernel."""
        from ... import ops, random as bfr
        from .base import NoneParameter

        base = bfr.reserve_samples(1)
        outs, _ = ops.sample_logprob([self], [NoneParameter()], [self.stream_id], 1, bfr.STATE.seed, base,
                                     out_dtype=torch.float32)
        bfr.commit_samples(1)
        return outs[0][0]

    def log_prob(self, input: Tensor) -> Tensor:
        """sum[-log sqrt(2 pi) - log sigma - (input - mu)^2 / (2 sigma^2)]  (gaussian.py:103-116)."""
        sigma = self.sigma
        z = (input - self.mu) / sigma
        return -(LOG_SQRT_2PI + torch.log(sigma) + 0.5 * z * z).sum()


class ScaledGaussianMixture(Parameter):
    """Scale mixture of two zero-mean Gaussians, used only as a prior (gaussian.py:119-171).

    State-dict keys (as the reference): pi, sigma1, sigma2, zero."""

    def __init__(self, pi: float, sigma1: float, sigma2: float) -> None:
        super(ScaledGaussianMixture, self).__init__()
        for name, value in (("pi", pi), ("sigma1", sigma1), ("sigma2", sigma2), ("zero", 0.0)):
            self.register_parameter(name, _frozen_scalar(value))
        self._consts = None

    def constants(self):
        """(pi, sigma1, sigma2) as the fp32 values the parameters hold.  Device-resident parameters are read once and
        cached on the host so that a forward never synchronises on them; the cache follows load_state_dict, `.to()` and
        in-place edits via the version counters and addresses, and an edit through `.data` — which moves neither — is
        caught on the device: every kernel that evaluates the prior compares the cached values with the scalars
        themselves (bf_prior_t.d_pi / d_sigma1 / d_sigma2) and reports a mismatch through the stale counter, after which
        `bayeformers_amd.invalidate_caches` drops this cache.  Host-resident parameters are simply read on every call."""
        if not self.pi.is_cuda:
            return float(self.pi), float(self.sigma1), float(self.sigma2)
        key = (self.pi._version, self.sigma1._version, self.sigma2._version,
               self.pi.data_ptr(), self.sigma1.data_ptr(), self.sigma2.data_ptr(), _bfr.STATE.stale_epoch)
        if self._consts is None or self._consts[0] != key:
            self._consts = (key, (float(self.pi), float(self.sigma1), float(self.sigma2)))
        return self._consts[1]

    def sample(self) -> Tensor:
        """Not implemented in the reference either: returns 0.0 (gaussian.py:152-158)."""
        return 0.0

    def log_prob(self, input: Tensor) -> Tensor:
        """sum log(pi N(x; 0, sigma1) + (1 - pi) N(x; 0, sigma2))  (gaussian.py:160-171), in log-sum-exp form:
        the reference's exp/log form underflows to -inf for |x| >= 14.3 at sigma1 = 1, this one stays finite."""
        t1 = -0.5 * (input / self.sigma1) ** 2 - torch.log(self.sigma1) - LOG_SQRT_2PI + torch.log(self.pi)
        t2 = -0.5 * (input / self.sigma2) ** 2 - torch.log(self.sigma2) - LOG_SQRT_2PI + torch.log1p(-self.pi)
        return torch.logaddexp(t1, t2).sum()


"""Default prior (gaussian.py:175-177): pi = 0.5, sigma1 = e^0, sigma2 = e^-6"""
DEFAULT_SCALED_GAUSSIAN_MIXTURE = ScaledGaussianMixture(0.5, np.exp(-0), np.exp(-6))
