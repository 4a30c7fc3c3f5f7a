# -*- coding: utf-8 -*-
"""Initialisers of a Gaussian's (mu, rho) pair.

API parity with /root/reference/bayeformers/nn/parameters/initializations.py: an `Initialization` is a callable
`(mu, rho) -> (mu, rho)` (:14-22), `Uniform(mu_range, rho_range)` draws both uniformly (:25-56) and
`DEFAULT_UNIFORM` is U(-0.2, 0.2) for mu, U(-5, -4) for rho (:60).  Construction-time host work only; nothing here
is on the per-step path.
"""
from typing import Sequence, Tuple

import torch
from torch.nn import Parameter

Range = Tuple[float, float]
TwoParameters = Tuple[Parameter, Parameter]


class Initialization:
    """Abstract initialiser: subclasses fill mu and rho in place and hand both back."""

    def __call__(self, mu: Parameter, rho: Parameter) -> TwoParameters:
        raise NotImplementedError(f"{type(self).__name__} does not define how to initialise (mu, rho)")


class Uniform(Initialization):
    """mu ~ U(mu_range) first, then rho ~ U(rho_range), both from torch's global generator.

    The draw order and the in-place `uniform_` are what make a seeded construction reproduce the reference's values
    bit for bit (initializations.py:54-55; checked in tests/test_host_api.py)."""

    def __init__(self, mu_range: Sequence[float], rho_range: Sequence[float]) -> None:
        self.mu_range = mu_range
        self.rho_range = rho_range

    def __call__(self, mu: Parameter, rho: Parameter) -> TwoParameters:
        with torch.no_grad():
            for tensor, (low, high) in ((mu, self.mu_range), (rho, self.rho_range)):
                tensor.uniform_(low, high)
        return mu, rho

    def __repr__(self) -> str:
        return f"Uniform(mu_range={tuple(self.mu_range)}, rho_range={tuple(self.rho_range)})"


DEFAULT_UNIFORM = Uniform(mu_range=(-0.2, 0.2), rho_range=(-5, -4))
