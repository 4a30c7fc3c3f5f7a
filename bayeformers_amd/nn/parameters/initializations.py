# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.parameters.initializations

Initialisation callbacks for (mu, rho) — same surface as
/root/reference/bayeformers/nn/parameters/initializations.py (Initialization :14-22, Uniform :25-56, default :60).
One-off host work, not on the per-step path.
"""
from typing import Tuple

from torch.nn import Parameter

TwoParameters = Tuple[Parameter, Parameter]
Range = Tuple[float, float]


class Initialization:
    """Callback that initialises mu and rho in place and returns them."""

    def __call__(self, mu: Parameter, rho: Parameter) -> TwoParameters:
        raise NotImplementedError("Initialization not implemented yet")


class Uniform(Initialization):
    """mu ~ U(mu_range), rho ~ U(rho_range), drawn from torch's global generator in that order — the same two
    ``uniform_`` calls as the reference (initializations.py:54-55), so a seeded construction reproduces it."""

    def __init__(self, mu_range: Range, rho_range: Range) -> None:
        super(Uniform, self).__init__()
        self.mu_range, self.rho_range = mu_range, rho_range

    def __call__(self, mu: Parameter, rho: Parameter) -> TwoParameters:
        mu.data = mu.data.uniform_(*self.mu_range)
        rho.data = rho.data.uniform_(*self.rho_range)
        return mu, rho


"""Default Uniform initialization (initializations.py:60)"""
DEFAULT_UNIFORM = Uniform((-0.2, 0.2), (-5, -4))
