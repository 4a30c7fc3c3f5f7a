"""model_oracle — TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path.

A CPU module that runs the reference's per-forward op sequence (torch-CPU fp32, torch's own RNG), used only as the
`cpu_baseline` leg of bench.py (kind "port"): the reference itself is Python and does not travel to the GPU box.
"""
import copy

import torch
import torch.nn as nn

from oracle import bayes_oracle as bo


class OracleLinear(nn.Module):
    """Linear.forward of the reference — /root/reference/bayeformers/nn/layers/linear.py:83-104 — as plain torch CPU
    ops in the reference's order: normal -> softplus -> mul/add -> four log-prob reductions -> F.linear."""

    def __init__(self, lin: nn.Linear, delta):
        super().__init__()
        w = lin.weight.data
        self.has_bias = lin.bias is not None
        if delta is None:  # default Uniform init + scale-mixture prior (initializations.py:60, gaussian.py:175-177)
            self.mu_w = torch.empty_like(w).uniform_(-0.2, 0.2)
            self.rho_w = torch.empty_like(w).uniform_(-5, -4)
            self.prior_w = self.prior_b = ("mixture", 0.5, 1.0, float(torch.tensor(-6.0).exp()))
            if self.has_bias:
                self.mu_b = torch.empty_like(lin.bias.data).uniform_(-0.2, 0.2)
                self.rho_b = torch.empty_like(lin.bias.data).uniform_(-5, -4)
        else:              # MOPED (linear.py:139-163)
            self.mu_w, self.rho_w = w, bo.moped_rho(w, delta)
            self.prior_w = ("gaussian", w, torch.ones_like(w))
            if self.has_bias:
                b = lin.bias.data
                self.mu_b, self.rho_b = b, bo.moped_rho(b, delta)
                self.prior_b = ("gaussian", b, torch.ones_like(b))
        if not self.has_bias:
            self.mu_b = self.rho_b = self.prior_b = None
        self.log_prior = torch.tensor(0.0)
        self.log_variational_posterior = torch.tensor(0.0)

    def forward(self, x):
        eps_w = torch.randn(self.mu_w.shape)
        eps_b = torch.randn(self.mu_b.shape) if self.has_bias else None
        y, self.log_prior, self.log_variational_posterior = bo.linear_forward(
            x, self.mu_w, self.rho_w, self.mu_b, self.rho_b, eps_w, eps_b, self.prior_w, self.prior_b)
        return y


def to_oracle(model: nn.Module, delta=None) -> nn.Module:
    """to_bayesian of the reference (bayeformers/__init__.py:50-61) with OracleLinear layers."""
    model = copy.deepcopy(model)

    def swap(m):
        for name, child in m.named_children():
            if child.__class__ is nn.Linear:
                setattr(m, name, OracleLinear(child, delta))
            swap(child)

    swap(model)
    return model


def log_probs(model: nn.Module):
    """Model.log_prior() / log_variational_posterior() — bayeformers/nn/model.py:70-89."""
    lp = lq = 0.0
    for m in model.modules():
        if isinstance(m, OracleLinear):
            lp = lp + m.log_prior
            lq = lq + m.log_variational_posterior
    return lp, lq
