# -*- coding: utf-8 -*-
"""bayeformers_amd.nn — `import bayeformers_amd.nn as bnn`.

Exports the names of /root/reference/bayeformers/nn/__init__.py:2-25 (plus the opt-in `Embedding` extension) and the
conversion registry `TORCH2BAYE`, which `to_bayesian` consults by EXACT class, as the reference does.
"""
import torch.nn as _torch_nn

from .layers.embedding import Embedding
from .layers.linear import Linear
from .model import Model, is_module_bayesian
from .parameters.base import NoneParameter, Parameter
from .parameters.gaussian import DEFAULT_SCALED_GAUSSIAN_MIXTURE, Gaussian, ScaledGaussianMixture
from .parameters.initializations import DEFAULT_UNIFORM, Initialization, Uniform

__all__ = ["Linear", "Embedding", "Model", "is_module_bayesian", "Parameter", "NoneParameter", "Gaussian",
           "ScaledGaussianMixture", "DEFAULT_SCALED_GAUSSIAN_MIXTURE", "Initialization", "Uniform", "DEFAULT_UNIFORM",
           "TORCH2BAYE"]

# frequentist class -> Bayesian class exposing `from_frequentist(layer, initialization, prior, delta, freeze)`;
# only nn.Linear by default (bayeformers_amd.enable_embedding() adds nn.Embedding)
TORCH2BAYE = {_torch_nn.Linear: Linear}
