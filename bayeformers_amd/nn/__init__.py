# -*- coding: utf-8 -*-
"""bayeformers_amd.nn — the names /root/reference/bayeformers/nn/__init__.py:2-25 exports."""
import torch.nn as nn

from .layers.embedding import Embedding
from .layers.linear import Linear
from .model import Model, is_module_bayesian
from .parameters.base import NoneParameter, Parameter
from .parameters.gaussian import DEFAULT_SCALED_GAUSSIAN_MIXTURE, Gaussian, ScaledGaussianMixture
from .parameters.initializations import DEFAULT_UNIFORM, Initialization, Uniform

"""Available Bayesian Layers (exact-class lookup, as the reference)"""
TORCH2BAYE = {nn.Linear: Linear}
