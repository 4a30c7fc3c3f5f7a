# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.layers.embedding

Bayesian equivalent of torch.nn.Embedding — an EXTENSION of the reference, which converts only nn.Linear
(/root/reference/bayeformers/nn/__init__.py:25) and leaves embedding tables frequentist.  The layer follows the
conventions of the reference's Linear (/root/reference/bayeformers/nn/layers/linear.py:25-165): a Gaussian
posterior over the table, a prior, `from_frequentist(layer, initialization, prior, delta, freeze)` with the same
MOPED rule, and detached `log_prior` / `log_variational_posterior` attributes refreshed by every forward.

Semantics (defined here, parity unpinned — there is no reference implementation to compare with): each Monte-Carlo
sample draws the WHOLE table, W_s = mu + softplus(rho) * eps_s, the forward returns its rows `ids`, and the
log-probs are those of the whole draw — exactly what `F.embedding(ids, weight.sample())` would give.  Only the
gathered rows are materialised (bf_embedding_fwd regenerates their eps from the Philox counter); the log-probs
come from one bf_sample_logprob pass over the table with no sample output.

Opt-in: `bayeformers_amd.enable_embedding()` adds it to TORCH2BAYE; by default `to_bayesian` behaves like the
reference.
"""
from typing import Optional

import torch
from torch import Size, Tensor
from torch.nn import Module

from ... import ops
from ... import random as bfr
from ..parameters.base import NoneParameter, Parameter
from ..parameters.gaussian import DEFAULT_SCALED_GAUSSIAN_MIXTURE, Gaussian
from ..parameters.initializations import DEFAULT_UNIFORM, Initialization
from .base import KernelLayer
from .linear import _moped


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, mu, rho, layer, S, seed, base, out_dtype):
        ctx.layer, ctx.S, ctx.seed, ctx.base = layer, S, seed, base
        ctx.counter = bfr.counter_snapshot()
        ctx.save_for_backward(ids)
        return ops.embedding_forward(ids, mu, rho, out_dtype, S, seed, base, 2 * layer.layer_id)

    @staticmethod
    def backward(ctx, grad):
        (ids,) = ctx.saved_tensors
        layer = ctx.layer
        with bfr.counter_override(ctx.counter):
            dmu, drho = ops.embedding_backward(ids, grad, layer.weight.mu, layer.weight.rho, ctx.S, ctx.seed,
                                               ctx.base, 2 * layer.layer_id, ctx.needs_input_grad[1],
                                               ctx.needs_input_grad[2])
        if layer.padding_idx is not None:  # nn.Embedding: the padding row receives no gradient (mu frozen or not)
            if dmu is not None:
                dmu[layer.padding_idx].zero_()
            if drho is not None:
                drho[layer.padding_idx].zero_()
        return None, dmu, drho, None, None, None, None, None


class Embedding(KernelLayer):
    """Bayesian embedding table: weight ~ N(mu, softplus(rho)) of shape [num_embeddings, embedding_dim].

    Attributes: num_embeddings, embedding_dim, padding_idx, initialization, weight (Gaussian), weight_prior,
        log_prior, log_variational_posterior, layer_id, out_dtype (None = float32; follows `.to(dtype)` / `.half()`
        of the enclosing module, the variational parameters themselves stay float32).
    """

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None,
                 initialization: Optional[Initialization] = DEFAULT_UNIFORM,
                 prior: Optional[Parameter] = DEFAULT_SCALED_GAUSSIAN_MIXTURE) -> None:
        super(Embedding, self).__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx = padding_idx
        self.initialization = initialization
        self.weight = Gaussian(Size((num_embeddings, embedding_dim)), self.initialization)
        self.weight_prior = prior
        self.bias = NoneParameter()        # lets the model-level code treat Linear and Embedding alike
        self.bias_prior = NoneParameter()
        self.out_dtype = None
        self._init_kernel_layer()

    def _apply(self, fn, recurse=True):
        # `model.to(torch.bfloat16)` / `.half()`: the table's mu/rho stay float32 masters (parameters/base.py), the
        # rows this layer emits take the requested precision, as a frequentist nn.Embedding's would
        probe = fn(torch.zeros(1, dtype=torch.float32))
        if probe.dtype.is_floating_point and probe.dtype != torch.float64:
            self.out_dtype = probe.dtype if probe.dtype != torch.float32 else None
        return super(Embedding, self)._apply(fn, recurse)

    def forward(self, input: Tensor) -> Tensor:
        """out[..., :] = W_s[input[...]] for the sample s the token belongs to (input is [S*B, ...], sample-major,
        inside `bnn.Model.monte_carlo(S)`; otherwise S = 1)."""
        if input.dtype not in (torch.int64, torch.int32):
            raise TypeError("bnn.Embedding expects integer token ids")
        if bfr.STATE.ctx is None:
            again = bfr.recompute_context()
            if again is not None:  # a checkpointed block recomputed during backward: the forward's own epsilon
                with again.replay():
                    return self.forward(input)
        ctx, base, S, slot = self._begin(self.weight.mu.device)
        ids = input.reshape(-1).to(torch.int64).contiguous()
        _, lp = ops.sample_logprob([self.weight], [self.weight_prior], [2 * self.layer_id], S, bfr.STATE.seed, base)
        slot.copy_(lp)
        out = _EmbeddingFn.apply(ids, self.weight.mu, self.weight.rho, self, S, bfr.STATE.seed, base,
                                 self.out_dtype or torch.float32)
        self._end(ctx, slot)
        return out.view(*input.shape, self.embedding_dim)

    @classmethod
    def from_frequentist(cls, embedding: Module, initialization: Optional[Initialization] = DEFAULT_UNIFORM,
                         prior: Optional[Parameter] = DEFAULT_SCALED_GAUSSIAN_MIXTURE, delta: float = None,
                         freeze: bool = False) -> "Embedding":
        """Bayesian table from an nn.Embedding, with the MOPED rule of Linear.from_frequentist (linear.py:139-150)
        when `delta` is given.  max_norm / sparse / scale_grad_by_freq tables are not supported."""
        if embedding.max_norm is not None or embedding.sparse or embedding.scale_grad_by_freq:
            raise NotImplementedError("bnn.Embedding: max_norm, sparse and scale_grad_by_freq are not supported")
        baye = cls(embedding.num_embeddings, embedding.embedding_dim, embedding.padding_idx, prior=prior)
        if delta is not None:
            baye.weight_prior = _moped(baye.weight, embedding.weight, delta, freeze)
        baye.out_dtype = embedding.weight.dtype if embedding.weight.dtype != torch.float32 else None
        return baye
