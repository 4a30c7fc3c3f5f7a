# -*- coding: utf-8 -*-
"""bayeformers_amd.nn.layers.linear

Bayesian equivalent of torch.nn.Linear — same constructor, attributes, state-dict keys and `from_frequentist`
as /root/reference/bayeformers/nn/layers/linear.py (Linear :25-81, forward :83-104, from_frequentist :106-165).

forward() is ONE call into the C-ABI (bf_linear_fwd): the fused HIP kernel draws eps from the Philox counter,
forms W_s = mu + softplus(rho) * eps for all S Monte-Carlo samples, accumulates log_prior and
log_variational_posterior exactly once per scalar, and the MFMA GEMM computes y[s] = x[s] W_s^T + b_s.
"""
from typing import Optional

import torch
import torch.nn as nn
from torch import Size, Tensor
from torch.nn import Module

from ... import ops
from ... import random as bfr
from ..parameters.base import NoneParameter, Parameter
from .base import KernelLayer
from ..parameters.gaussian import DEFAULT_SCALED_GAUSSIAN_MIXTURE, Gaussian
from ..parameters.initializations import DEFAULT_UNIFORM, Initialization



def _note_forward(layer, S, x, cdt, need_grad) -> None:
    """A forward that records gradients tells the layer's deferred-reduction manager (training.DeferredParamGrads), if it has
    one, which shape its backward will have: the manager decides BEFORE the backward pass whether this step defers."""
    d = layer.__dict__.get("_bf_pg_defer")
    if d is not None and need_grad:
        d.note_forward(layer, S, x.shape[0] // S, cdt)


def _backward(ctx, grad):
    """Shared backward of the per-layer and the planned forward: one bf_linear_bwd call."""
    x, *rest = ctx.saved_tensors
    act = getattr(ctx, "act", 0)
    pre = rest[0] if rest else None
    layer, S, seed, base, cdt = ctx.layer, ctx.S, ctx.seed, ctx.base, ctx.cdt
    need_x, need_mu_w, _, need_mu_b, _ = ctx.needs_input_grad[:5]
    w_samples = None
    kept = getattr(ctx, "kept", None)
    if kept is not None:
        plan, gi, token, w_s = kept
        if plan.arena_owner[gi % len(plan.arenas)] == (gi, token):  # no later forward has overwritten the arena
            w_samples = w_s
    with bfr.counter_override(ctx.counter):
        return ops.linear_backward(layer, x, grad, S, seed, base, cdt, need_x, need_mu_w, need_mu_b, w_samples,
                                   act if pre is not None else 0, pre)


class _LinearFn(torch.autograd.Function):
    """Autograd node of the fused forward (bf_linear_fwd).  Backward = bf_linear_bwd: the reference's graph, i.e.
    gradients through F.linear(x, mu + eps*softplus(rho)) with eps regenerated from the Philox counter and the
    log-prob scalars detached (they carry no gradient in the reference, layers/linear.py:99-102)."""

    @staticmethod
    def forward(ctx, x, mu_w, rho_w, mu_b, rho_b, layer, S, seed, base, lp_out, need_grad=True):
        ctx.layer, ctx.S, ctx.seed, ctx.base = layer, S, seed, base
        ctx.counter = bfr.counter_snapshot(need_grad)
        ctx.cdt = layer.compute_dtype or bfr.get_compute_dtype()
        ctx.save_for_backward(x)
        _note_forward(layer, S, x, ctx.cdt, need_grad)
        return ops.linear_forward(layer, x, S, seed, base, lp_out)

    @staticmethod
    def backward(ctx, grad):
        dx, dmu_w, drho_w, dmu_b, drho_b = _backward(ctx, grad)
        return dx, dmu_w, drho_w, dmu_b, drho_b, None, None, None, None, None, None


class _PlannedLinearFn(torch.autograd.Function):
    """Forward of a layer whose weights were sampled by the model's cross-layer plan: only the MFMA GEMM is left."""

    @staticmethod
    def forward(ctx, x, mu_w, rho_w, mu_b, rho_b, w_s, b_s, layer, S, seed, base, act, need_grad=False):
        ctx.layer, ctx.S, ctx.seed, ctx.base = layer, S, seed, base
        fwd = bfr.STATE.ctx
        if fwd is not None and fwd.plan is not None and id(layer) in fwd.plan.group_of:
            ctx.kept = (fwd.plan, fwd.plan.group_of[id(layer)], fwd.token, w_s)  # see _backward
        # need_grad is decided by the caller from torch.is_grad_enabled(): inside forward() grad mode is always off and
        # needs_input_grad reflects requires_grad even under no_grad — an inference step must pay neither the copy of
        # the device counter nor the second store below
        ctx.counter = bfr.counter_snapshot(need_grad)
        ctx.cdt = w_s.dtype
        _note_forward(layer, S, x, ctx.cdt, need_grad)
        # an activation fused into the GEMM while gradients are recorded: the launch also stores the pre-activation,
        # and the backward folds act' (and the bias gradient's column sums) into one pass over the output gradient
        if act and need_grad:
            y, pre = ops.planned_linear_forward(x, w_s, b_s, S, layer.out_features, layer.in_features, act, True)
            ctx.act = act
            ctx.save_for_backward(x, pre)
            return y
        ctx.save_for_backward(x)
        return ops.planned_linear_forward(x, w_s, b_s, S, layer.out_features, layer.in_features, act)

    @staticmethod
    def backward(ctx, grad):
        dx, dmu_w, drho_w, dmu_b, drho_b = _backward(ctx, grad)
        return dx, dmu_w, drho_w, dmu_b, drho_b, None, None, None, None, None, None, None, None


class _StackedLinearFn(torch.autograd.Function):
    """Layers that read the same activations (query / key / value) under autograd: forward = ONE bf_gemm_nt_layers launch
    over their stacked sampled weights, backward = the parameter gradients of every layer (bf_linear_bwd without dx)
    and ONE input gradient dx = sum_l dy_l W_l as a single contraction over the stacked layers (bf_gemm_nn_layers) —
    autograd would otherwise run L input-gradient GEMMs and add their results."""

    @staticmethod
    def forward(ctx, x, w_stack, b_stack, run, S, seed, base, *params):
        L = len(run)
        N, K = run[0].out_features, run[0].in_features
        M = x.shape[0] // S
        ctx.run, ctx.S, ctx.seed, ctx.base, ctx.cdt = run, S, seed, base, w_stack.dtype
        fwd = bfr.STATE.ctx
        ctx.kept = (fwd.plan, fwd.plan.group_of[id(run[0])], fwd.token, w_stack)
        ctx.counter = bfr.counter_snapshot()
        ctx.save_for_backward(x)
        for layer in run:
            _note_forward(layer, S, x, ctx.cdt, True)
        y = ops.gemm_nt_layers(x, w_stack, b_stack, L, S, M, N, K, M * K, x.dtype)
        return tuple(y[l].view(S * M, N) for l in range(L))

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        run, S, seed, base, cdt = ctx.run, ctx.S, ctx.seed, ctx.base, ctx.cdt
        L, N = len(run), run[0].out_features
        M = x.shape[0] // S
        grads = [g if g is not None else torch.zeros((S * M, N), dtype=cdt, device=x.device) for g in grads]
        grads = [(g if g.dtype == cdt else g.to(cdt)) for g in grads]
        grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
        plan, gi, token, w_stack = ctx.kept
        arena_alive = plan.arena_owner[gi % len(plan.arenas)] == (gi, token)  # no later forward overwrote the samples
        need = ctx.needs_input_grad
        out = [None] * (7 + 4 * L)
        with bfr.counter_override(ctx.counter):
            dx = None
            if need[0] and arena_alive:
                # the L output gradients as one [L, S, M, N] tensor: in place when they are consecutive slabs of one
                # buffer (what ops.attention_backward returns), stacked otherwise
                g0 = grads[0]
                slabs = all(g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr()
                            and g.storage_offset() == g0.storage_offset() + l * S * M * N for l, g in enumerate(grads))
                dy = torch.as_strided(g0, (L, S, M, N), (S * M * N, M * N, N, 1)) if slabs else torch.stack(grads).view(L, S, M, N)
                dx = ops.gemm_nn_layers(dy, w_stack).view(S * M, -1)
            for l, layer in enumerate(run):
                o = 7 + 4 * l
                lone = need[0] and not arena_alive  # samples gone: this layer's own dx from regenerated weights
                r = ops.linear_backward(layer, x, grads[l], S, seed, base, cdt, lone, need[o], need[o + 2])
                if lone:
                    dx = r[0] if dx is None else dx + r[0]
                out[o], out[o + 1], out[o + 2], out[o + 3] = r[1], r[2], r[3], r[4]
        if dx is not None and dx.dtype != x.dtype:
            dx = dx.to(x.dtype)
        out[0] = dx
        return tuple(out)


class _FFNPairFn(torch.autograd.Function):
    """dense -> exact GELU -> dense of a transformer feed-forward pair (HF BertIntermediate + BertOutput.dense) as ONE autograd
    node.  Forward: the two planned GEMMs as they run anyway (GELU in the first one's epilogue, pre-activation kept).  Backward:
    the gradient that reaches the pre-activation, (dy W_down) o gelu'(pre), comes out of the down layer's input-gradient GEMM
    itself (bf_gemm_nn_actgrad) — autograd's two nodes need a pass of their own over the [S*M, 3072] gradient for it
    (gelu_bwd_colsum: 130 us per layer of a BERT-base step).  The intermediate never leaves this node, so it has exactly one
    consumer by construction.  Everything else is the two layers' ordinary backward (ops.linear_backward)."""

    @staticmethod
    def forward(ctx, x, up, down, S, seed, base, w1, b1, w2, b2, *params):
        ctx.up, ctx.down, ctx.S, ctx.seed, ctx.base, ctx.cdt = up, down, S, seed, base, w1.dtype
        fwd = bfr.STATE.ctx
        ctx.kept = tuple((fwd.plan, fwd.plan.group_of[id(l)], fwd.token, w) for l, w in ((up, w1), (down, w2)))
        ctx.counter = bfr.counter_snapshot(True)
        _note_forward(up, S, x, ctx.cdt, True)
        _note_forward(down, S, x, ctx.cdt, True)   # (same rows per sample: the intermediate is [S*M, N1])
        h, pre = ops.planned_linear_forward(x, w1, b1, S, up.out_features, up.in_features, 1, True)
        y = ops.planned_linear_forward(h, w2, b2, S, down.out_features, down.in_features, 0)
        ctx.save_for_backward(x, pre, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, h = ctx.saved_tensors
        up, down, S, seed, base, cdt = ctx.up, ctx.down, ctx.S, ctx.seed, ctx.base, ctx.cdt
        M = x.shape[0] // S
        alive = [plan.arena_owner[gi % len(plan.arenas)] == (gi, token) for plan, gi, token, _ in ctx.kept]
        w1 = ctx.kept[0][3] if alive[0] else None
        w2 = ctx.kept[1][3] if alive[1] else None
        need = ctx.needs_input_grad   # x, then (after the 9 non-tensor arguments) mu_w, rho_w, mu_b, rho_b of up and of down
        dy = (dy if dy.dtype == cdt else dy.to(cdt)).contiguous()
        with bfr.counter_override(ctx.counter):
            dyv = dy.view(S, M, down.out_features)
            fused = w2 is not None and ops.gemm_nn_actgrad_supported(dyv, w2, pre)
            # the down layer: its parameter gradients (and, without the fused form, its input gradient)
            r2 = ops.linear_backward(down, h, dy, S, seed, base, cdt, not fused, need[14], need[16], w2)
            if fused:
                dpre = ops.gemm_nn_actgrad(dyv, w2, pre).view(S * M, -1)
                r1 = ops.linear_backward(up, x, dpre, S, seed, base, cdt, need[0], need[10], need[12], w1)
            else:
                r1 = ops.linear_backward(up, x, r2[0], S, seed, base, cdt, need[0], need[10], need[12], w1, 1, pre)
        dx = r1[0]
        return (dx, None, None, None, None, None, None, None, None, None) + tuple(r1[1:]) + tuple(r2[1:])


class Linear(KernelLayer):
    """Bayesian Linear layer with Gaussian weight/bias posteriors and a prior per parameter.

    Attributes (as the reference): in_features, out_features, initialization, weight, weight_prior, bias,
    bias_prior, log_prior, log_variational_posterior.  Additional: layer_id (Philox stream), compute_dtype
    (None = bayeformers_amd.get_compute_dtype()), log_prob_samples ([S, 2] float64 of the last forward).
    """

    def __init__(self, in_features: int, out_features: int, bias: Optional[bool] = True,
                 initialization: Optional[Initialization] = DEFAULT_UNIFORM,
                 prior: Optional[Parameter] = DEFAULT_SCALED_GAUSSIAN_MIXTURE) -> None:
        super(Linear, self).__init__()
        self.in_features, self.out_features = in_features, out_features
        self.initialization = initialization

        size = Size((self.out_features, self.in_features))
        self.weight = Gaussian(size, self.initialization)
        self.weight_prior = prior

        if bias:
            size = Size((self.out_features,))
            self.bias = Gaussian(size, self.initialization)
            self.bias_prior = prior
        else:
            self.bias = NoneParameter()
            self.bias_prior = NoneParameter()

        self._init_kernel_layer()
        self._small_m = False   # seen with <= 64 rows per sample: single fused kernel, left out of the sampling plan
        self.activation = None  # "gelu": exact GELU fused into the GEMM epilogue (bayeformers_amd.fuse_activations)
        self._shared_input = None  # tuple of layers fed the same activations (bayeformers_amd.fuse_shared_inputs)
        self._plan = ops.LinearPlan()

    def forward(self, input: Tensor) -> Tensor:
        """y = x W^T + b with W ~ N(mu_w, softplus(rho_w)), b ~ N(mu_b, softplus(rho_b))  (linear.py:83-104).

        Inside `bnn.Model` with S Monte-Carlo samples in flight the input is [S*B, ..., in_features]
        (sample-major) and slab s is multiplied by W_s; otherwise S = 1 and one fresh sample index is used."""
        if bfr.STATE.ctx is None:
            again = bfr.recompute_context()
            if again is not None:  # a checkpointed block recomputed during backward: the forward's own epsilon
                with again.replay():
                    return self.forward(input)
        ctx, base, S, slot = self._begin(input.device)
        x2 = input.reshape(-1, self.in_features)
        if x2.shape[0] == 0:
            # empty batch: like the reference, the weights are still sampled and the log-probs refreshed
            gs = [self.weight] + ([self.bias] if isinstance(self.bias, Gaussian) else [])
            prs = [self.weight_prior] + ([self.bias_prior] if isinstance(self.bias, Gaussian) else [])
            sids = [2 * self.layer_id + i for i in range(len(gs))]
            _, lp = ops.sample_logprob(gs, prs, sids, S, bfr.STATE.seed, base)
            slot.copy_(lp)
            self._end(ctx, slot)
            return input.new_empty(*input.shape[:-1], self.out_features)
        mu_b = self.bias.mu if isinstance(self.bias, Gaussian) else None
        rho_b = self.bias.rho if isinstance(self.bias, Gaussian) else None
        want_act = self.activation == "gelu"
        need_grad = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (x2, self.weight.mu, self.weight.rho, mu_b, rho_b))
        rows_per_sample = x2.shape[0] // S
        small = rows_per_sample <= ops.fused_small_rows(self.out_features, self.in_features) and self.in_features % 32 == 0 and x2.dtype != torch.float64
        if small != self._small_m:
            self._small_m = small  # the model rebuilds its sampling plan without / with this layer next forward
        if ctx is not None and ctx.plan is not None and not small and id(self) in ctx.plan.group_of:
            w_s, b_s = ctx.plan.ensure(self, ctx.token, bfr.STATE.seed, base, ctx.lp_buf)
            if self._shared_input is not None and not want_act:
                y = self._stacked_forward(ctx, x2, S, base, need_grad)
                if y is not None:
                    self._lp_view, self._lp_dirty = slot, True
                    return y.view(*input.shape[:-1], self.out_features)
            # with gradients the fused launch also stores the pre-activation (16-bit outputs, N % 8 == 0: what the fused
            # backward pass takes); otherwise the activation runs as a separate autograd op
            fused = want_act and (not need_grad or (x2.dtype != torch.float32 and self.out_features % 8 == 0
                                                    and w_s.dtype == x2.dtype))
            y = _PlannedLinearFn.apply(x2, self.weight.mu, self.weight.rho, mu_b, rho_b, w_s, b_s, self, S,
                                       bfr.STATE.seed, base, 1 if fused else 0, need_grad)
            if want_act and not fused:
                y = torch.nn.functional.gelu(y)
            self._lp_view, self._lp_dirty = slot, True
            return y.view(*input.shape[:-1], self.out_features)
        y = _LinearFn.apply(x2, self.weight.mu, self.weight.rho, mu_b, rho_b, self, S, bfr.STATE.seed, base, slot,
                            need_grad)
        if want_act:
            y = torch.nn.functional.gelu(y)
        self._end(ctx, slot)
        return y.view(*input.shape[:-1], self.out_features)

    def _stacked_forward(self, ctx, x2: Tensor, S: int, base: int = 0, need_grad: bool = False) -> Optional[Tensor]:
        """Layers that read the same activations (query / key / value): whichever of them runs first multiplies x by
        the stacked sampled weights of all of them in ONE launch (bf_gemm_nt_layers); the others pick their slab up.
        With gradients the launch sits in _StackedLinearFn, whose backward computes ONE input gradient for the run.
        Returns None when the run is not stacked in this plan or the input is not the one the outputs came from."""
        run = self._shared_input
        entry = ctx.plan.stacked.get(id(run[0]))
        if entry is None or entry[0] != run or ctx.plan.group_of[id(run[0])] != ctx.plan.group_of[id(self)]:
            return None
        ident = (x2.data_ptr(), tuple(x2.shape), x2.dtype, x2._version, need_grad)
        cached = ctx.shared_out.get(id(run[0]))
        if cached is None or cached[0] != ident:
            if x2.dtype != entry[1].dtype or not x2.is_contiguous():
                return None
            M = x2.shape[0] // S
            if need_grad:
                # what bf_gemm_nn_layers takes; other shapes keep the per-layer autograd path
                if (x2.dtype == torch.float32 or self.out_features % 64 or self.in_features % 8
                        or M * self.in_features < 128 * 128 or len(run) > 4
                        or any(isinstance(l.bias, NoneParameter) for l in run)):
                    return None
                params = [t for l in run for t in (l.weight.mu, l.weight.rho, l.bias.mu, l.bias.rho)]
                y = _StackedLinearFn.apply(x2, entry[1], entry[2], run, S, bfr.STATE.seed, base, *params)
            else:
                y = ops.gemm_nt_layers(x2, entry[1], entry[2], len(run), S, M, self.out_features, self.in_features,
                                       M * self.in_features, x2.dtype)
            # the input tensor is kept with the outputs: its storage cannot be recycled (same address, shape and
            # version) for another activation while the entry is alive, e.g. when a block runs twice in one forward
            cached = (ident, y, x2)
            ctx.shared_out[id(run[0])] = cached
        return cached[1][run.index(self)].view(-1, self.out_features)

    @classmethod
    def from_frequentist(cls, linear: Module, initialization: Optional[Initialization] = DEFAULT_UNIFORM,
                         prior: Optional[Parameter] = DEFAULT_SCALED_GAUSSIAN_MIXTURE, delta: float = None,
                         freeze: bool = False) -> "Linear":
        """Bayesian layer from a frequentist nn.Linear (linear.py:106-165).

        With `delta` (MOPED, Krishnan et al. 2020): mu <- w (shares storage), rho <- log(exp(delta |w|) - 1) with
        -inf -> 0.0, mu frozen if `freeze`, prior <- Gaussian(mu = w, rho = 1), same for the bias.  The fp32
        expressions are the reference's, so the resulting rho is bit-identical.  As in the reference
        (linear.py:137) `initialization` is not forwarded to the constructor."""
        baye = cls(linear.in_features, linear.out_features, linear.bias is not None, prior=prior)
        if delta is not None:
            baye.weight_prior = _moped(baye.weight, linear.weight, delta, freeze)
            if linear.bias is not None:
                baye.bias_prior = _moped(baye.bias, linear.bias, delta, freeze)
        return baye


def _moped(posterior: Gaussian, source: Tensor, delta: float, freeze: bool) -> Gaussian:
    """MOPED initialisation of one parameter from its pretrained value, returning its empirical-Bayes prior
    (linear.py:139-150 for the weight, :152-163 for the bias): the posterior mean takes over the pretrained
    tensor's storage, sigma = delta * |w| is stored as rho = log(exp(delta |w|) - 1) (fp32; where that expression
    gives -inf, i.e. delta |w| < ~6e-8, rho is set to 0), and the prior is N(w, softplus(1))."""
    pretrained = source.data
    rho = torch.log(torch.exp(delta * torch.abs(pretrained)) - 1.0)
    rho[rho == float("-inf")] = 0.0
    posterior.mu.data = pretrained
    posterior.rho.data = rho
    posterior.mu.requires_grad = not freeze

    prior = Gaussian(posterior.mu.size())  # draws its Uniform init like the reference, then is overwritten
    prior.mu.data = pretrained
    prior.rho.data = torch.ones_like(source)
    return prior
