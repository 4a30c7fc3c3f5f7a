# -*- coding: utf-8 -*-
"""bayeformers_amd — MI355X-native Monte-Carlo variational forward path with the BayeFormers API.

    from bayeformers_amd import to_bayesian
    import bayeformers_amd.nn as bnn

`to_bayesian` mirrors /root/reference/bayeformers/__init__.py:19-63.
"""
import types
from copy import deepcopy
from typing import Optional

import torch.nn

from . import nn  # noqa: F401  (bayeformers_amd.nn)
from .nn import TORCH2BAYE
from . import random as bfr
from .nn.model import Model
from .nn.parameters.base import Parameter
from .nn.parameters.gaussian import DEFAULT_SCALED_GAUSSIAN_MIXTURE
from .nn.parameters.initializations import DEFAULT_UNIFORM, Initialization
from .ops import invalidate_caches  # noqa: F401
from .random import (get_compute_dtype, manual_seed, set_compute_dtype, set_kl_gradient,  # noqa: F401
                     use_device_counter)

__all__ = ["to_bayesian", "invalidate_caches", "fuse_activations", "fuse_residual_layernorm", "fuse_shared_inputs", "fuse_ffn_pairs", "fuse_attention", "fuse_embeddings", "enable_embedding", "nn", "manual_seed", "set_compute_dtype", "get_compute_dtype",
           "use_device_counter", "set_kl_gradient"]


def to_bayesian(model: torch.nn.Module, initialization: Optional[Initialization] = DEFAULT_UNIFORM,
                prior: Optional[Parameter] = DEFAULT_SCALED_GAUSSIAN_MIXTURE, delta: float = None,
                freeze: bool = False) -> Model:
    """Deep-copy `model`, swap every layer whose exact class is in TORCH2BAYE for its Bayesian equivalent
    (`from_frequentist(layer, initialization, prior, delta, freeze)`, children visited in `named_children` order,
    depth first) and wrap the result in `bnn.Model`.

    Arguments / keyword arguments are the reference's: `delta` not None selects MOPED initialisation from the
    pretrained weights (Krishnan et al., arXiv:1906.05323), `freeze` freezes the posterior means."""

    def swap(module):
        for name, child in module.named_children():
            bayesian_cls = TORCH2BAYE.get(child.__class__)
            if bayesian_cls is not None:
                setattr(module, name, bayesian_cls.from_frequentist(child, initialization, prior, delta, freeze))
            swap(child)

    new_model = deepcopy(model)
    swap(new_model)
    return Model(model=new_model)


def enable_embedding(enable: bool = True) -> None:
    """Opt in to converting torch.nn.Embedding tables too (bnn.Embedding — an extension, the reference's TORCH2BAYE
    holds only nn.Linear).  Off by default so that `to_bayesian` converts exactly what the reference converts."""
    if enable:
        TORCH2BAYE[torch.nn.Embedding] = nn.Embedding
    else:
        TORCH2BAYE.pop(torch.nn.Embedding, None)


class _FusedIntoDense(torch.nn.Module):
    """Placeholder left where an activation was folded into the preceding Bayesian layer's GEMM epilogue."""

    def forward(self, x):
        return x


def _is_exact_gelu(fn) -> bool:
    if isinstance(fn, torch.nn.GELU):
        return fn.approximate == "none"
    if fn is torch.nn.functional.gelu:
        return True
    # transformers.activations.GELUActivation (ACT2FN["gelu"]): .act is torch.nn.functional.gelu
    return fn.__class__.__name__ == "GELUActivation" and getattr(fn, "act", None) is torch.nn.functional.gelu


def fuse_activations(model: torch.nn.Module) -> int:
    """Fold `dense -> exact GELU` pairs into the dense layer's GEMM epilogue (bf_gemm_nt_act).

    Recognises the HuggingFace pattern `module.dense` (a bnn.Linear) followed by `module.intermediate_act_fn`
    (BertIntermediate and its relatives).  Forward-only optimisation: whenever a gradient may be needed the layer
    falls back to a separate GELU so autograd stays intact.  Returns the number of fused pairs."""
    fused = 0
    for m in model.modules():
        dense = getattr(m, "dense", None)
        act = getattr(m, "intermediate_act_fn", None)
        if isinstance(dense, nn.Linear) and act is not None and _is_exact_gelu(act):
            dense.activation = "gelu"
            m.intermediate_act_fn = _FusedIntoDense().train(m.training)  # (a fresh module would be in training mode)
            fused += 1
    return fused


def _dense_residual_norm_forward(self, hidden_states, input_tensor):
    """forward of an HF `*Output` block — LayerNorm(dropout(dense(h)) + input) — with the residual add and the
    normalisation done by one HBM pass (bf_add_layernorm) behind the Bayesian dense layer's GEMM; with gradients
    enabled the same kernel runs inside an autograd function whose backward is bf_add_layernorm_bwd."""
    return _residual_norm(self, self.dense(hidden_states), input_tensor)


def _residual_norm(self, hidden_states, input_tensor):
    """The part of an HF `*Output` block behind its dense layer: LayerNorm(dropout(h) + input)."""
    from . import ops

    ln = self.LayerNorm
    dropping = self.dropout.training and self.dropout.p > 0  # (the dropout module's own mode, as its forward reads it)
    if not ops.layernorm_supported(hidden_states, input_tensor, ln):
        return ln((self.dropout(hidden_states) if dropping else hidden_states) + input_tensor)
    # training mode (/root/reference/examples/bert_glue.py:221): the hidden dropout runs INSIDE the kernel, its Philox mask
    # regenerated in the backward pass — no mask tensor, no extra pass over the dense output
    drop = ops.Dropout(self.dropout.p, bfr.STATE.seed, bfr.dropout_call(), bfr.dropout_site(self), bfr.dropout_origin(),
                       bfr.dropout_counter()) if dropping else None
    if torch.is_grad_enabled() and (hidden_states.requires_grad or input_tensor.requires_grad or ln.weight.requires_grad):
        if ops._NO_TWIN:
            return ops.AddLayerNormFn.apply(hidden_states, input_tensor, ln.weight, ln.bias, ln.eps, drop)
        # The output of such a block has two consumers in a transformer layer — the next dense layer and the next block's
        # residual connection.  The block hands out `y` with its second alias attached; the next block of this kind takes
        # the alias as its residual input, so the two gradients reach this block's backward separately and are added
        # inside its kernel (AddLayerNormFn, twin).  A consumer that does not know about the alias just uses `y`.
        residual = getattr(input_tensor, "_bf_twin", None)
        y, y2 = ops.AddLayerNormFn.apply(hidden_states, residual if residual is not None else input_tensor, ln.weight, ln.bias,
                                         ln.eps, drop, True)
        y._bf_twin = y2
        return y
    return ops.add_layernorm(hidden_states, input_tensor, ln.weight, ln.bias, ln.eps, drop)


def _layernorm_forward(self, input):
    """nn.LayerNorm.forward on bf_add_layernorm (no residual), differentiable through bf_add_layernorm_bwd."""
    from . import ops

    if input.shape[-1] != self.normalized_shape[0] or not ops.layernorm_supported(input, None, self):
        return torch.nn.functional.layer_norm(input, self.normalized_shape, self.weight, self.bias, self.eps)
    if torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
        return ops.AddLayerNormFn.apply(input, None, self.weight, self.bias, self.eps)
    return ops.add_layernorm(input, None, self.weight, self.bias, self.eps)


def _ffn_pair_chunk(self, attention_output):
    """`feed_forward_chunk` of an HF transformer layer — output(intermediate(a), a) — with the two Bayesian dense layers and
    the GELU between them as ONE autograd node (nn.layers.linear._FFNPairFn) when a gradient is recorded and both layers were
    sampled by the model's cross-layer plan; anything else (inference, layers outside the plan, another activation) runs
    the module's own method."""
    from . import ops
    from .nn.layers.linear import _FFNPairFn
    from .nn.parameters.gaussian import Gaussian

    up, down = self.intermediate.dense, self.output.dense
    ctx = bfr.STATE.ctx
    a = attention_output
    plan = ctx.plan if ctx is not None else None
    usable = (plan is not None and torch.is_grad_enabled() and id(up) in plan.group_of and id(down) in plan.group_of
              and not up._small_m and not down._small_m and up.activation == "gelu" and down.activation is None
              and isinstance(self.intermediate.intermediate_act_fn, _FusedIntoDense)
              and up._shared_input is None and down._shared_input is None
              and isinstance(up.bias, Gaussian) and isinstance(down.bias, Gaussian)
              and a.is_cuda and a.dtype in (torch.bfloat16, torch.float16) and a.shape[-1] == up.in_features
              and up.out_features % 8 == 0 and (a.numel() // up.in_features) % ctx.S == 0
              and (a.numel() // up.in_features) // ctx.S > 128
              and (a.requires_grad or any(p.requires_grad for l in (up, down) for p in l.parameters())))
    if not usable:
        return self._bf_plain_ffn_chunk(attention_output)
    S, seed, base = ctx.S, bfr.STATE.seed, ctx.sample_base
    w1, b1 = plan.ensure(up, ctx.token, seed, base, ctx.lp_buf)
    w2, b2 = plan.ensure(down, ctx.token, seed, base, ctx.lp_buf)
    if w1.dtype != a.dtype or w2.dtype != a.dtype:
        return self._bf_plain_ffn_chunk(attention_output)
    x2 = a.reshape(-1, up.in_features)
    params = [t for l in (up, down) for t in (l.weight.mu, l.weight.rho, l.bias.mu, l.bias.rho)]
    y = _FFNPairFn.apply(x2 if x2.is_contiguous() else x2.contiguous(), up, down, S, seed, base, w1, b1, w2, b2, *params)
    for l in (up, down):  # what Linear.forward leaves behind: the log-probs of this forward are the plan's
        l._lp_view, l._lp_dirty = ctx.slot(l), True
    y = y.view(*a.shape[:-1], down.out_features)
    out = self.output
    if getattr(out.LayerNorm, "_bf_fused", False) and isinstance(out.forward, types.MethodType) and out.forward.__func__ is _dense_residual_norm_forward:
        return _residual_norm(out, y, a)
    return out.LayerNorm(out.dropout(y) + a)


def fuse_ffn_pairs(model: torch.nn.Module) -> int:
    """Make the feed-forward pair of every HuggingFace-style transformer layer (modules with `intermediate.dense` and
    `output.dense` as bnn.Linear children and a `feed_forward_chunk` method) ONE autograd node in training: the GELU's
    backward then rides in the epilogue of the down-projection's input-gradient GEMM (bf_gemm_nn_actgrad) instead of a pass
    of its own over the intermediate-sized gradient.  Call after `fuse_activations`.  Training-time rewrite; inference is
    untouched.  Returns the number of layers rewritten."""
    fused = 0
    for m in model.modules():
        inter, out = getattr(m, "intermediate", None), getattr(m, "output", None)
        if (inter is not None and out is not None and isinstance(getattr(inter, "dense", None), nn.Linear)
                and isinstance(getattr(out, "dense", None), nn.Linear) and hasattr(m, "feed_forward_chunk")
                and isinstance(getattr(out, "LayerNorm", None), torch.nn.LayerNorm) and hasattr(out, "dropout")
                and inter.dense.out_features == out.dense.in_features and not hasattr(m, "_bf_plain_ffn_chunk")):
            m._bf_plain_ffn_chunk = m.feed_forward_chunk
            m.feed_forward_chunk = types.MethodType(_ffn_pair_chunk, m)
            fused += 1
    return fused


def fuse_residual_layernorm(model: torch.nn.Module) -> int:
    """Fuse `LayerNorm(dropout(dense(h)) + input)` blocks whose dense layer is a bnn.Linear (HF BertSelfOutput,
    BertOutput and their relatives: attributes `dense`, `dropout`, `LayerNorm`, forward(hidden_states,
    input_tensor)) into dense GEMM -> one add+LayerNorm pass.  Inference-time optimisation, like fuse_activations.
    Returns the number of fused blocks."""
    fused = 0
    for m in model.modules():
        dense, ln, drop = getattr(m, "dense", None), getattr(m, "LayerNorm", None), getattr(m, "dropout", None)
        if (isinstance(dense, nn.Linear) and isinstance(ln, torch.nn.LayerNorm) and isinstance(drop, torch.nn.Dropout)
                and ln.elementwise_affine and ln.bias is not None and len(ln.normalized_shape) == 1
                and ln.normalized_shape[0] == dense.out_features and dense.out_features % 8 == 0
                and dense.out_features <= 4096 and m.__class__.__name__.endswith("Output")):
            m.forward = types.MethodType(_dense_residual_norm_forward, m)
            ln._bf_fused = True
            fused += 1
    # the remaining stand-alone LayerNorms (the embedding block's) run on the same kernel without a residual
    for m in model.modules():
        if (isinstance(m, torch.nn.LayerNorm) and not getattr(m, "_bf_fused", False) and m.elementwise_affine
                and m.bias is not None and len(m.normalized_shape) == 1 and m.normalized_shape[0] % 8 == 0
                and m.normalized_shape[0] <= 4096):
            m.forward = types.MethodType(_layernorm_forward, m)
            m._bf_fused = True
    return fused


def _embeddings_forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None,
                        past_key_values_length: int = 0):
    """forward of an HF `*Embeddings` block — LayerNorm(word[ids] + type[type_ids] + pos[pos_ids]) — as one pass
    (bf_embed_layernorm) when it is the plain inference case; anything else runs the module's own forward."""
    from . import ops

    w, t, p, ln = self.word_embeddings, self.token_type_embeddings, self.position_embeddings, self.LayerNorm
    dropping = self.dropout.training and self.dropout.p > 0
    plain = (input_ids is not None and inputs_embeds is None and input_ids.dim() == 2 and input_ids.is_cuda
             and input_ids.dtype == torch.long
             and not (torch.is_grad_enabled() and (w.weight.requires_grad or t.weight.requires_grad or
                                                   p.weight.requires_grad or ln.weight.requires_grad))
             and w.weight.dtype == t.weight.dtype == p.weight.dtype and w.weight.dtype in ops._TORCH2BF
             and ln.weight.dtype in (torch.float32, w.weight.dtype) and ln.bias is not None
             and ln.bias.dtype == ln.weight.dtype and w.weight.shape[1] % 8 == 0 and w.weight.shape[1] <= 4096
             and past_key_values_length + input_ids.shape[1] <= p.weight.shape[0]
             and (token_type_ids is None or token_type_ids.dtype == torch.long)
             and (position_ids is None or (position_ids.dtype == torch.long and position_ids.dim() == 2
                                           and position_ids.shape[1] == input_ids.shape[1]
                                           and position_ids.shape[0] in (1, input_ids.shape[0]))))
    if not plain:
        # training: the module's own forward — but on ONE copy of the batch when the ids are sample_bayesian's S-fold
        # repeat of it (the block is deterministic without dropout, every copy gets the same rows): the table gradients
        # are then scattered from B*T rows instead of S*B*T, after one sum over the copies
        rep = getattr(input_ids, "_bf_repeat", None) if input_ids is not None else None
        orig = rep[1]() if rep is not None else None  # (samples, weak reference to the tensor the ids repeat)
        if orig is not None and inputs_embeds is None:
            S = rep[0]
            B = orig.shape[0]

            def one_copy(t):  # a per-row companion of the ids: None, its own original, or rows that cannot differ
                if t is None or t.shape[0] == 1:
                    return t, True
                r = getattr(t, "_bf_repeat", None)
                src = r[1]() if r is not None else None
                if src is not None and r[0] == S and src.shape[0] == B:
                    return src, True
                if t.shape[0] == S * B and t.stride(0) == 0:
                    return t[:B], True
                return None, False

            tt, ok1 = one_copy(token_type_ids)
            pp, ok2 = one_copy(position_ids)
            if ok1 and ok2 and orig.dim() == 2 and orig.shape[0] * S == input_ids.shape[0]:
                # training mode: the block up to its LayerNorm is still the same for every copy — only the dropout that
                # ends it draws a mask per copy, so it moves behind the repeat
                saved = self.dropout
                if dropping:
                    self.dropout = torch.nn.Identity()
                try:
                    e = self._bf_plain_forward(input_ids=orig, token_type_ids=tt, position_ids=pp, inputs_embeds=None,
                                               past_key_values_length=past_key_values_length)
                finally:
                    self.dropout = saved
                e = e.repeat(S, *([1] * (e.dim() - 1)))
                return torch.nn.functional.dropout(e, saved.p, training=True) if dropping else e
        return self._bf_plain_forward(input_ids=input_ids, token_type_ids=token_type_ids, position_ids=position_ids,
                                      inputs_embeds=inputs_embeds, past_key_values_length=past_key_values_length)
    if position_ids is None and past_key_values_length:
        position_ids = self.position_ids[:, past_key_values_length:input_ids.shape[1] + past_key_values_length]
    e = ops.embed_layernorm(input_ids, token_type_ids, position_ids, w.weight, t.weight, p.weight, ln.weight, ln.bias,
                            ln.eps)
    return torch.nn.functional.dropout(e, self.dropout.p, training=True) if dropping else e


def fuse_embeddings(model: torch.nn.Module) -> int:
    """Run embedding blocks of the HuggingFace BERT family (modules holding `word_embeddings`,
    `token_type_embeddings`, `position_embeddings`, `LayerNorm`, `dropout`) as ONE launch: three table gathers, two
    full-size adds and the LayerNorm of their result become bf_embed_layernorm.  Inference-time rewrite like the other
    fuse_* functions: with dropout active, gradients needed or unusual arguments the module's own forward runs.
    Only blocks whose default positions are 0 .. L-1 (BERT, ELECTRA, ...) are rewritten; the RoBERTa family, which derives
    position ids from the padding mask, keeps its own forward.  Returns the number of blocks rewritten."""
    fused = 0
    for m in model.modules():
        parts = [getattr(m, n, None) for n in ("word_embeddings", "token_type_embeddings", "position_embeddings")]
        ln, drop = getattr(m, "LayerNorm", None), getattr(m, "dropout", None)
        if (all(isinstance(e, torch.nn.Embedding) for e in parts) and isinstance(ln, torch.nn.LayerNorm)
                and isinstance(drop, torch.nn.Dropout) and ln.elementwise_affine and ln.bias is not None
                and getattr(m, "position_embedding_type", "absolute") == "absolute"
                # RoBERTa-style blocks (XLM-R, CamemBERT, ...) number positions from padding_idx + 1 and skip padding
                # tokens (create_position_ids_from_input_ids): not the arange positions the kernel assumes — left alone
                and not hasattr(m, "padding_idx") and not hasattr(m, "create_position_ids_from_input_ids")
                and not hasattr(m, "_bf_plain_forward")):
            m._bf_plain_forward = m.forward
            m.forward = types.MethodType(_embeddings_forward, m)
            fused += 1
    return fused


def fuse_shared_inputs(model: torch.nn.Module, names=("query", "key", "value")) -> int:
    """Multiply the activations of an attention block by its query / key / value weights in ONE launch
    (bf_gemm_nt_layers): marks modules that hold `names` as bnn.Linear children of one shape (HF BertSelfAttention
    and its relatives, which call them on the same hidden states).  The sampling plan then lays their sampled
    weights out back to back, and whichever of the layers runs first computes all outputs; a layer that is handed a
    different input simply runs on its own.  With gradients enabled the launch sits in one autograd node whose backward
    computes ONE input gradient for the three layers (bf_gemm_nn_layers).  Returns the number of fused blocks."""
    fused = 0
    for m in model.modules():
        group = tuple(getattr(m, n, None) for n in names)
        if all(isinstance(l, nn.Linear) for l in group) and len({(l.in_features, l.out_features) for l in group}) == 1:
            for l in group:
                l._shared_input = group
            fused += 1
    if isinstance(model, Model):
        model.refresh()
    return fused


_ATTENTION_NAME = "bayeformers_amd"


def _attention_interface(module, query, key, value, attention_mask, dropout: float = 0.0, scaling=None, **kwargs):
    """Attention function in the HuggingFace `AttentionInterface` convention: query/key/value [B, H, T, D], returns
    ([B, T, H, D], None).  Runs bf_attention_fwd (with bf_attention_bwd as its backward when a gradient is needed) when
    it applies (head size 64, T a multiple of 128, no mask or a key-padding mask; attention_probs_dropout runs inside the
    kernels on the Philox keep-mask of csrc/bf_philox.h); anything else goes to the
    framework's scaled-dot-product attention."""
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    from . import ops

    need_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    # attention_probs_dropout (training mode) runs inside the kernels, forward and backward, for any supported length
    usable = not kwargs.get("is_causal", False) and ops.attention_supported(query, key, value)
    key_mask = mask_off = None
    ready = getattr(attention_mask, "_bf_key_mask", None) if attention_mask is not None else None
    if usable and ready is not None and ready.shape == (query.shape[0], query.shape[2]):
        # built once per forward by _padding_mask_interface: additive fp32 [B, T] + the device flag "hides nothing"
        key_mask, mask_off = ready, attention_mask._bf_mask_off
    elif usable and attention_mask is not None:
        m = attention_mask
        B, H, T, _ = query.shape
        # a padding mask: [B, 1, 1 or T (broadcast), T]; per-query structure is not handled here
        if m.dim() == 4 and m.shape[0] == B and m.shape[1] == 1 and m.shape[3] == T and (m.shape[2] == 1 or m.stride(2) == 0):
            row = m[:, 0, 0, :]
            if row.dtype == torch.bool:
                key_mask = torch.where(row, 0.0, float("-inf")).to(torch.float32)
            else:
                key_mask = row.to(torch.float32).contiguous()
        else:
            usable = False
    if not usable:
        if attention_mask is not None and attention_mask.dtype not in (torch.bool, query.dtype):
            attention_mask = attention_mask.to(query.dtype)  # the framework's kernels want bool or the query's dtype
        return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)
    scale = scaling if scaling is not None else query.shape[-1] ** -0.5
    drop = ops.Dropout(dropout, bfr.STATE.seed, bfr.dropout_call(), bfr.dropout_site(module), bfr.dropout_origin(),
                       bfr.dropout_counter()) if dropout > 0.0 else None
    if need_grad:  # training: the same kernel, with bf_attention_bwd behind it
        return ops.AttentionFn.apply(query, key, value, key_mask, mask_off, scale, drop), None
    return ops.attention_forward(query, key, value, key_mask, scale, mask_off, drop=drop), None


def _padding_mask_interface(batch_size, q_length=None, kv_length=None, q_offset=0, kv_offset=0, mask_function=None,
                            attention_mask=None, **kwargs):
    """Mask function in the HuggingFace `AttentionMaskInterface` convention for models routed through
    `_attention_interface`.  A plain bidirectional padding mask [B, T] becomes, ONCE per forward and without a host
    round trip, the additive fp32 key mask bf_attention_fwd reads plus a one-byte device flag "nothing is hidden" that
    lets the kernel skip the mask (the framework's own function answers that question with `mask.all()` on the host —
    a device synchronisation in every forward, and a different code path under HIP-graph capture).  The 4-D tensor
    returned ([B, 1, 1, T] additive) is what the framework's attention takes when the kernel does not apply.
    Anything else (4-D masks, extra mask functions, cached keys) goes to the framework's scaled-dot-product mask."""
    from transformers.masking_utils import bidirectional_mask_function, sdpa_mask

    plain = (attention_mask is not None and attention_mask.dim() == 2 and mask_function is bidirectional_mask_function
             and not kwargs.get("use_vmap", False) and q_offset == 0 and kv_offset == 0
             and attention_mask.shape == (batch_size, kv_length))
    if attention_mask is None and mask_function is bidirectional_mask_function:
        return None
    if not plain:
        return sdpa_mask(batch_size=batch_size, q_length=q_length, kv_length=kv_length, q_offset=q_offset,
                         kv_offset=kv_offset, mask_function=mask_function, attention_mask=attention_mask, **kwargs)
    visible = attention_mask if attention_mask.dtype == torch.bool else attention_mask != 0
    additive = torch.where(visible, 0.0, float("-inf")).to(torch.float32)  # one launch (fp32 already: .to is a no-op)
    out = additive[:, None, None, :]
    out._bf_key_mask = additive
    out._bf_mask_off = visible.all().reshape(1)  # stays on the device
    return out


def fuse_attention(model: torch.nn.Module) -> bool:
    """Route the wrapped HuggingFace model's attention through bf_attention_fwd: registers an attention function in
    transformers' AttentionInterface (mask format: the scaled-dot-product one) and selects it in the model's config.
    The function falls back to the framework's attention for anything it does not take.  Returns False (and changes
    nothing) when the model has no HuggingFace config or transformers lacks the interface."""
    try:
        from transformers import AttentionInterface
        from transformers.masking_utils import AttentionMaskInterface, sdpa_mask
    except ImportError:
        return False
    inner = model.model if isinstance(model, Model) and model.model is not None else model
    config = getattr(inner, "config", None)
    if config is None or not hasattr(config, "_attn_implementation"):
        return False
    AttentionInterface.register(_ATTENTION_NAME, _attention_interface)
    AttentionMaskInterface.register(_ATTENTION_NAME, _padding_mask_interface)
    for m in inner.modules():
        c = getattr(m, "config", None)
        if c is not None and hasattr(c, "_attn_implementation"):
            c._attn_implementation = _ATTENTION_NAME
    return True
