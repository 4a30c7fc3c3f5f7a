"""Host-side plumbing between torch tensors and the C-ABI (include/bayeformers_amd.h).

torch is used here for device memory, the current HIP stream and autograd bookkeeping only; every arithmetic
step of the Monte-Carlo forward path runs in the HIP kernels of csrc/.  There is no CPU fallback: tensors that
are not on a ROCm device raise.
"""
import ctypes
import os
from typing import Optional

import torch
from torch import Tensor

from . import _C
from . import random as bfr

_TORCH2BF = {torch.float32: _C.BF_DT_F32, torch.bfloat16: _C.BF_DT_BF16, torch.float16: _C.BF_DT_F16}

_WORKSPACES = {}


def _require_device(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _C.BayeFormersAMDError(
            f"{what} lives on '{t.device}': the bayeformers_amd forward path runs only on a ROCm device "
            "(HIP kernels, no CPU fallback) — move the module and its inputs to 'cuda'")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def workspace(device: torch.device, nbytes: int) -> Tensor:
    """Per-(device, stream) scratch buffer, grown on demand (stream-ordered reuse is safe on one stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream_ptr())
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def philox_normal_host(n: int, seed: int, sample: int, stream_id: int, offset: int = 0) -> Tensor:
    """Host twin of the device epsilon (bf_philox_normal_host): fp32 CPU tensor of n normals."""
    out = torch.empty(int(n), dtype=torch.float32)
    _C.check(_C.lib().bf_philox_normal_host(out.data_ptr(), int(n), int(seed), int(sample) & 0xFFFFFFFF,
                                            int(stream_id), int(offset)), "bf_philox_normal_host")
    return out


_FUSED_ROWS_SET = False


def fused_small_rows(N: int, K: int) -> int:
    """Rows per sample up to which bf_linear_fwd runs an N x K layer as ONE fused kernel (bf_fused_small_rows_for: the
    measured crossover; BF_FUSED_SMALL_MAX_ROWS caps it for developer A/B runs)."""
    global _FUSED_ROWS_SET
    lib = _C.lib()
    if not _FUSED_ROWS_SET:
        _FUSED_ROWS_SET = True
        v = os.environ.get("BF_FUSED_SMALL_MAX_ROWS")
        if v is not None:
            _C.check(lib.bf_set_fused_small_max_rows(int(v)), "bf_set_fused_small_max_rows")
    return lib.bf_fused_small_rows_for(int(N), int(K))


def philox_normal(n: int, S: int, seed: int, sample_base: int, stream_id: int, device="cuda") -> Tensor:
    """Device epsilon: [S, n] fp32 (test hook for the RNG contract)."""
    out = torch.empty((int(S), int(n)), dtype=torch.float32, device=device)
    _C.check(_C.lib().bf_philox_normal(out.data_ptr(), int(n), int(S), int(seed), int(sample_base) & 0xFFFFFFFF,
                                       int(stream_id), _stream_ptr()), "bf_philox_normal")
    return out


_NO_ALIAS = os.environ.get("BF_NO_PRIOR_ALIAS") is not None  # developer A/B: always read the prior's mu / rho


def prior_alias(gaussian, prior) -> Optional[float]:
    """sigma_p when `prior` is the MOPED prior of a FROZEN mean — Gaussian(mu = the posterior's mean, rho = one constant),
    /root/reference/bayeformers/nn/layers/linear.py:147-150 with freeze=True — else None.  The sampling kernel then reads
    8 instead of 16 bytes per scalar (bf_prior_t.pi == 1).  Checked on the tensors' CONTENTS (after a device move the two
    means are equal copies, no longer one storage), once per state: the verdict is cached with the tensors' addresses and
    version counters, so an in-place edit of either (an optimizer step on a trainable mean, load_state_dict) is seen and
    re-checked.  A trainable mean is never aliased: it leaves the prior's at its first update.
    An edit through `.data` (the reference's idiom, layers/linear.py:140-150) moves no version counter: the sampling kernels
    therefore spot-check the assertion on the device (one element per wave against prior.mu / prior.rho), poison the
    log-prior and bump the library's stale counter when it fails; `stale_priors_seen()` notices that at the next forward
    and `invalidate_caches(model)` drops every cached verdict."""
    mu, pmu, prho = gaussian.mu, prior.mu, prior.rho
    if _NO_ALIAS or mu.requires_grad or pmu.shape != mu.shape or prho.shape != mu.shape or pmu.dtype != torch.float32:
        return None
    state = (mu.data_ptr(), mu._version, pmu.data_ptr(), pmu._version, prho.data_ptr(), prho._version, bfr.STATE.stale_epoch)
    hit = getattr(prior, "_bf_alias", None)
    if hit is not None and hit[0] == state:
        return hit[1]
    with torch.no_grad():
        sigma_p, rho_p = None, 0.0
        lo, hi = torch.aminmax(prho)
        if bool(lo == hi) and (pmu.data_ptr() == mu.data_ptr() or torch.equal(pmu, mu)):
            sigma_p, rho_p = float(torch.nn.functional.softplus(lo.float())), float(lo)
            if not (sigma_p > 0.0 and sigma_p < float("inf")):
                sigma_p = None
    prior._bf_alias = (state, sigma_p, rho_p)
    return sigma_p


_WARNED_EPOCH = [0]


def stale_epoch() -> int:
    """The library's stale-prior counter (bf_stale_counter) as of the running forward: some kernel bumps it when it finds a
    prior's baked constants — an asserted MOPED alias, a mixture's pi / sigma1 / sigma2 — different from the tensors they
    were read from (an in-place edit through `.data`).  Every cached copy of prior state (`prior_alias`,
    `ScaledGaussianMixture.constants`, the sampling plan) remembers the epoch it was made in and is void in a later one —
    whichever module, model or bare layer it belongs to.  `refresh_stale_epoch()` reads the counter (a word of pinned host
    memory: no synchronisation) once per forward."""
    return bfr.STATE.stale_epoch


def refresh_stale_epoch() -> None:
    now = _C.stale_counter()
    if now != bfr.STATE.stale_epoch:
        bfr.STATE.stale_epoch = now
        if now != _WARNED_EPOCH[0]:
            import warnings

            _WARNED_EPOCH[0] = now
            warnings.warn("bayeformers_amd: a prior's tensors were edited in place through `.data` after a forward had "
                          "cached their state; the log_prior of the forward(s) since the edit is NaN — every cached copy "
                          "is void from now on (call bayeformers_amd.invalidate_caches(model) right after such an edit)")


def invalidate_caches(module) -> None:
    """Forget every host-side copy of `module`'s prior state: the MOPED-alias verdicts (`prior_alias`), the mixture priors'
    constants (`ScaledGaussianMixture.constants`) and the sampling plans built on them.  Call it after editing a prior's or
    a frozen mean's tensors in place through `.data` — every other edit (optimizer steps, load_state_dict, `.to()`,
    in-place ops on the parameter itself) is seen through the version counters and addresses."""
    from .nn.model import Model
    from .nn.parameters.gaussian import Gaussian, ScaledGaussianMixture

    for m in module.modules():
        if isinstance(m, Gaussian):
            m.__dict__.pop("_bf_alias", None)
        elif isinstance(m, ScaledGaussianMixture):
            m._consts = None
        elif isinstance(m, Model):
            m._plan = None


def fill_prior(dst: "_C.bf_prior_t", prior, gaussian=None) -> bool:
    """Describe a prior module to the kernels.  Returns False for a user-defined Parameter (generic path)."""
    from .nn.parameters.base import NoneParameter
    from .nn.parameters.gaussian import Gaussian, ScaledGaussianMixture

    if isinstance(prior, ScaledGaussianMixture):
        pi, s1, s2 = prior.constants()
        dst.kind, dst.pi, dst.sigma1, dst.sigma2 = _C.BF_PRIOR_MIXTURE, pi, s1, s2
        dst.d_mu = dst.d_rho = None
        # the kernels compare the three values with the device scalars they are a copy of (an edit through .data is
        # invisible to the host-side cache; host-resident scalars are re-read by constants() on every call instead)
        on_dev = prior.pi.is_cuda and prior.sigma1.is_cuda and prior.sigma2.is_cuda
        dst.d_pi, dst.d_sigma1, dst.d_sigma2 = ((prior.pi.data_ptr(), prior.sigma1.data_ptr(), prior.sigma2.data_ptr())
                                                if on_dev else (None, None, None))
        return True
    if isinstance(prior, Gaussian):
        _require_device(prior.mu, "prior.mu")
        dst.kind = _C.BF_PRIOR_GAUSSIAN
        dst.d_mu, dst.d_rho = prior.mu.data_ptr(), prior.rho.data_ptr()
        sigma_p = prior_alias(gaussian, prior) if gaussian is not None else None
        # (asserted alias: sigma2 carries the constant rho the kernels spot-check prior.rho against)
        dst.pi, dst.sigma1, dst.sigma2 = (1.0, sigma_p, prior._bf_alias[2]) if sigma_p is not None else (0.0, 0.0, 0.0)
        dst.d_pi = dst.d_sigma1 = dst.d_sigma2 = None
        return True
    if prior is None or isinstance(prior, NoneParameter):
        dst.kind = _C.BF_PRIOR_NONE
        dst.d_mu = dst.d_rho = dst.d_pi = dst.d_sigma1 = dst.d_sigma2 = None
        return True
    return False


def fill_tensor(dst: "_C.bf_tensor_t", gaussian, prior, stream_id: int) -> bool:
    mu, rho = gaussian.mu, gaussian.rho
    _require_device(mu, "mu")
    if not (mu.is_contiguous() and rho.is_contiguous() and mu.dtype == torch.float32 and rho.dtype == torch.float32):
        raise _C.BayeFormersAMDError("mu/rho must be contiguous fp32 tensors")
    dst.d_mu, dst.d_rho, dst.n = mu.data_ptr(), rho.data_ptr(), mu.numel()
    dst.stream_id = stream_id
    dst.d_sample_out = None
    dst.out_dtype = _C.BF_DT_F32
    if isinstance(prior, type(gaussian)) and prior.mu.numel() != mu.numel():
        raise _C.BayeFormersAMDError("Gaussian prior must have the shape of the parameter it is a prior of")
    return fill_prior(dst.prior, prior, gaussian)


def sample_logprob(gaussians, priors, stream_ids, S: int, seed: int, sample_base: int, out_dtype=None):
    """Fused sampling + log-probs of 1 or 2 Gaussian parameters (bf_sample_logprob).

    Returns (samples, logprob) where samples is a list of [S, *shape] tensors (or None when out_dtype is None)
    and logprob is a [S, 2] float64 tensor {log_prior, log_variational_posterior} summed over the parameters."""
    n = len(gaussians)
    arr = (_C.bf_tensor_t * n)()
    dev = gaussians[0].mu.device
    outs = []
    for i, (g, pr, sid) in enumerate(zip(gaussians, priors, stream_ids)):
        if not fill_tensor(arr[i], g, pr, sid):
            raise _C.BayeFormersAMDError("sample_logprob: user-defined priors go through Linear's generic path")
        if out_dtype is not None:
            o = torch.empty((S,) + tuple(g.mu.shape), dtype=out_dtype, device=dev)
            arr[i].d_sample_out, arr[i].out_dtype = o.data_ptr(), _TORCH2BF[out_dtype]
            outs.append(o)
    lib = _C.lib()
    need = lib.bf_sample_logprob_workspace_bytes(arr, n, S)
    ws = workspace(dev, need)
    lp = torch.empty((S, 2), dtype=torch.float64, device=dev)
    _C.check(lib.bf_sample_logprob(arr, n, S, seed, sample_base & 0xFFFFFFFF, lp.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _stream_ptr()), "bf_sample_logprob")
    return (outs if out_dtype is not None else None), lp


def gemm_nt(x: Tensor, w: Tensor, bias: Optional[Tensor], S: int, M: int, N: int, K: int, x_sample_stride: int,
            y_dtype: torch.dtype, act: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """y[s] = act(x[s] w[s]^T + bias[s]) on the matrix cores (bf_gemm_nt_act).  w: [S,N,K]; returns [S,M,N] (`out`: a
    contiguous [S,M,N] tensor of y_dtype to write into — a window of a larger sample-major buffer)."""
    y = out if out is not None else torch.empty((S, M, N), dtype=y_dtype, device=x.device)
    _C.check(_C.lib().bf_gemm_nt_act(x.data_ptr(), _TORCH2BF[x.dtype], x_sample_stride, w.data_ptr(),
                                     _TORCH2BF[w.dtype], bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                     _TORCH2BF[y_dtype], S, M, N, K, act, _stream_ptr()), "bf_gemm_nt_act")
    return y


def gemm_nt_act_pre(x: Tensor, w: Tensor, bias: Optional[Tensor], S: int, M: int, N: int, K: int, x_sample_stride: int,
                    y_dtype: torch.dtype, act: int):
    """(act(y), y) with y[s] = x[s] w[s]^T + bias[s] (bf_gemm_nt_act_pre): the forward of a training step keeps the
    pre-activation for the backward of the fused activation.  Both [S,M,N]."""
    y = torch.empty((S, M, N), dtype=y_dtype, device=x.device)
    pre = torch.empty((S, M, N), dtype=y_dtype, device=x.device)
    _C.check(_C.lib().bf_gemm_nt_act_pre(x.data_ptr(), _TORCH2BF[x.dtype], x_sample_stride, w.data_ptr(),
                                         _TORCH2BF[w.dtype], bias.data_ptr() if bias is not None else None,
                                         y.data_ptr(), pre.data_ptr(), _TORCH2BF[y_dtype], S, M, N, K, act,
                                         _stream_ptr()), "bf_gemm_nt_act_pre")
    return y, pre


def gemm_nt_layers(x: Tensor, w: Tensor, bias: Optional[Tensor], L: int, S: int, M: int, N: int, K: int,
                   x_sample_stride: int, y_dtype: torch.dtype, act: int = 0) -> Tensor:
    """y[l][s] = act(x[s] w[l][s]^T + bias[l][s]) for L layers sharing x, one launch (bf_gemm_nt_layers).
    w: [L,S,N,K]; bias: [L,S,N] fp32 or None; returns [L,S,M,N]."""
    y = torch.empty((L, S, M, N), dtype=y_dtype, device=x.device)
    _C.check(_C.lib().bf_gemm_nt_layers(x.data_ptr(), _TORCH2BF[x.dtype], x_sample_stride, w.data_ptr(),
                                        _TORCH2BF[w.dtype], bias.data_ptr() if bias is not None else None,
                                        y.data_ptr(), _TORCH2BF[y_dtype], L, S, M, N, K, act, _stream_ptr()),
             "bf_gemm_nt_layers")
    return y


def gemm_tn(a: Tensor, b: Tensor) -> Tensor:
    """out[i] = a[i]^T b[i] in fp32 (bf_gemm_tn): the weight-gradient GEMM dW = dy^T x.  a: [batch, Mc, N],
    b: [batch, Mc, K], 16-bit; returns [batch, N, K] float32."""
    _require_device(a, "a")
    batch, Mc, N = a.shape
    K = b.shape[2]
    if b.shape[:2] != a.shape[:2] or a.dtype != b.dtype or not (a.is_contiguous() and b.is_contiguous()):
        raise _C.BayeFormersAMDError("gemm_tn: a [batch,Mc,N] and b [batch,Mc,K] must be contiguous, same dtype")
    out = torch.empty((batch, N, K), dtype=torch.float32, device=a.device)
    _C.check(_C.lib().bf_gemm_tn(a.data_ptr(), b.data_ptr(), out.data_ptr(), _TORCH2BF[a.dtype], batch, Mc, N, K,
                                 _stream_ptr()), "bf_gemm_tn")
    return out


def gemm_nn(x: Tensor, w: Tensor) -> Tensor:
    """y[s] = x[s] w[s] (bf_gemm_nn): the input-gradient GEMM dx = dy W_s.  x: [S, M, N], w: [S, N, K], 16-bit;
    returns [S, M, K] of the same dtype."""
    _require_device(x, "x")
    S, M, N = x.shape
    K = w.shape[2]
    if w.shape[:2] != (S, N) or x.dtype != w.dtype or not (x.is_contiguous() and w.is_contiguous()):
        raise _C.BayeFormersAMDError("gemm_nn: x [S,M,N] and w [S,N,K] must be contiguous, same dtype")
    y = torch.empty((S, M, K), dtype=x.dtype, device=x.device)
    _C.check(_C.lib().bf_gemm_nn(x.data_ptr(), w.data_ptr(), y.data_ptr(), _TORCH2BF[x.dtype], S, M, N, K,
                                 _stream_ptr()), "bf_gemm_nn")
    return y


def gemm_nn_actgrad_supported(x: Tensor, w: Tensor, pre: Tensor) -> bool:
    S, M, N = x.shape
    K = w.shape[2]
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and pre.dtype == x.dtype
            and x.is_contiguous() and w.is_contiguous() and pre.is_contiguous() and pre.numel() == S * M * K):
        return False
    return bool(_C.lib().bf_gemm_nn_actgrad_supported(x.data_ptr(), w.data_ptr(), x.data_ptr(), pre.data_ptr(), _TORCH2BF[x.dtype],
                                                      S, M, N, K))


def gemm_nn_actgrad(x: Tensor, w: Tensor, pre: Tensor, act: int = 1) -> Tensor:
    """y[s] = (x[s] w[s]) o act'(pre[s]) (bf_gemm_nn_actgrad): the input-gradient GEMM of a layer whose input was act(pre), with
    the activation's derivative in its epilogue.  x: [S, M, N], w: [S, N, K], pre: [S, M, K] (or [S*M, K]); returns [S, M, K]."""
    S, M, N = x.shape
    K = w.shape[2]
    y = torch.empty((S, M, K), dtype=x.dtype, device=x.device)
    _C.check(_C.lib().bf_gemm_nn_actgrad(x.data_ptr(), w.data_ptr(), y.data_ptr(), pre.data_ptr(), _TORCH2BF[x.dtype], S, M, N, K,
                                         int(act), _stream_ptr()), "bf_gemm_nn_actgrad")
    return y


def gemm_nn_layers(x: Tensor, w: Tensor) -> Tensor:
    """y[s] = sum_l x[l][s] w[l][s] (bf_gemm_nn_layers): the one input gradient of L layers that read the same
    activations.  x: [L, S, M, N], w: [L, S, N, K], 16-bit; returns [S, M, K]."""
    _require_device(x, "x")
    L, S, M, N = x.shape
    K = w.shape[3]
    if w.shape[:3] != (L, S, N) or x.dtype != w.dtype or not (x.is_contiguous() and w.is_contiguous()):
        raise _C.BayeFormersAMDError("gemm_nn_layers: x [L,S,M,N] and w [L,S,N,K] must be contiguous, same dtype")
    y = torch.empty((S, M, K), dtype=x.dtype, device=x.device)
    _C.check(_C.lib().bf_gemm_nn_layers(x.data_ptr(), w.data_ptr(), y.data_ptr(), _TORCH2BF[x.dtype], L, S, M, N, K,
                                        _stream_ptr()), "bf_gemm_nn_layers")
    return y


class LinearPlan:
    """Cached ctypes descriptors of one bnn.Linear (pointers are refreshed per call; structs are reused).  A cache only:
    copying or pickling the layer starts it afresh (ctypes structs holding pointers can be neither)."""

    def __deepcopy__(self, memo):
        return LinearPlan()

    def __reduce__(self):
        return (LinearPlan, ())

    def __init__(self):
        self.w = _C.bf_tensor_t()
        self.b = _C.bf_tensor_t()


def linear_forward(layer, x: Tensor, S: int, seed: int, sample_base: int, lp_out: Tensor) -> Tensor:
    """Linear.forward for S samples: y[s] = x[s] W_s^T + b_s and lp_out[s] = {log_prior, log_q}.

    x: [S*M, K] (sample-major) or [M, K] with S == 1.  lp_out: [S, 2] float64 on x's device."""
    from .nn.parameters.base import NoneParameter

    _require_device(x, "input")
    K, N = layer.in_features, layer.out_features
    if x.dtype not in _TORCH2BF:
        raise _C.BayeFormersAMDError(f"unsupported input dtype {x.dtype}")
    if not x.is_contiguous():
        x = x.contiguous()
    rows = x.numel() // K
    if rows % S:
        raise _C.BayeFormersAMDError(f"input rows ({rows}) are not a multiple of the sample count S={S}")
    M = rows // S
    cdt = layer.compute_dtype or bfr.get_compute_dtype()
    if x.dtype != torch.float32 and x.dtype != cdt:
        raise _C.BayeFormersAMDError(f"input dtype {x.dtype} does not match compute dtype {cdt} (fp32 inputs always do)")
    if cdt == torch.float32 and x.dtype != torch.float32:
        raise _C.BayeFormersAMDError("compute dtype fp32 needs fp32 inputs")
    has_bias = not isinstance(layer.bias, NoneParameter)
    plan = layer._plan
    known = fill_tensor(plan.w, layer.weight, layer.weight_prior, 2 * layer.layer_id)
    if has_bias:
        known = fill_tensor(plan.b, layer.bias, layer.bias_prior, 2 * layer.layer_id + 1) and known
    if not known:
        return _linear_forward_generic(layer, x, S, M, N, K, seed, sample_base, lp_out, cdt, has_bias)
    lib = _C.lib()
    y = torch.empty((S * M, N), dtype=x.dtype, device=x.device)
    need = lib.bf_linear_fwd_workspace_bytes(S, M, N, K, int(has_bias), _TORCH2BF[cdt], _TORCH2BF[x.dtype])
    ws = workspace(x.device, need)
    _C.check(lib.bf_linear_fwd(x.data_ptr(), _TORCH2BF[x.dtype], M * K, ctypes.byref(plan.w),
                               ctypes.byref(plan.b) if has_bias else None, y.data_ptr(), _TORCH2BF[x.dtype],
                               _TORCH2BF[cdt], S, M, N, K, seed, sample_base & 0xFFFFFFFF, lp_out.data_ptr(),
                               ws.data_ptr(), ws.numel(), _stream_ptr()), "bf_linear_fwd")
    return y


def linear_forward_ws(layer, x: Tensor, S: int, seed: int, sample_base: int, lp_out: Tensor, row_shares: int = 0) -> Tensor:
    """Linear.forward for S samples in ONE launch, weight-stationary (bf_linear_fwd_ws): the measured alternative to
    sampling launch + GEMM for large M (LABBOOK.md 4.3).  Same arguments and results as linear_forward; 16-bit x only."""
    from .nn.parameters.base import NoneParameter

    if not hasattr(_C.lib(), "bf_linear_fwd_ws"):
        raise _C.BayeFormersAMDError("bf_linear_fwd_ws is a developer-build entry point: python -m bayeformers_amd.build --dev, "
                                     "then BF_LIB_PATH=bayeformers_amd/lib/libbayeformers_amd_dev.so")
    _require_device(x, "input")
    K, N = layer.in_features, layer.out_features
    x = x if x.is_contiguous() else x.contiguous()
    M = x.numel() // K // S
    cdt = layer.compute_dtype or bfr.get_compute_dtype()
    has_bias = not isinstance(layer.bias, NoneParameter)
    w, b = _C.bf_tensor_t(), _C.bf_tensor_t()
    ok = fill_tensor(w, layer.weight, layer.weight_prior, 2 * layer.layer_id)
    if has_bias:
        ok = fill_tensor(b, layer.bias, layer.bias_prior, 2 * layer.layer_id + 1) and ok
    if not ok:
        raise _C.BayeFormersAMDError("linear_forward_ws: user-defined priors are not supported")
    lib = _C.lib()
    y = torch.empty((S * M, N), dtype=x.dtype, device=x.device)
    ws = workspace(x.device, lib.bf_linear_fwd_ws_workspace_bytes(S, N))
    _C.check(lib.bf_linear_fwd_ws(x.data_ptr(), _TORCH2BF[x.dtype], M * K, ctypes.byref(w),
                                  ctypes.byref(b) if has_bias else None, y.data_ptr(), _TORCH2BF[x.dtype],
                                  _TORCH2BF[cdt], S, M, N, K, seed, sample_base & 0xFFFFFFFF, int(row_shares),
                                  lp_out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()), "bf_linear_fwd_ws")
    return y


def _linear_forward_generic(layer, x, S, M, N, K, seed, sample_base, lp_out, cdt, has_bias):
    """User-defined prior (any Parameter with log_prob): the kernels sample W_s/b_s in fp32 and produce log q;
    the prior's own log_prob is then called on each sample, as the reference does (layers/linear.py:99-100)."""
    from .nn.parameters.base import NoneParameter

    gs = [layer.weight] + ([layer.bias] if has_bias else [])
    sids = [2 * layer.layer_id] + ([2 * layer.layer_id + 1] if has_bias else [])
    outs, lp = sample_logprob(gs, [NoneParameter()] * len(gs), sids, S, seed, sample_base, out_dtype=torch.float32)
    for s in range(S):
        v = layer.weight_prior.log_prob(outs[0][s])
        if has_bias:
            v = v + layer.bias_prior.log_prob(outs[1][s])
        lp[s, 0] = v
    lp_out.copy_(lp)
    w = outs[0] if cdt == torch.float32 else outs[0].to(cdt)
    y = gemm_nt(x, w, outs[1] if has_bias else None, S, M, N, K, M * K, x.dtype)
    return y.view(S * M, N)


def planned_linear_forward(x: Tensor, w_s: Tensor, b_s: Optional[Tensor], S: int, N: int, K: int, act: int = 0,
                           want_pre: bool = False):
    """y[s] = act(x[s] W_s^T + b_s) with W_s/b_s already sampled by the model's cross-layer plan (plan.SamplePlan).
    want_pre: return (y, pre-activation) — what the backward of a fused activation needs."""
    _require_device(x, "input")
    if x.dtype not in _TORCH2BF:
        raise _C.BayeFormersAMDError(f"unsupported input dtype {x.dtype}")
    if x.dtype != torch.float32 and x.dtype != w_s.dtype:
        raise _C.BayeFormersAMDError(
            f"input dtype {x.dtype} does not match compute dtype {w_s.dtype} (fp32 inputs always do)")
    if not x.is_contiguous():
        x = x.contiguous()
    rows = x.numel() // K
    if rows % S:
        raise _C.BayeFormersAMDError(f"input rows ({rows}) are not a multiple of the sample count S={S}")
    M = rows // S
    if want_pre:
        y, pre = gemm_nt_act_pre(x, w_s, b_s, S, M, N, K, M * K, x.dtype, act)
        return y.view(S * M, N), pre.view(S * M, N)
    return gemm_nt(x, w_s, b_s, S, M, N, K, M * K, x.dtype, act).view(S * M, N)


COLSUMS_FOLDED = [0]  # bias gradients whose column sums came with the output gradient (tests, diagnostics)

# Column sums a gradient's PRODUCER left for its consumer (attention_backward -> linear_backward of query / key / value).
# Between the two the gradient passes through autograd — as the same tensor object or as views of it (HF's head split /
# merge) — so the hand-over is keyed by where the gradient lives: (storage address, byte offset, elements), valid while that
# very storage is alive AND unmodified: autograd's engine may add a second gradient into the producer's tensor IN PLACE
# (an output with two consumers); that moves the version counter every view of the storage shares, and the offer is void.
# An offer is taken once; a new bnn.Model forward drops what nobody took.
_COLSUM_OFFERS = {}


def _offer_key(t: Tensor):
    return (t.untyped_storage().data_ptr(), t.storage_offset() * t.element_size(), t.numel(), t.dtype)


def offer_colsum(grad: Tensor, colsum: Tensor) -> None:
    """`colsum` [S, N] fp32 = per-sample column sums of the contiguous gradient `grad` ([S*M, N] rows), as stored."""
    from torch.multiprocessing.reductions import StorageWeakRef

    if len(_COLSUM_OFFERS) > 64:
        _COLSUM_OFFERS.clear()
    _COLSUM_OFFERS[_offer_key(grad)] = (StorageWeakRef(grad.untyped_storage()), colsum, grad._version)


def take_colsum(grad: Tensor, S: int, N: int) -> Optional[Tensor]:
    if not _COLSUM_OFFERS or not grad.is_contiguous():
        return None
    hit = _COLSUM_OFFERS.pop(_offer_key(grad), None)
    if hit is None:
        return None
    ref, colsum, version = hit
    if ref.expired() or grad._version != version or tuple(colsum.shape) != (S, N) or colsum.device != grad.device:
        return None  # the storage the offer described is gone (its address may have been reused) or shapes do not match
    return colsum


def linear_backward(layer, x: Tensor, grad_y: Tensor, S: int, seed: int, sample_base: int, cdt: torch.dtype,
                    need_x: bool, need_mu_w: bool, need_mu_b: bool, w_samples: Optional[Tensor] = None,
                    act: int = 0, act_pre: Optional[Tensor] = None, dy_colsum: Optional[Tensor] = None):
    """Gradients of the sampled-weight linear layer (bf_linear_bwd).  Returns (dx, dmu_w, drho_w, dmu_b, drho_b);
    entries that are not needed are None.  x: [S*M, K] as saved by the forward; grad_y: [S*M, N].
    act / act_pre: the forward fused act() into its GEMM and kept the pre-activation [S*M, N]: grad_y is the gradient
    of act(y)."""
    from .nn.parameters.base import NoneParameter

    K, N = layer.in_features, layer.out_features
    M = x.shape[0] // S
    has_bias = not isinstance(layer.bias, NoneParameter)
    if dy_colsum is None and has_bias and not act:
        # the kernel that produced grad_y may have left its per-sample column sums with it (attention_backward): valid for
        # exactly this tensor, as it is (same dtype: no rounding in between)
        cs = take_colsum(grad_y, S, N) if (grad_y.dtype == cdt and grad_y.numel() == S * M * N) else None
        if cs is not None:
            dy_colsum = cs
            COLSUMS_FOLDED[0] += 1
    xg = (x if x.dtype == cdt else x.to(cdt)).contiguous()
    dy = grad_y.reshape(S * M, N)
    dy = (dy if dy.dtype == cdt else dy.to(cdt)).contiguous()
    dev = x.device
    w, b = _C.bf_tensor_t(), _C.bf_tensor_t()
    fill_tensor(w, layer.weight, NoneParameter(), 2 * layer.layer_id)
    if w_samples is not None and w_samples.dtype == cdt:
        # the forward's W_s are still resident (sampling plan): read instead of regenerated
        w.d_sample_out, w.out_dtype = w_samples.data_ptr(), _TORCH2BF[cdt]
    if has_bias:
        fill_tensor(b, layer.bias, NoneParameter(), 2 * layer.layer_id + 1)
    dx = torch.empty((S * M, K), dtype=cdt, device=dev) if need_x else None
    # A parameter whose gradient has a standing destination — its slot in a flat all-reduce bucket
    # (training.GradientBuckets, between zero() and finish()) — gets the kernel's result written there: no copy into the
    # bucket afterwards, and autograd is handed None for it (the slot IS the accumulated gradient).
    sunk = []

    def dest(param, shape, needed=True):
        if not needed:
            return None
        sink = getattr(param, "_bf_grad_sink", None)
        slot = sink.slot(param) if sink is not None else None
        if slot is not None and slot.dtype == torch.float32 and slot.shape == shape and slot.device == dev:
            sunk.append((sink, param))
            return slot
        return torch.empty(shape, dtype=torch.float32, device=dev)

    # A layer whose weight reduction is DEFERRED (training.DeferredParamGrads, armed for this step): bf_linear_bwd leaves the
    # per-sample gradients dW_s in the manager's buffer and skips the reduction; ONE bf_param_grad_table launch after the
    # backward pass reduces every such layer into the manager's gradient buffers.  Autograd is handed None for mu / rho.
    defer = getattr(layer, "_bf_pg_defer", None)
    dw_keep = defer.keep_buffer(layer, S, M, cdt, seed, sample_base) if defer is not None else None
    db_keep = None
    if dw_keep is not None:
        dmu_w = drho_w = None
        db_keep = defer.keep_bias_buffer(layer, dy_colsum if not act else None) if has_bias else None
    else:
        dmu_w = dest(layer.weight.mu, (N, K), need_mu_w)
        drho_w = dest(layer.weight.rho, (N, K))
    if db_keep is not None:
        dmu_b = drho_b = None
    else:
        dmu_b = dest(layer.bias.mu, (N,), has_bias and need_mu_b) if has_bias else None
        drho_b = dest(layer.bias.rho, (N,)) if has_bias else None
    lib = _C.lib()
    if act:
        if act_pre is None or act_pre.dtype != cdt or act_pre.numel() != S * M * N:
            raise _C.BayeFormersAMDError("linear_backward: a fused activation needs the forward's pre-activation "
                                         f"as a [{S * M}, {N}] {cdt} tensor")
        act_pre = act_pre.contiguous()
    need = lib.bf_linear_bwd_workspace_bytes(S, M, N, K, int(has_bias), _TORCH2BF[cdt], act)
    ws = workspace(dev, need)
    ptr = lambda t: t.data_ptr() if t is not None else None
    if dw_keep is not None:
        drho_keep = defer.grad_view(layer.weight.rho)  # (where the table launch will write; nothing is written there now)
    _C.check(lib.bf_linear_bwd(xg.data_ptr(), M * K, dy.data_ptr(), _TORCH2BF[cdt], ctypes.byref(w),
                               ctypes.byref(b) if has_bias else None, ptr(dx), ptr(dmu_w),
                               ptr(drho_w) if dw_keep is None else drho_keep.data_ptr(), ptr(dmu_b),
                               ptr(drho_b) if db_keep is None else defer.grad_view(layer.bias.rho).data_ptr(), S, M, N, K, seed,
                               sample_base & 0xFFFFFFFF, act, ptr(act_pre) if act else None,
                               ptr(dy_colsum), ptr(dw_keep), ptr(db_keep), ws.data_ptr(), ws.numel(), _stream_ptr()), "bf_linear_bwd")
    if dx is not None and dx.dtype != x.dtype:
        dx = dx.to(x.dtype)
    if sunk:
        gone = {id(p) for _, p in sunk}
        for sink, param in sunk:
            sink.arrived(param)
        if has_bias:
            dmu_b = None if (dmu_b is not None and id(layer.bias.mu) in gone) else dmu_b
            drho_b = None if id(layer.bias.rho) in gone else drho_b
        dmu_w = None if (dmu_w is not None and id(layer.weight.mu) in gone) else dmu_w
        drho_w = None if id(layer.weight.rho) in gone else drho_w
    return dx, dmu_w, drho_w, dmu_b, drho_b


def kl_grad(gaussian, prior, stream_id: int, S: int, seed: int, sample_base: int, g: Tensor, need_mu: bool):
    """Gradient of sum_s g[s,0]*log_prior_s + g[s,1]*log_q_s w.r.t. one Gaussian's (mu, rho) (bf_kl_grad)."""
    t = _C.bf_tensor_t()
    if not fill_tensor(t, gaussian, prior, stream_id):
        raise _C.BayeFormersAMDError("kl_grad: user-defined priors have no kernel gradient")
    dev = gaussian.mu.device
    dmu = torch.empty_like(gaussian.mu, dtype=torch.float32) if need_mu else None
    drho = torch.empty_like(gaussian.rho, dtype=torch.float32)
    _C.check(_C.lib().bf_kl_grad(ctypes.byref(t), S, seed, sample_base & 0xFFFFFFFF, g.data_ptr(),
                                 dmu.data_ptr() if dmu is not None else None, drho.data_ptr(), _stream_ptr()),
             "bf_kl_grad")
    return dmu, drho


def embedding_forward(ids: Tensor, mu: Tensor, rho: Tensor, out_dtype: torch.dtype, S: int, seed: int,
                      sample_base: int, stream_id: int) -> Tensor:
    """Rows ids of S table draws W_s = mu + softplus(rho)*eps_s (bf_embedding_fwd); ids is [S*T] sample-major."""
    _require_device(mu, "Embedding.weight.mu")
    _require_device(ids, "Embedding input")
    V, D = mu.shape
    n = ids.numel()
    if n % S:
        raise _C.BayeFormersAMDError(f"embedding_forward: {n} tokens are not a multiple of S={S}")
    out = torch.empty((n, D), dtype=out_dtype, device=mu.device)
    if n:
        _C.check(_C.lib().bf_embedding_fwd(ids.data_ptr(), mu.data_ptr(), rho.data_ptr(), out.data_ptr(),
                                           _TORCH2BF[out_dtype], n, n // S, V, D, seed, sample_base & 0xFFFFFFFF,
                                           stream_id, _stream_ptr()), "bf_embedding_fwd")
    return out


def embedding_backward(ids: Tensor, grad: Tensor, mu: Tensor, rho: Tensor, S: int, seed: int, sample_base: int,
                       stream_id: int, need_mu: bool, need_rho: bool):
    """Scatter-add of the output gradient into (dmu, drho) of the table (bf_embedding_bwd)."""
    V, D = mu.shape
    n = ids.numel()
    grad = grad.contiguous()
    dmu = torch.zeros_like(mu, dtype=torch.float32) if need_mu else None
    drho = torch.zeros_like(rho, dtype=torch.float32) if need_rho else None
    if n and (need_mu or need_rho):
        _C.check(_C.lib().bf_embedding_bwd(ids.data_ptr(), grad.data_ptr(), _TORCH2BF[grad.dtype], rho.data_ptr(),
                                           dmu.data_ptr() if need_mu else None, drho.data_ptr() if need_rho else None,
                                           n, n // S, V, D, seed, sample_base & 0xFFFFFFFF, stream_id, _stream_ptr()),
                 "bf_embedding_bwd")
    return dmu, drho


class Dropout:
    """One dropout of the training-mode forward, as the kernels take it (the dropout contract of csrc/bf_philox.h):
    rate p, the Philox seed, `call` = the number of the forward it belongs to (reserved with the forward's sample indices, so
    a backward pass and the recomputation of a checkpointed block find the same mask) and `site` = the module.
    `origin` = (first global sample of this process's shard, samples in the shard): the kernels number their groups from
    first_sample x (groups per sample), so a sample's masks do not depend on which rank runs it (random.dropout_origin).
    `counter`: device-counter mode (random.use_device_counter) — a 1-element int32 device tensor, the forward's copy of the
    device-resident call counter, which the kernels add to `call` (then 0): forward, backward and a recomputed block read
    the same copy, and a step replayed from a HIP graph draws fresh masks every replay (random.dropout_counter)."""
    __slots__ = ("p", "seed", "call", "site", "origin", "counter")

    def __init__(self, p: float, seed: int, call: int, site: int, origin=(0, 1), counter=None):
        self.p, self.seed, self.call, self.site = float(p), int(seed), int(call) & 0xFFFFFFFF, int(site) & 0x7FFFFFFF
        self.origin = (int(origin[0]), max(1, int(origin[1])))
        self.counter = counter

    @property
    def d_call(self):
        return self.counter.data_ptr() if self.counter is not None else None

    def first_group(self, units: int, groups_per_unit: int) -> int:
        """Global index of the first group of a tensor made of `units` rows (or sequences) of `groups_per_unit` groups each,
        the units being this shard's samples in equal slabs."""
        start, s_local = self.origin
        if start == 0:
            return 0
        if units % s_local:
            # a shard that does not start at sample 0 must know how many units one sample has: numbering its groups from 0
            # instead would draw the masks of global samples 0.. — correlated across ranks, and silently
            raise _C.BayeFormersAMDError(
                f"dropout: a tensor of {units} rows / sequences cannot be split into this shard's {s_local} Monte-Carlo samples "
                f"(first global sample {start}); the dropped tensor must hold the samples in equal, sample-major slabs")
        return start * (units // s_local) * int(groups_per_unit)

    @property
    def keep_scale(self) -> float:
        thresh = min(65535, int(self.p * 65536.0 + 0.5))
        return 1.0 / (1.0 - thresh / 65536.0)


def dropout_keep_host(first_group: int, n_groups: int, d: "Dropout") -> Tensor:
    """Host twin of the kernels' keep decisions (bf_dropout_keep_host): uint8 [n_groups, 8], 1 = kept."""
    out = torch.empty((int(n_groups), 8), dtype=torch.uint8)
    _C.check(_C.lib().bf_dropout_keep_host(out.data_ptr(), int(first_group), int(n_groups), d.p, d.seed, d.call, d.site),
             "bf_dropout_keep_host")
    return out


def add_layernorm(x: Tensor, residual: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float,
                  drop: Optional[Dropout] = None) -> Tensor:
    """LayerNorm(x + residual) over the last axis in one pass (bf_add_layernorm); residual may be None.
    drop: LayerNorm(dropout(x) + residual) (bf_add_layernorm_dropout)."""
    _require_device(x, "add_layernorm input")
    N = x.shape[-1]
    x2 = x.reshape(-1, N)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    r2 = None
    if residual is not None:
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise _C.BayeFormersAMDError("add_layernorm: residual must match the input's shape and dtype")
        r2 = residual.reshape(-1, N)
        r2 = r2 if r2.is_contiguous() else r2.contiguous()
    if gamma.dtype != beta.dtype or gamma.dtype not in (torch.float32, x.dtype):
        raise _C.BayeFormersAMDError("add_layernorm: gamma/beta must be float32 or have the input's dtype")
    out = torch.empty_like(x2)
    if drop is not None and drop.p > 0.0:
        _C.check(_C.lib().bf_add_layernorm_dropout(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, gamma.data_ptr(),
                                                   beta.data_ptr(), _TORCH2BF[gamma.dtype], out.data_ptr(),
                                                   _TORCH2BF[x.dtype], x2.shape[0], N, float(eps), drop.p, drop.seed,
                                                   drop.call, drop.site, drop.first_group(x2.shape[0], N // 8), drop.d_call,
                                                   _stream_ptr()),
                 "bf_add_layernorm_dropout")
        return out.view(x.shape)
    _C.check(_C.lib().bf_add_layernorm(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, gamma.data_ptr(),
                                       beta.data_ptr(), _TORCH2BF[gamma.dtype], out.data_ptr(), _TORCH2BF[x.dtype],
                                       x2.shape[0], N, float(eps), _stream_ptr()), "bf_add_layernorm")
    return out.view(x.shape)


def embed_layernorm(ids: Tensor, type_ids: Optional[Tensor], pos_ids: Optional[Tensor], word: Tensor, type_table: Tensor,
                    pos_table: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    """LayerNorm(word[ids] + type[type_ids or 0] + pos[pos_ids or position in sequence]) in one pass
    (bf_embed_layernorm).  ids: [B, L] int64; type_ids: [B, L] or None; pos_ids: [1 or B, L] or None; returns [B, L, N].
    An id outside its table gives a NaN output row (no out-of-bounds read)."""
    _require_device(ids, "embed_layernorm ids")
    B, L = ids.shape
    N = word.shape[1]
    ids = ids.contiguous()
    if type_ids is not None:
        type_ids = type_ids.expand(B, L).contiguous()
    pos_rows = 0
    if pos_ids is not None:
        pos_ids = pos_ids.contiguous()
        pos_rows = pos_ids.numel()  # [1, L]: r % L; [B, L]: r
    out = torch.empty((B, L, N), dtype=word.dtype, device=word.device)
    ptr = lambda t: t.data_ptr() if t is not None else None
    _C.check(_C.lib().bf_embed_layernorm(ids.data_ptr(), ptr(type_ids), ptr(pos_ids), word.data_ptr(), type_table.data_ptr(),
                                         pos_table.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _TORCH2BF[gamma.dtype],
                                         out.data_ptr(), _TORCH2BF[word.dtype], B * L, N, L, pos_rows, word.shape[0],
                                         type_table.shape[0], pos_table.shape[0], float(eps),
                                         _stream_ptr()), "bf_embed_layernorm")
    return out


def attention_supported(q: Tensor, k: Tensor, v: Tensor) -> bool:
    """q, k, v as the attention hook gets them: [B, H, T, 64] views of the projections' [B*T, H*64] outputs."""
    if not (q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and k.dtype == q.dtype and v.dtype == q.dtype):
        return False
    if q.dim() != 4 or q.shape != k.shape or q.shape != v.shape:
        return False
    B, H, T, D = q.shape
    if D != 64 or T < 128 or T % 128 or B > 65535 or H > 65535:
        return False
    st = (T * H * D, D, H * D, 1)  # BHTD view of a contiguous [B, T, H, D] tensor
    return all(t.stride() == st and t.data_ptr() % 16 == 0 for t in (q, k, v))


def attention_forward(q: Tensor, k: Tensor, v: Tensor, key_mask: Optional[Tensor], scaling: float,
                      mask_off: Optional[Tensor] = None, want_lse: bool = False, drop: Optional[Dropout] = None,
                      want_keep: bool = False):
    """softmax(q k^T * scaling + key_mask) v (bf_attention_fwd).  q, k, v: [B, H, T, 64] views as described by
    attention_supported; key_mask: additive fp32 [B, T] or None; mask_off: optional 1-element bool/uint8 device tensor,
    true = the mask is all zeros (the kernel then skips it).  Returns [B, T, H, 64] contiguous — and, with want_lse, the
    [B, H, T] fp32 log-sum-exp rows bf_attention_bwd needs."""
    B, H, T, D = q.shape
    out = torch.empty((B, T, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=q.device) if want_lse else None
    if drop is not None and drop.p > 0.0:
        # attention_probs_dropout in the kernel (bf_attention_fwd_dropout); with want_keep the decisions come back as one
        # bit per probability ([B, H, T, T/32] int32) for bf_attention_bwd_dropout
        keep = torch.empty((B, H, T, T // 32), dtype=torch.int32, device=q.device) if want_keep else None
        _C.check(_C.lib().bf_attention_fwd_dropout(q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                   key_mask.data_ptr() if key_mask is not None else None,
                                                   mask_off.data_ptr() if mask_off is not None else None, out.data_ptr(),
                                                   lse.data_ptr() if lse is not None else None, _TORCH2BF[q.dtype], B, T, H,
                                                   D, H * D, float(scaling), drop.p, drop.seed, drop.call, drop.site,
                                                   drop.first_group(B, H * T * (T // 32) * 4),
                                                   keep.data_ptr() if keep is not None else None, drop.d_call, _stream_ptr()),
                 "bf_attention_fwd_dropout")
        res = (out,) + ((lse,) if want_lse else ()) + ((keep,) if want_keep else ())
        return res if len(res) > 1 else out
    _C.check(_C.lib().bf_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                       key_mask.data_ptr() if key_mask is not None else None,
                                       mask_off.data_ptr() if mask_off is not None else None, out.data_ptr(),
                                       lse.data_ptr() if lse is not None else None,
                                       _TORCH2BF[q.dtype], B, T, H, D, H * D, float(scaling), _stream_ptr()),
             "bf_attention_fwd")
    return (out, lse) if want_lse else out


def attention_backward(q: Tensor, k: Tensor, v: Tensor, key_mask: Optional[Tensor], mask_off: Optional[Tensor],
                       out: Tensor, grad_out: Tensor, lse: Tensor, scaling: float, drop_p: float = 0.0,
                       keep: Optional[Tensor] = None, colsum_samples: int = 0):
    """Gradients of attention_forward (bf_attention_bwd).  Returns (dq, dk, dv), each [B, T, H, 64] contiguous.
    colsum_samples = S (one-tile sequences, B % S == 0): the per-sample column sums of dq / dk / dv come out of the same
    launch (bf_attention_bwd_colsum) and are offered to the backward of the layers that produced q, k, v (offer_colsum)."""
    B, H, T, D = q.shape
    go = grad_out if (grad_out.dtype == q.dtype and grad_out.is_contiguous()) else grad_out.to(q.dtype).contiguous()
    # one buffer, three slabs: the gradients of a stacked query / key / value launch arrive as one [3, ...] tensor
    dqkv = torch.empty((3, B, T, H, D), dtype=q.dtype, device=q.device)
    dq, dk, dv = dqkv[0], dqkv[1], dqkv[2]
    delta = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    if colsum_samples > 0 and T == 128 and B % colsum_samples == 0 and not _NO_COLSUM_FOLD:
        S = int(colsum_samples)
        partial = workspace(q.device, B * H * 3 * D * 4)
        colsum = torch.empty((3, S, H * D), dtype=torch.float32, device=q.device)
        _C.check(_C.lib().bf_attention_bwd_colsum(q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                  key_mask.data_ptr() if key_mask is not None else None,
                                                  mask_off.data_ptr() if mask_off is not None else None, out.data_ptr(),
                                                  go.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                                                  dk.data_ptr(), dv.data_ptr(), _TORCH2BF[q.dtype], B, T, H, D, H * D,
                                                  float(scaling), float(drop_p), keep.data_ptr() if drop_p > 0.0 else None,
                                                  S, partial.data_ptr(), colsum.data_ptr(), _stream_ptr()),
                 "bf_attention_bwd_colsum")
        for t, g in enumerate((dq, dk, dv)):
            offer_colsum(g, colsum[t])
        return dq, dk, dv
    if drop_p > 0.0:
        _C.check(_C.lib().bf_attention_bwd_dropout(q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                   key_mask.data_ptr() if key_mask is not None else None,
                                                   mask_off.data_ptr() if mask_off is not None else None, out.data_ptr(),
                                                   go.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                                                   dk.data_ptr(), dv.data_ptr(), _TORCH2BF[q.dtype], B, T, H, D, H * D,
                                                   float(scaling), float(drop_p), keep.data_ptr(), _stream_ptr()),
                 "bf_attention_bwd_dropout")
        return dq, dk, dv
    _C.check(_C.lib().bf_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                       key_mask.data_ptr() if key_mask is not None else None,
                                       mask_off.data_ptr() if mask_off is not None else None, out.data_ptr(),
                                       go.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                       dv.data_ptr(), _TORCH2BF[q.dtype], B, T, H, D, H * D, float(scaling),
                                       _stream_ptr()), "bf_attention_bwd")
    return dq, dk, dv


class AttentionFn(torch.autograd.Function):
    """bf_attention_fwd with bf_attention_bwd as its backward: nothing but q, k, v, the output and one fp32 row
    statistic per query is kept; the probabilities are recomputed in the backward kernels."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, mask_off, scaling, drop=None):
        ctx.drop_p = drop.p if drop is not None else 0.0
        fwd = bfr.STATE.ctx   # inside an S-sample forward: the backward also leaves dq / dk / dv's per-sample column sums
        ctx.cs_samples = fwd.S if fwd is not None else 0
        if ctx.drop_p > 0.0:  # training mode: probabilities dropped in the kernel, one keep bit each kept for the backward
            out, lse, keep = attention_forward(q, k, v, key_mask, scaling, mask_off, want_lse=True, drop=drop, want_keep=True)
            ctx.save_for_backward(q, k, v, out, lse, keep)
        else:
            out, lse = attention_forward(q, k, v, key_mask, scaling, mask_off, want_lse=True)
            ctx.save_for_backward(q, k, v, out, lse)
        ctx.key_mask, ctx.mask_off, ctx.scaling = key_mask, mask_off, scaling
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, out, lse = ctx.saved_tensors[:5]
        keep = ctx.saved_tensors[5] if ctx.drop_p > 0.0 else None
        dq, dk, dv = attention_backward(q, k, v, ctx.key_mask, ctx.mask_off, out, grad_out, lse, ctx.scaling, ctx.drop_p, keep,
                                        colsum_samples=ctx.cs_samples)
        # q, k, v came in as [B, H, T, 64] views of [B, T, H*64] projections: hand the gradients back in that view
        return dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2), None, None, None, None


def add_layernorm_backward(x: Tensor, residual: Optional[Tensor], gamma: Tensor, grad_out: Tensor, eps: float,
                           drop: Optional[Dropout] = None, grad_out2: Optional[Tensor] = None):
    """Gradients of add_layernorm (bf_add_layernorm_bwd): returns (dz, dgamma, dbeta); dz is the gradient of both x
    and residual, dgamma / dbeta are fp32.  With `drop` (bf_add_layernorm_dropout_bwd) returns (dz, dgamma, dbeta, dx):
    dz is the residual's gradient, dx = dz o keep / (1 - p) the dropped input's.  grad_out2: the gradient of the
    output's second consumer, added to grad_out inside the kernel (bf_add_layernorm_bwd_sum)."""
    N = x.shape[-1]
    x2 = x.reshape(-1, N)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        r2 = r2 if r2.is_contiguous() else r2.contiguous()
    g2 = grad_out.reshape(-1, N)
    g2 = (g2 if g2.dtype == x.dtype else g2.to(x.dtype)).contiguous()
    dz = torch.empty_like(x2)
    dgb = torch.empty((2, N), dtype=torch.float32, device=x.device)  # one buffer: one cast for both in the caller
    dgamma, dbeta = dgb[0], dgb[1]
    lib = _C.lib()
    need = lib.bf_add_layernorm_bwd_workspace_bytes(x2.shape[0], N)
    ws = workspace(x.device, need)
    if grad_out2 is not None:
        h2 = grad_out2.reshape(-1, N)
        h2 = (h2 if h2.dtype == x.dtype else h2.to(x.dtype)).contiguous()
        dropping = drop is not None and drop.p > 0.0
        dx = torch.empty_like(x2) if dropping else None
        _C.check(lib.bf_add_layernorm_bwd_sum(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, gamma.data_ptr(),
                                              _TORCH2BF[gamma.dtype], g2.data_ptr(), h2.data_ptr(), dz.data_ptr(),
                                              dx.data_ptr() if dropping else None, dgamma.data_ptr(), dbeta.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _TORCH2BF[x.dtype], x2.shape[0], N, float(eps),
                                              drop.p if dropping else 0.0, drop.seed if dropping else 0,
                                              drop.call if dropping else 0, drop.site if dropping else 0,
                                              drop.first_group(x2.shape[0], N // 8) if dropping else 0,
                                              drop.d_call if dropping else None, _stream_ptr()),
                 "bf_add_layernorm_bwd_sum")
        return (dz.view(x.shape), dgamma, dbeta, dx.view(x.shape)) if dropping else (dz.view(x.shape), dgamma, dbeta)
    if drop is not None and drop.p > 0.0:
        dx = torch.empty_like(x2)
        _C.check(lib.bf_add_layernorm_dropout_bwd(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, gamma.data_ptr(),
                                                  _TORCH2BF[gamma.dtype], g2.data_ptr(), dz.data_ptr(), dx.data_ptr(),
                                                  dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  _TORCH2BF[x.dtype], x2.shape[0], N, float(eps), drop.p, drop.seed, drop.call,
                                                  drop.site, drop.first_group(x2.shape[0], N // 8), drop.d_call, _stream_ptr()),
                 "bf_add_layernorm_dropout_bwd")
        return dz.view(x.shape), dgamma, dbeta, dx.view(x.shape)
    _C.check(lib.bf_add_layernorm_bwd(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, gamma.data_ptr(),
                                      _TORCH2BF[gamma.dtype], g2.data_ptr(), dz.data_ptr(), dgamma.data_ptr(),
                                      dbeta.data_ptr(), ws.data_ptr(), ws.numel(), _TORCH2BF[x.dtype], x2.shape[0], N,
                                      float(eps), _stream_ptr()), "bf_add_layernorm_bwd")
    return dz.view(x.shape), dgamma, dbeta


_NO_COLSUM_FOLD = os.environ.get("BF_NO_COLSUM_FOLD") is not None  # developer A/B: every bias gradient's column sums by a pass of its own
_NO_TWIN = os.environ.get("BF_NO_LN_TWIN") is not None  # developer A/B: let autograd add the two consumers' gradients


class AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(x + residual) * gamma + beta with both directions in the HIP kernels; nothing but the inputs is saved.

    twin=True returns the output TWICE — (y, an alias of y that shares its storage) — for the callers that know the output
    has two consumers (in a transformer layer: the next dense layer and the next residual connection).  Each consumer takes
    its own alias, so autograd hands backward() the two gradients separately and the kernel adds them on load
    (bf_add_layernorm_bwd_sum) instead of autograd adding them with an activation-sized pass of its own.  Whatever the
    graph looks like the result is the plain one: an alias nobody used has no gradient, further consumers of either alias
    are summed by autograd as always."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, drop=None, twin=False):
        ctx.eps, ctx.has_res = eps, residual is not None
        ctx.drop = drop if (drop is not None and drop.p > 0.0) else None
        ctx.save_for_backward(x, residual if residual is not None else x, gamma)
        y = add_layernorm(x, residual, gamma, beta, eps, ctx.drop)
        if twin:
            ctx.set_materialize_grads(False)  # an unused alias arrives as None, not as a tensor of zeros
            return y, y.detach()
        return y

    @staticmethod
    def backward(ctx, grad_out, grad_twin=None):
        x, residual, gamma = ctx.saved_tensors
        if grad_out is None:
            grad_out, grad_twin = grad_twin, None
        if grad_out is None:
            return None, None, None, None, None, None, None
        dx = None
        if ctx.drop is not None:  # the mask is regenerated from (seed, call, site): nothing was stored
            dz, dgamma, dbeta, dx = add_layernorm_backward(x, residual if ctx.has_res else None, gamma, grad_out, ctx.eps, ctx.drop,
                                                           grad_out2=grad_twin)
        else:
            dz, dgamma, dbeta = add_layernorm_backward(x, residual if ctx.has_res else None, gamma, grad_out, ctx.eps,
                                                       grad_out2=grad_twin)
        need = ctx.needs_input_grad
        if gamma.dtype != torch.float32 and (need[2] or need[3]):
            # dgamma and dbeta are the two rows of one fp32 buffer: cast them with one launch
            both = dgamma._base.to(gamma.dtype) if dgamma._base is not None else torch.stack((dgamma, dbeta)).to(gamma.dtype)
            dgamma, dbeta = both[0], both[1]
        return ((dx if dx is not None else dz) if need[0] else None, dz if (ctx.has_res and need[1]) else None,
                dgamma if need[2] else None, dbeta if need[3] else None, None, None, None)


def layernorm_supported(x: Tensor, residual: Optional[Tensor], ln) -> bool:
    n = x.shape[-1]
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and n % 8 == 0 and n <= 4096 and
            (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype)) and
            ln.weight.dtype in (torch.float32, x.dtype) and ln.bias is not None and ln.bias.dtype == ln.weight.dtype)
