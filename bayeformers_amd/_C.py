"""ctypes binding of the C-ABI library (include/bayeformers_amd.h -> lib/libbayeformers_amd.so).

The product path has no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BF_LIB_PATH") or os.path.join(_HERE, "lib", "libbayeformers_amd.so")  # BF_LIB_PATH: developer A/B of kernel variants

BF_DT_F32, BF_DT_BF16, BF_DT_F16 = 0, 1, 2
BF_PRIOR_MIXTURE, BF_PRIOR_GAUSSIAN, BF_PRIOR_NONE = 0, 1, 2


class bf_prior_t(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("pi", ctypes.c_float),
        ("sigma1", ctypes.c_float),
        ("sigma2", ctypes.c_float),
        ("d_mu", ctypes.c_void_p),
        ("d_rho", ctypes.c_void_p),
        ("d_pi", ctypes.c_void_p),      # mixture: the device scalars pi / sigma1 / sigma2 were read from (re-checked by
        ("d_sigma1", ctypes.c_void_p),  # the kernels, bf_stale_counter)
        ("d_sigma2", ctypes.c_void_p),
    ]


class bf_tensor_t(ctypes.Structure):
    _fields_ = [
        ("d_mu", ctypes.c_void_p),
        ("d_rho", ctypes.c_void_p),
        ("n", ctypes.c_uint64),
        ("prior", bf_prior_t),
        ("stream_id", ctypes.c_uint32),
        ("out_dtype", ctypes.c_int32),
        ("d_sample_out", ctypes.c_void_p),
    ]


# every symbol include/bayeformers_amd.h declares: name -> (restype, argtypes)
_vp, _i, _u32, _u64, _i64, _sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int64,
                                  ctypes.c_size_t)
_tp = ctypes.POINTER(bf_tensor_t)
SYMBOLS = {
    "bf_version": (_i, []),
    "bf_last_error": (ctypes.c_char_p, []),
    "bf_stale_counter": (_i, [ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]),
    "bf_set_sample_counter": (_i, [_vp]),
    "bf_get_sample_counter": (_vp, []),
    "bf_device_info": (_i, [ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bf_philox_normal_host": (_i, [_vp, _u64, _u64, _u32, _u32, _u64]),
    "bf_philox_normal": (_i, [_vp, _u64, _i, _u64, _u32, _u32, _vp]),
    "bf_sample_logprob_workspace_bytes": (_sz, [_tp, _i, _i]),
    "bf_sample_logprob": (_i, [_tp, _i, _i, _u64, _u32, _vp, _vp, _sz, _vp]),
    "bf_sample_table_bytes": (_sz, [_tp, _i, ctypes.POINTER(ctypes.c_uint32)]),
    "bf_sample_table_build": (_i, [_tp, _i, _vp, _sz, _vp, _vp]),
    "bf_sample_logprob_table": (_i, [_vp, _i, _u32, _u32, _i, _u64, _u32, _vp, _i, _vp]),
    "bf_reduce_logprob": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "bf_gemm_nt": (_i, [_vp, _i, _i64, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_nt_act": (_i, [_vp, _i, _i64, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_nt_act_pre": (_i, [_vp, _i, _i64, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_nt_layers": (_i, [_vp, _i, _i64, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_nn_layers": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_nn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_tn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "bf_gemm_prepare": (_i, [_i, _i, _i, _i, _vp]),
    "bf_gemm_schedule": (_sz, [_i, _i, _i, _i, _i, _vp, _sz, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bf_gemm_schedule_policy": (_sz, [_i, _i, _i, _i, _i, _i, _vp, _sz, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bf_gemm_schedule_fetch_rows": (ctypes.c_int64, [_vp, _i, _i]),
    "bf_linear_fwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "bf_linear_fwd": (_i, [_vp, _i, _i64, _tp, _tp, _vp, _i, _i, _i, _i, _i, _i, _u64, _u32, _vp, _vp, _sz, _vp]),
    "bf_linear_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "bf_linear_bwd": (_i, [_vp, _i64, _vp, _i, _tp, _tp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _u64, _u32, _i, _vp,
                           _vp, _vp, _vp, _vp, _sz, _vp]),
    "bf_linear_bwd_splits": (_i, [_i, _i, _i, _i, _i]),
    "bf_gemm_nn_actgrad_supported": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "bf_gemm_nn_actgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bf_param_grad_table_bytes": (_sz, [_vp, _i, ctypes.POINTER(ctypes.c_uint32)]),
    "bf_param_grad_table_build": (_i, [_vp, _i, _vp, _sz]),
    "bf_param_grad_table": (_i, [_vp, _i, _u32, _i, _u64, _u32, _vp]),
    "bf_kl_grad": (_i, [_tp, _i, _u64, _u32, _vp, _vp, _vp, _vp]),
    "bf_embedding_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i64, _i, _u64, _u32, _u32, _vp]),
    "bf_embedding_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i64, _i64, _i, _u64, _u32, _u32, _vp]),
    "bf_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, ctypes.c_float, _vp]),
    "bf_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64,
                              ctypes.c_float, _vp]),
    "bf_add_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "bf_add_layernorm_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i64, _i, ctypes.c_float, _vp]),
    "bf_embed_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _i, _i64, _i64, _i64, _i64, ctypes.c_float, _vp]),
    "bf_add_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, ctypes.c_float, _vp]),
    "bf_dropout_keep_host": (_i, [_vp, _u64, _u64, ctypes.c_float, _u64, _u32, _u32]),
    "bf_attention_fwd_dropout": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, ctypes.c_float,
                                      ctypes.c_float, _u64, _u32, _u32, _u64, _vp, _vp, _vp]),
    "bf_attention_bwd_dropout": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64,
                                      ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "bf_add_layernorm_dropout": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, ctypes.c_float, ctypes.c_float, _u64, _u32,
                                      _u32, _u64, _vp, _vp]),
    "bf_add_layernorm_dropout_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i64, _i, ctypes.c_float,
                                          ctypes.c_float, _u64, _u32, _u32, _u64, _vp, _vp]),
    "bf_add_layernorm_bwd_sum": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i64, _i, ctypes.c_float,
                                      ctypes.c_float, _u64, _u32, _u32, _u64, _vp, _vp]),
    "bf_attention_bwd_colsum": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64,
                                     ctypes.c_float, ctypes.c_float, _vp, _i, _vp, _vp, _vp]),
    "bf_profile_enable": (_i, [_i]),
    "bf_profile_reset": (_i, []),
    "bf_probe_stream_read": (_i, [_vp, _sz, _vp, _vp]),
    "bf_fused_small_max_rows": (_i, []),
    "bf_set_fused_small_max_rows": (_i, [_i]),
    "bf_fused_small_rows_for": (_i, [_i, _i]),
    "bf_profile_read": (_i, [_i, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                             ctypes.POINTER(ctypes.c_double)]),
    "bf_profile_read_launches": (_sz, [_i, _vp, _vp, _sz]),
}
class bf_pgrad_t(ctypes.Structure):
    _fields_ = [("d_dw", ctypes.c_void_p), ("d_rho", ctypes.c_void_p), ("d_dmu", ctypes.c_void_p), ("d_drho", ctypes.c_void_p),
                ("n", ctypes.c_uint64), ("stream_id", ctypes.c_uint32), ("splits", ctypes.c_int32)]


BF_PROF_SAMPLE, BF_PROF_GEMM, BF_PROF_FUSED_SMALL, BF_PROF_FUSED_WS = 0, 1, 2, 3
BF_ACT_NONE, BF_ACT_GELU = 0, 1

ABI_VERSION = 6  # bf_version() of the library these bindings describe (include/bayeformers_amd.h: BF_VERSION_*)

# developer-build entry points (csrc/bf_dev_api.h): bound when the loaded library has them (BF_LIB_PATH=..._dev.so)
DEV_SYMBOLS = {
    "bf_linear_fwd_ws_workspace_bytes": (_sz, [_i, _i]),
    "bf_linear_fwd_ws": (_i, [_vp, _i, _i64, _tp, _tp, _vp, _i, _i, _i, _i, _i, _i, _u64, _u32, _i, _vp, _vp, _sz, _vp]),
}

_lib = None


class BayeFormersAMDError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BayeFormersAMDError(
                f"{LIB_PATH} is missing: build it with `python -m bayeformers_amd.build` "
                "(there is no CPU or PyTorch fallback for the Monte-Carlo forward path)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DEV_SYMBOLS.items():
            fn = getattr(l, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        if l.bf_version() != ABI_VERSION:  # a stale .so called through newer signatures corrupts the stack: refuse it
            raise BayeFormersAMDError(
                f"{LIB_PATH} is version {l.bf_version()}, these bindings expect {ABI_VERSION}: rebuild it with "
                "`python -m bayeformers_amd.build`")
        _lib = l
    return _lib


_stale = None


def stale_counter() -> int:
    """Current value of the library's stale-prior counter (bf_stale_counter): a host-memory word the kernels bump when a
    prior's baked constants fail their device-side re-check.  Reading it does not synchronise."""
    global _stale
    if _stale is None:
        p = ctypes.POINTER(ctypes.c_uint32)()
        if lib().bf_stale_counter(ctypes.byref(p)) != 0:
            return 0  # no HIP device in this process (host-only tests): nothing can have bumped it
        _stale = p
    return int(_stale[0])


def check(rc, what):
    if rc != 0:
        msg = lib().bf_last_error()
        raise BayeFormersAMDError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
