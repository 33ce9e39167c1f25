"""ctypes binding of ``libglowk.so`` (the C ABI declared in ``include/glowk.h``).

The product path has no CPU fallback: if the HIP library has not been built (``__graft_entry__.build()``
or ``python -m audiosourcesep_amd.build``) importing a compute entry point raises ``GlowkLibraryMissing``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLOWK_LIB") or os.path.join(_HERE, "libglowk.so")   # GLOWK_LIB: A/B timing of two builds (scripts/ab.py)


class GlowkLibraryMissing(RuntimeError):
    pass


class GlowkError(RuntimeError):
    pass


class GlowkRangeError(GlowkError):
    """GLOWK_ERR_RANGE: a call in a split arithmetic (f16x3 / f16x2) left the fp16 range; its outputs are not usable."""


class GlowkConfigStruct(ctypes.Structure):
    """``glowk_config`` of include/glowk.h (field order and types must match)."""
    _fields_ = [
        ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
        ("L", ctypes.c_int32), ("K", ctypes.c_int32), ("F", ctypes.c_int32),
        ("learntop", ctypes.c_int32), ("use_logit", ctypes.c_int32),
        ("minval", ctypes.c_float), ("maxval", ctypes.c_float), ("alpha", ctypes.c_float),
        ("bn_eps", ctypes.c_float),
    ]


class GlowkProfile(ctypes.Structure):
    """``glowk_profile`` of include/glowk.h."""
    _fields_ = [("net_ms", ctypes.c_double * 4), ("net_launches", ctypes.c_int64 * 4)]


# tensor ids (enum glowk_tensor_id) keyed by the flat parameter names used across the repo
STEP_TENSOR_IDS = {
    "actnorm/log_scale": 0, "actnorm/shift": 1,
    "inv1x1/P": 2, "inv1x1/sign_S": 3, "inv1x1/L": 4, "inv1x1/log_S": 5, "inv1x1/U": 6,
    "nn/conv1/kernel": 7, "nn/conv1/bias": 8,
    "nn/bn1/gamma": 9, "nn/bn1/beta": 10, "nn/bn1/mean": 11, "nn/bn1/var": 12,
    "nn/conv2/kernel": 13, "nn/conv2/bias": 14,
    "nn/bn2/gamma": 15, "nn/bn2/beta": 16, "nn/bn2/mean": 17, "nn/bn2/var": 18,
    "nn/conv3/kernel": 19, "nn/conv3/bias": 20,
    "inv1x1/P_inv": 21,
}
PRIOR_TENSOR_IDS = {"prior/loc": 100, "prior/log_scale": 101}
PREC_F32, PREC_F16X3, PREC_F16X2 = 0, 1, 2
OK, ERR, ERR_RANGE = 0, 1, 2                       # enum glowk_status
RANGE_IGNORE, RANGE_ERROR, RANGE_FALLBACK = 0, 1, 2   # enum glowk_range_policy

_vp, _i, _fp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)

# every symbol include/glowk.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "glowk_version": (_i, []),
    "glowk_last_error": (ctypes.c_char_p, []),
    "glowk_reload_env": (None, []),
    "glowk_debug_stamps": (_i, [ctypes.POINTER(ctypes.c_uint64), _i]),
    "glowk_create": (_i, [ctypes.POINTER(GlowkConfigStruct), _i, ctypes.POINTER(_vp)]),
    "glowk_destroy": (_i, [_vp]),
    "glowk_tensor_size": (ctypes.c_size_t, [_vp, _i, _i]),
    "glowk_set_tensor": (_i, [_vp, _i, _i, _i, _fp, ctypes.c_size_t]),
    "glowk_get_tensor": (_i, [_vp, _i, _i, _i, _fp, ctypes.c_size_t]),
    "glowk_finalize_weights": (_i, [_vp]),
    "glowk_actnorm_data_init": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "glowk_set_precision": (_i, [_vp, _i]),
    "glowk_get_precision": (_i, [_vp]),
    "glowk_set_range_policy": (_i, [_vp, _i]),
    "glowk_get_range_policy": (_i, [_vp]),
    "glowk_range_status": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_int64), _vp]),
    "glowk_range_probe_begin": (_i, [_vp]),
    "glowk_range_probe_end": (_i, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _vp]),
    "glowk_workspace_bytes": (ctypes.c_size_t, [_vp, _i, _i]),
    "glowk_reserve": (_i, [_vp, _i, _i]),
    "glowk_max_tiles": (_i, [_vp]),
    "glowk_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "glowk_inverse": (_i, [_vp, _vp, _i, _vp, _vp]),
    "glowk_log_prob": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "glowk_log_prob_sum": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "glowk_sum_f64": (_i, [_vp, ctypes.c_size_t, _vp, _i, ctypes.c_double, _vp]),
    "glowk_log_prob_grad": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "glowk_sample": (_i, [_vp, _vp, _i, _vp, _vp]),
    "glowk_prior_log_prob": (_i, [_vp, _vp, _i, _vp, _vp]),
    "glowk_fused_steps": (ctypes.c_int64, [_vp]),
    "glowk_kernel_families": (_i, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "glowk_profile_begin": (_i, [_vp]),
    "glowk_profile_end": (_i, [_vp, ctypes.POINTER(GlowkProfile)]),
    "glowk_squeeze": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "glowk_unsqueeze": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "glowk_preprocess_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "glowk_preprocess_inverse": (_i, [_vp, _vp, _i, _vp, _vp]),
    "glowk_step_forward": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "glowk_step_inverse": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "glowk_coupling_net": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "glowk_param_vector_size": (ctypes.c_size_t, [_vp]),
    "glowk_param_offset": (_i, [_vp, _i, _i, _i, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "glowk_param_grad": (_i, [_vp, _vp, _i, ctypes.c_float, _vp, _vp, _vp]),
    "glowk_apply_gradients": (_i, [_vp, _vp, _i, ctypes.c_float, _vp]),
    "glowk_crc32c": (ctypes.c_uint32, [_vp, ctypes.c_size_t]),
    "glowk_basis_update": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, _vp, _vp,
                                ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
    "glowk_basis_mix": (_i, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "glowk_random": (_i, [_vp, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, _i, _i, ctypes.c_uint64, _vp]),
    "glowk_add_noise": (_i, [_vp, _vp, ctypes.c_size_t, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _i, ctypes.c_uint64, _vp]),
}

_lib = None


def load():
    """Load libglowk.so once and declare every prototype.  Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GlowkLibraryMissing(
            "HIP library %s not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the compute path." % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64; if libglowk.so is loaded before it, the system HIP runtime gets
    # bound instead and the two runtimes in one process end in "no ROCm-capable device is detected" at the first hipSetDevice
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().glowk_last_error()
        raise (GlowkRangeError if rc == ERR_RANGE else GlowkError)(msg.decode() if msg else "glowk error %d" % rc)
