"""ctypes binding of libcellvit_amd.so (the C ABI declared in include/cellvit_amd.h).

There is deliberately NO fallback: if the shared library is missing or a call fails the product
path raises.  The library is built in-tree by ``python -m cellvit_amd.build`` (also called from
``__graft_entry__.build()``) so that it travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcellvit_amd.so")
_EXP_DIR = os.path.join(os.path.dirname(HERE), "build", "experimental")      # experiment libraries never sit in the package
if os.environ.get("CVA_LIB") == "abl":      # experiment flavour (`python -m cellvit_amd.build --ablation`): the one that honours CVA_* switches
    LIB_PATH = os.path.join(_EXP_DIR, "libcellvit_amd_abl.so")
elif os.environ.get("CVA_LIB", "").endswith(".so"):     # a copy of an earlier build under build/experimental/ (same-call A/B runs of tools/)
    LIB_PATH = os.path.join(_EXP_DIR, os.path.basename(os.environ["CVA_LIB"]))

CV_OK, CV_ERR_INVALID, CV_ERR_HIP, CV_ERR_STATE, CV_ERR_SHAPE, CV_ERR_UNSUPPORTED, CV_ERR_MISSING = range(7)
DTYPE_F16, DTYPE_F32, DTYPE_F8 = 0, 1, 2


class cv_config(C.Structure):
    _fields_ = [
        ("arch", C.c_int32), ("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32),
        ("mlp_ratio", C.c_int32), ("extract_layers", C.c_int32 * 4),
        ("num_nuclei_classes", C.c_int32), ("num_tissue_classes", C.c_int32), ("regression_loss", C.c_int32),
        ("patch_size", C.c_int32), ("window_size", C.c_int32), ("n_global", C.c_int32),
        ("global_attn_indexes", C.c_int32 * 8), ("neck_chans", C.c_int32), ("compute_dtype", C.c_int32),
    ]


class cv_outputs(C.Structure):
    _fields_ = [
        ("tissue_types", C.c_void_p), ("nuclei_binary_map", C.c_void_p), ("hv_map", C.c_void_p),
        ("nuclei_type_map", C.c_void_p), ("regression_map", C.c_void_p), ("tokens_nhwc", C.c_void_p),
        ("binary_argmax", C.c_void_p), ("type_argmax", C.c_void_p),
    ]


# name -> (restype, argtypes); the CPU test suite checks that every one of these resolves.
SYMBOLS = {
    "cv_last_error": (C.c_char_p, []),
    "cv_build_is_ablation": (C.c_int, []),
    "cv_build_flags": (C.c_char_p, []),
    "cv_create": (C.c_int, [C.POINTER(cv_config), C.POINTER(C.c_void_p)]),
    "cv_destroy": (C.c_int, [C.c_void_p]),
    "cv_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "cv_finalize": (C.c_int, [C.c_void_p]),
    "cv_set_geometry": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cv_set_derived": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "cv_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(cv_outputs), C.c_void_p]),
    "cv_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                C.POINTER(cv_outputs), C.c_void_p]),
    "cv_op_argmax_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_op_normalize_u8": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "cv_pool_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cv_mx8_quantize_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cv_op_linear_mx8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_op_layernorm_mx8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "cv_geometry_flags": (C.c_int, [C.c_void_p]),
    "cv_op_attention_rows_mx8": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]),
    "cv_op_attention_mx8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_set_debug": (C.c_int, [C.c_void_p, C.c_int]),
    "cv_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "cv_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cv_op_linear": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_op_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_void_p]),
    "cv_op_layernorm_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_float, C.c_void_p]),
    "cv_op_conv3x3": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_op_convT2x2": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_op_attention": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cv_op_deconv_block": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 7 + [C.c_void_p]),
}

SYMBOLS.update({
    "cv_stream_wait_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cv_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "cv_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "cv_pp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cv_pp_destroy": (C.c_int, [C.c_void_p]),
    "cv_pp_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cv_pp_run_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cv_pp_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "cv_pp_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "cv_stitch_overlaps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.POINTER(C.c_int32), C.c_void_p]),
    "cv_stitch_ring_flags": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cv_stitch_repair_rings": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "cv_write_cells_json": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int] + [C.c_void_p] * 11),
    "cv_write_geojson": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "cv_render_cells": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 11 + [C.POINTER(C.c_void_p)]),
    "cv_textbuf_compact": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "cv_textbuf_free": (None, [C.c_void_p]),
    "cv_write_rows": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32),
                                C.POINTER(C.c_int64)]),
    "cv_stitch_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.POINTER(C.c_int32), C.c_void_p]),
})


class cv_instance(C.Structure):
    _fields_ = [("id", C.c_int32), ("rmin", C.c_int32), ("cmin", C.c_int32), ("rmax", C.c_int32), ("cmax", C.c_int32),
                ("npix", C.c_int32), ("type", C.c_int32), ("contour_off", C.c_int32), ("contour_len", C.c_int32),
                ("reserved", C.c_int32), ("cx", C.c_double), ("cy", C.c_double), ("type_prob", C.c_double)]


# numpy view of a cv_instance array copied from the device (64 bytes per record)
import numpy as _np  # noqa: E402
REC_DTYPE = _np.dtype([("id", "<i4"), ("rmin", "<i4"), ("cmin", "<i4"), ("rmax", "<i4"), ("cmax", "<i4"), ("npix", "<i4"),
                       ("type", "<i4"), ("contour_off", "<i4"), ("contour_len", "<i4"), ("reserved", "<i4"),
                       ("cx", "<f8"), ("cy", "<f8"), ("type_prob", "<f8")])
assert REC_DTYPE.itemsize == C.sizeof(cv_instance)

_lib = None


def load() -> C.CDLL:
    """Load the HIP extension; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64: import it FIRST so that this library binds to the same HIP
    # runtime instance (two runtimes in one process do not share devices/streams).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension with `python -m cellvit_amd.build` "
            "(cellvit_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the ABI and the binding drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_EXC = {
    CV_ERR_INVALID: ValueError, CV_ERR_HIP: RuntimeError, CV_ERR_STATE: RuntimeError,
    CV_ERR_SHAPE: AssertionError, CV_ERR_UNSUPPORTED: NotImplementedError, CV_ERR_MISSING: RuntimeError,
}


def check(rc: int) -> None:
    """Map a C status to the exception type the reference raises at the same place."""
    if rc == CV_OK:
        return
    msg = load().cv_last_error()
    msg = msg.decode("utf-8", "replace") if msg else f"cellvit_amd error {rc}"
    raise _EXC.get(rc, RuntimeError)(msg)
