"""ctypes binding of libk82.so (include/k82.h).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# K82_LIB_PATH: A/B timing of kernel variants only (tools/ab_conv.sh); the default is the in-tree build
LIB_PATH = os.environ.get("K82_LIB_PATH") or os.path.join(_HERE, "libk82.so")

K82_OK, K82_ENOSPC = 0, -28

EXPORTS = (
    "k82_create", "k82_fork", "k82_destroy", "k82_last_error", "k82_set_tensor", "k82_finalize", "k82_durations",
    "k82_decode", "k82_synthesize", "k82_reserve", "k82_pcm16", "k82_pcm16_static", "k82_set_precision", "k82_prof_enable", "k82_prof_read", "k82_debug_enable",
    "k82_debug_info", "k82_debug_copy", "k82_debug_force", "k82_op_conv1d",
)


class K82Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_token", "hidden_dim", "style_dim", "n_layer", "max_dur", "text_kernel", "bert_hidden",
        "bert_heads", "bert_inter", "bert_layers", "bert_max_pos", "bert_embed")]


class K82Noise(C.Structure):
    _fields_ = [("noise", C.c_void_p), ("seeds", C.c_void_p)]


_lib = None


def load():
    """Load libk82.so once.  Raises RuntimeError (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C tts_amd/csrc` (or __graft_entry__.build()). "
            "tts_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HIP runtimes in
    # one process cannot both own the GPU ("no ROCm-capable device is detected"), so torch is
    # imported FIRST: libk82.so's DT_NEEDED libamdhip64.so.7 then binds to the copy already loaded.
    # Without torch (a plain C consumer) the system runtime under /opt/rocm/lib is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    lib.k82_create.argtypes = [C.POINTER(K82Config), i32, C.POINTER(vp)]
    lib.k82_destroy.argtypes = [vp]
    lib.k82_last_error.argtypes = [vp]
    lib.k82_last_error.restype = C.c_char_p
    lib.k82_set_tensor.argtypes = [vp, C.c_char_p, f32p, C.POINTER(i64), i32]
    lib.k82_finalize.argtypes = [vp]
    lib.k82_durations.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.k82_decode.argtypes = [vp, C.POINTER(K82Noise), vp, i64, vp, vp]
    lib.k82_synthesize.argtypes = [vp, vp, vp, i32, i32, vp, vp, C.POINTER(K82Noise), vp, vp, i64, vp, vp]
    lib.k82_set_precision.argtypes = [vp, C.c_int]
    lib.k82_reserve.argtypes = [vp, i32, i32, i64]
    lib.k82_pcm16.argtypes = [vp, vp, vp, i64, vp]
    lib.k82_pcm16_static.argtypes = [vp, vp, vp, i64, vp]
    lib.k82_prof_enable.argtypes = [vp, i32]
    lib.k82_prof_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.k82_debug_enable.argtypes = [vp, i32]
    lib.k82_debug_info.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    lib.k82_debug_copy.argtypes = [vp, C.c_char_p, vp]
    lib.k82_debug_force.argtypes = [vp, C.c_char_p, vp, i64, i64]
    lib.k82_op_conv1d.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp, i32,
                                  C.POINTER(C.c_float)]
    for n in EXPORTS:
        if n != "k82_last_error":
            getattr(lib, n).restype = C.c_int
    _lib = lib
    return lib
