"""The C-ABI library loads without a GPU and exports every symbol include/k82.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "k82.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(k82_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from tts_amd import _lib
    names = declared_functions()
    assert len(names) >= 15 and "k82_durations" in names and "k82_decode" in names
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/k82.h but not exported by libk82.so"
    assert sorted(_lib.EXPORTS) == names


def test_create_without_gpu_fails_cleanly():
    """No GPU here: k82_create must return an error code and a message, never abort."""
    import torch
    if torch.cuda.is_available():
        return
    from tts_amd import _lib, spec
    from tts_amd.engine import _cfg_struct
    lib = _lib.load()
    h = ctypes.c_void_p()
    cfg = _cfg_struct(spec.DEFAULT_CONFIG)
    rc = lib.k82_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc != 0 and len(lib.k82_last_error(None)) > 0


def test_create_rejects_foreign_architecture():
    from tts_amd import _lib, spec
    from tts_amd.engine import _cfg_struct
    lib = _lib.load()
    cfg = _cfg_struct(spec.DEFAULT_CONFIG)
    cfg.hidden_dim = 384
    h = ctypes.c_void_p()
    assert lib.k82_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -22
    assert b"unsupported architecture" in lib.k82_last_error(None)


def test_product_never_imports_the_oracle():
    for dp, _, fns in os.walk(os.path.join(ROOT, "tts_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn), encoding="utf-8").read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn
