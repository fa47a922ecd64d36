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


def test_bench_refuses_a_rank_count_it_cannot_deliver():
    """bench.py --gpus N must never silently measure fewer GPUs: launched plainly with N = 2 on a
    machine without GPUs it starts two ranks, both refuse (no GPU for the rank), and the launcher exits non-zero
    without a JSON line; with a WORLD_SIZE that disagrees with --gpus it names the mismatch."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        return
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout and "needs GPU" in r.stderr and "--gpus 2" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr
