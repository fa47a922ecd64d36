"""The one entry point that allocates nothing: k82_pcm16_static (include/k82.h) against k82_pcm16.

Collected last on purpose (the round's end runs `pytest -x`).  Two calls in a row through the C-ABI, the second
with a smaller input -- the static staging array keeps the first call's contents, so stale data would show -- each
checked bit for bit against the existing entry point and against the reference's own expression
(basic.tts_kokoro_2_play_pause_works.py:25: clip(x, -1, 1) * 32767 -> '<i2').
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K82_PCM16_STATIC_MAX = 4 * 1024 * 1024


def _ref(x):
    return (np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2")


def test_pcm16_static_matches_pcm16_twice_second_call_smaller(engine):
    rng = np.random.default_rng(77)
    big = (rng.standard_normal(1_000_003) * 0.8).astype(np.float32)
    small = (rng.standard_normal(4_099) * 0.8).astype(np.float32)
    for x in (big, small):
        want = engine.pcm16(x)                               # existing entry point (arena staging)
        got = np.full(x.size + 8, 0x5a5a, dtype=np.int16)     # canary beyond the end
        engine.pcm16_static(x, out=got)
        assert np.array_equal(got[:x.size], want)
        assert np.array_equal(got[:x.size], _ref(x))
        assert (got[x.size:] == 0x5a5a).all()
    # the first call again: nothing of the smaller call lingers in what the larger one reads
    assert np.array_equal(engine.pcm16_static(big), _ref(big))


def test_pcm16_static_device_pointers_and_limits(engine):
    import torch
    x = torch.linspace(-1.5, 1.5, 70_001, dtype=torch.float32, device="cuda")
    y = torch.zeros(70_001, dtype=torch.int16, device="cuda")
    lib, h = engine.lib, engine.h
    assert lib.k82_pcm16_static(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), None) == 0
    assert np.array_equal(y.cpu().numpy(), _ref(x.cpu().numpy()))
    # exactly the maximum is accepted, one more is refused with a clear error and writes nothing
    full = np.zeros(K82_PCM16_STATIC_MAX, dtype=np.float32)
    full[-1] = 1.0
    out = engine.pcm16_static(full)
    assert out[-1] == 32767 and out[0] == 0
    too_big = np.zeros(K82_PCM16_STATIC_MAX + 1, dtype=np.float32)
    sink = np.full(too_big.size, 7, dtype=np.int16)
    with pytest.raises(ValueError, match="at most"):
        engine.pcm16_static(too_big, out=sink)
    assert (sink == 7).all()
    assert engine.pcm16_static(np.zeros(0, np.float32)).size == 0
