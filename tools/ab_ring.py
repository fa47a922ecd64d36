"""A/B of the bf16-mode ring pipeline (conv_mfma.hip, TAPS > 0) against the generic tap loop (K82_RING=0) on ONE box:
bit-identity of durations and audio on a ragged batch, then the configs[1]-shaped step time of both, per precision mode.
usage: python tools/ab_ring.py [bf16 bf16x3]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402

modes = sys.argv[1:] or ["bf16", "bf16x3"]
sds = spec.synthetic_state_dicts(0)
rng = np.random.default_rng(5)
id_lists = [rng.integers(1, 178, size=n).tolist() for n in (37, 90, 12, 200, 65)]
pack = spec.synthetic_voice_pack(1)
ref_s = np.stack([pack[len(x) - 1, 0] for x in id_lists]).astype(np.float32)
B, T = 64, 256
ids64 = np.zeros((B, T), dtype=np.int64)
ids64[:, 1:T - 1] = np.random.default_rng(1000).integers(1, 178, size=(B, T - 2))
lens64 = np.full((B,), T, dtype=np.int32)
ref64 = np.repeat(pack[T - 3], B, axis=0).astype(np.float32)

for mode in modes:
    res = {}
    for ring in ("1", "0"):
        os.environ["K82_RING"] = ring
        eng = Engine(sds, precision=mode)
        ids, lens = eng.pack(id_lists)
        with eng.session():
            pred, nfr = eng.durations(ids, lens, ref_s)
            audio, off = eng.decode(seeds=np.arange(len(id_lists), dtype=np.uint64))
        audio = audio.copy()
        times = []
        for rep in range(3):
            t0 = time.perf_counter()
            with eng.session():
                _, nfr64 = eng.durations(ids64, lens64, ref64)
                a64, _ = eng.decode(seeds=np.arange(B, dtype=np.uint64))
            times.append(time.perf_counter() - t0)
        eng.prof_enable(True)
        with eng.session():
            eng.durations(ids64, lens64, ref64)
            eng.decode(seeds=np.arange(B, dtype=np.uint64))
        ms, n, fl = eng.prof_read("conv_mfma_2x2")
        eng.prof_enable(False)
        res[ring] = (pred, audio, min(times), ms, fl / (ms * 1e-3) / 1e12, a64[:100000].copy())
        eng.close()
    same = bool(np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][5], res["0"][5]))
    print(json.dumps({"mode": mode, "bit_identical_ring_vs_generic": same,
                      "max_abs_diff": float(np.abs(res["1"][1] - res["0"][1]).max()),
                      "step_s_ring_incl_d2h": res["1"][2], "step_s_generic_incl_d2h": res["0"][2],
                      "conv_ms_ring": res["1"][3], "conv_ms_generic": res["0"][3],
                      "conv_TFLOPs_ring": res["1"][4], "conv_TFLOPs_generic": res["0"][4]}), flush=True)
