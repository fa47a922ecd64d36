"""cfg5 breakdown: one sentence at a time (B=1), where the milliseconds go.  usage: prof_latency.py [dtype] [n]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec
from tts_amd.engine import Engine
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
eng = Engine(spec.synthetic_state_dicts(0), precision=dtype)
rng = np.random.default_rng(0)
pack = spec.synthetic_voice_pack(1)
T = 100
ids = [list(rng.integers(1, 178, size=T)) for _ in range(n + 3)]
fd = np.full((1, T + 2), 4, np.int32)          # 4 frames per token -> 408 frames, 10.2 s of audio
acc = {"dur": [], "dec": []}
fam = ("conv_mfma_2x2", "conv_mfma_1x4", "lstm", "instnorm", "attention", "istft", "gather", "source", "other")
for i, x in enumerate(ids):
    if i == 3:
        eng.prof_enable(True)
    i2, lens = eng.pack([x])
    t0 = time.perf_counter()
    eng.durations(i2, lens, pack[T - 1], forced_dur=fd)
    t1 = time.perf_counter()
    eng.decode(seeds=np.array([i], np.uint64))
    t2 = time.perf_counter()
    if i >= 3:
        acc["dur"].append((t1 - t0) * 1e3); acc["dec"].append((t2 - t1) * 1e3)
out = {"dtype": dtype, "tokens": T + 2, "frames": 4 * (T + 2), "durations_ms_p50": float(np.median(acc["dur"])),
       "decode_ms_p50": float(np.median(acc["dec"])), "decode_gpu_ms_by_family": {}}
for f in fam:
    ms, cnt, _ = eng.prof_read(f)
    out["decode_gpu_ms_by_family"][f] = {"ms": ms / n, "launches": cnt / n}
print(json.dumps(out))
