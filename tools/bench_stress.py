"""cfg4 of SURVEY.md 8d on one GPU: maximum-context stress.  B sentences of 510 phoneme ids (+2 pads),
durations FORCED to a constant (default 4 frames/token -> F = 2048, 51.2 s of audio per sentence), so the
frame-rate work is reproducible with random weights.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--dur", type=int, default=4)
ap.add_argument("--steps", type=int, default=2)
args = ap.parse_args()
B, T = args.batch, 512
rng = np.random.default_rng(0)
ids = np.zeros((B, T), np.int64)
ids[:, 1:511] = rng.integers(1, 178, size=(B, 510))
lens = np.full(B, T, np.int32)
pack = spec.synthetic_voice_pack(1)
ref_s = np.repeat(pack[509], B, axis=0).astype(np.float32)
forced = np.full((B, T), args.dur, np.int32)
eng = Engine(spec.synthetic_state_dicts(0))
seeds = np.arange(B, dtype=np.uint64)
ts = []
for it in range(args.steps + 1):
    t0 = time.perf_counter()
    pred, nfr = eng.durations(ids, lens, ref_s, forced_dur=forced)
    audio, off = eng.decode(seeds=seeds)
    ts.append(time.perf_counter() - t0)
assert (pred == args.dur).all() and int(nfr.sum()) == B * T * args.dur and np.isfinite(audio).all()
dt = float(np.mean(ts[1:]))
a = len(audio) / 24000.0
print(json.dumps({"config": f"cfg4 (1 GPU, fp32): {B} x 510 tokens, forced {args.dur} frames/token", "frames": int(nfr.sum()),
                  "audio_s_per_step": a, "s_per_step": dt, "audio_s_per_s": a / dt,
                  "includes": "H2D ids, D2H of the whole waveform (%.1f GB)" % (audio.nbytes / 1e9), "audio_finite": True}))
eng.close()
