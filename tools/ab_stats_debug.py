"""Stage-by-stage comparison of two engines on one batch: K82_FUSED_STATS=1 (epilogue partials) vs 0 (sweep kernel)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from tts_amd import spec
from tts_amd.engine import Engine
lengths = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [37, 90, 12, 254, 65, 33]
id_lists, ref_s = H.make_batch(7, lengths)
names = sys.argv[2].split(",")
out = {}
for f in ("1", "0"):
    os.environ["K82_FUSED_STATS"] = f
    e = Engine(spec.synthetic_state_dicts(0))
    e.debug_enable(True)
    ids, lens = e.pack(id_lists)
    pred, nfr = e.durations(ids, lens, ref_s)
    audio, off = e.decode(seeds=list(range(len(lengths))))
    out[f] = {n: e.debug_get(n).copy() for n in names}
    out[f]["audio"] = audio.copy()
    out[f]["nfr"] = nfr.copy()
    e.close()
print("frames", out["1"]["nfr"])
for n in names + ["audio"]:
    a, b = out["1"][n], out["0"][n]
    if n == "audio":
        print(n, float(np.abs(a - b).max()))
        continue
    for s in range(a.shape[0]):
        mult = {"F0": 2, "N": 2, "dec_encode": 1, "dec_out": 2, "gen0_noise_conv": 20, "gen0_xsource": 20, "gen0_out": 20,
                "gen1_xsource": 120, "gen1_out": 120, "conv_post": 120}.get(n)
        L = a.shape[2] if mult is None else min(a.shape[2], int(out["1"]["nfr"][s]) * mult)
        d = np.abs(a[s][:, :L] - b[s][:, :L])
        d = np.where(np.isfinite(d), d, 0)
        print(n, "sentence", s, "max diff", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape), "scale", float(np.abs(np.where(np.isfinite(b[s]), b[s], 0)).max()))
