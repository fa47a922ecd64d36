"""Are two builds of the library bit-identical stage by stage?  GPU box:  python tools/ab_stage_bits.py <libA.so> <libB.so> [T ...]
Each library runs in its own child process (one HIP runtime binding per process); the parent compares the debug taps."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = ["bert_out", "d_en", "d", "dur_lstm", "duration", "en", "shared", "f0_blk0", "F0", "N", "t_en", "asr", "dec_encode", "dec_out", "gen1_out", "conv_post"]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from tts_amd import spec
    from tts_amd.engine import Engine
    out, Ts = sys.argv[2], [int(v) for v in sys.argv[3:]]
    eng = Engine(spec.synthetic_state_dicts(0))
    eng.debug_enable(True)
    res = {}
    for T in Ts:
        rng = np.random.default_rng(900 + T)
        ids = [rng.integers(1, 178, size=T - 2).tolist()]
        ref = spec.synthetic_voice_pack(1)[T - 3, 0][None].astype(np.float32)
        i2, lens = eng.pack(ids)
        with eng.session():
            eng.durations(i2, lens, ref)
            res.update({f"T{T}_{k}": eng.debug_get(k) for k in STAGES[:5]})
            a, _ = eng.decode(seeds=np.array([3], np.uint64))
            res.update({f"T{T}_{k}": eng.debug_get(k) for k in STAGES[5:]})
            res[f"T{T}_audio"] = a.copy()
    np.savez(out, **res)
    sys.exit(0)
libs, Ts = sys.argv[1:3], (sys.argv[3:] or ["8", "64", "256"])
files = []
for lib in libs:
    f = tempfile.mktemp(suffix=".npz")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f] + Ts, env=dict(os.environ, K82_LIB_PATH=os.path.abspath(lib)))
    files.append(np.load(f))
a, b = files
for k in a.files:
    same = np.array_equal(a[k], b[k])
    print(f"{k:22s} {'identical' if same else 'DIFFERENT  max-abs %.3e' % float(np.abs(a[k].astype(np.float64) - b[k]).max())}")
