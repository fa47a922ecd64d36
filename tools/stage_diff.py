"""Print per-stage max-abs differences between the HIP engine and the CPU oracle (GPU box)."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402
import helpers as H  # noqa: E402

lengths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,33,20").split(",")]
sds = spec.synthetic_state_dicts(0)
t = time.time()
eng = Engine(sds)
print("engine up in %.1fs" % (time.time() - t), flush=True)
model = R.build(spec.DEFAULT_CONFIG, sds)
id_lists, ref_s = H.make_batch(3, lengths)
eng.debug_enable(True)
ids, lens = eng.pack(id_lists)
pred, nfr = eng.durations(ids, lens, ref_s)
print("nfr", nfr.tolist(), flush=True)
noise = np.concatenate([R.counter_noise(100 + b, int(nfr[b]) * 600) for b in range(len(lengths))])
audio, off = eng.decode(noise=noise)
print("decode done; audio finite:", np.isfinite(audio).all(), flush=True)
raw = eng.debug_get("duration")
for b, x in enumerate(id_lists):
    taps = {}
    nz = noise[off[b] * 9 // 1 // 1: off[b + 1] * 9].reshape(-1, 9) if False else noise.reshape(-1, 9)[off[b]:off[b + 1]]
    T, F = int(lens[b]), int(nfr[b])
    tf = os.environ.get("FREE_RUN") != "1"
    f0g = eng.debug_get("F0")[b, 0, :2 * F] if tf else None
    ng = eng.debug_get("N")[b, 0, :2 * F] if tf else None
    if tf:
        # low-rate phase check (must be bit-exact)
        f0_lo = f0g.astype(np.float32)
        h = np.arange(1, 10, dtype=np.float32)
        fn = (f0_lo[:, None] * h[None, :]).astype(np.float32)
        q = (fn / np.float32(24000.0)).astype(np.float32)
        rad = np.fmod(q, np.float32(1.0)).astype(np.float32)
        rad = np.where(rad < 0, (rad + np.float32(1.0)).astype(np.float32), rad).astype(np.float32)
        cs = np.cumsum(rad.astype(np.float64), axis=0).astype(np.float32)
        ph = ((cs * np.float32(2.0)).astype(np.float32) * R.PI_F).astype(np.float32)
        ph = (ph * np.float32(300)).astype(np.float32)
        got = eng.debug_get("src_ph")[b][:, :2 * F].T
        bad = np.argwhere(got != ph)
        print("src_ph mismatches:", len(bad), "of", ph.size, "first:", bad[:3].tolist())
        if len(bad):
            i, hh = bad[0]
            print("  f0", f0_lo[i], "fn", fn[i, hh], "q", q[i, hh], "rad", rad[i, hh], "cs", cs[i, hh], "ph ref", ph[i, hh], "got", got[i, hh])
        full = R.sinegen_phase(f0_lo)
        print("  oracle full-rate phase max", full.max())
    a_ref, pd_ref = H.oracle_run(model, x, ref_s[b], 1.0, nz, forced=pred[b, :lens[b]], taps=taps, f0=f0g, n=ng)
    pd_free = model.durations(__import__("torch").tensor([[0] + x + [0]]), __import__("torch").from_numpy(ref_s[b][None]), 1.0)[0].numpy()
    print(f"--- sentence {b}: T={T} F={F} dur match={np.array_equal(pd_free, pred[b, :T])} "
          f"raw maxdiff={np.abs(raw[b, 0, :T] - taps['duration'].numpy()).max():.3e}")
    for name, base, mul, add in H.STAGES:
        if name not in taps:
            continue
        ref = H.to_bct(name, taps[name])
        L = (T if base == "T" else F) * mul + add
        got = eng.debug_get(name)[b][: ref.shape[0], :L]
        d = np.abs(got - ref[:, :L])
        print(f"{name:18s} shape {ref.shape} max|ref|={np.abs(ref).max():9.3e} maxdiff={d.max():9.3e} "
              f"at {np.unravel_index(d.argmax(), d.shape)}")
    d = np.abs(audio[off[b]:off[b + 1]] - a_ref)
    snr = 10 * np.log10((a_ref.astype(np.float64) ** 2).sum() / max(((audio[off[b]:off[b + 1]] - a_ref).astype(np.float64) ** 2).sum(), 1e-30))
    print(f"audio SNR {snr:.1f} dB")
    print(f"{'audio':18s} n={a_ref.size} max|ref|={np.abs(a_ref).max():.3f} maxdiff={d.max():.3e} at {d.argmax()}")
