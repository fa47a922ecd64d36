"""Where is the engine noisier than torch?  (VERDICT r2, weak #1.)  Runs on the GPU box.

For seeded inputs of T tokens the same fp32 parameters are evaluated three ways -- the HIP engine, the fp32 CPU oracle
and the oracle in float64 -- free-running, and for every stage of the prosody branch the table shows
    |engine - fp64|,  |oracle32 - fp64|,  and their ratio
(max-abs over the valid region, relative to max |fp64|).  A ratio that jumps between two consecutive stages names the
kernel that adds more rounding noise than torch's CPU kernel does.
usage: python tools/stage_noise.py [--token] [--seeds N] [T ...]      (default 8 64 128; --token: token-rate stages only)
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402
import helpers as H  # noqa: E402

STAGES_T = ["bert_out", "d_en", "d", "dur_lstm", "duration"]
STAGES_F = ["en", "shared", "f0_blk0", "f0_blk1", "f0_blk2", "F0", "n_blk0", "n_blk1", "n_blk2", "N", "t_en", "asr"]
SEEDS = {8: 2101, 64: 2102, 256: 2103, 512: 2104}


def bct(name, t):
    if name == "duration":
        return t.detach().numpy().reshape(1, -1)
    if name.startswith(("f0_blk", "n_blk")):
        return t.detach().numpy()[0]
    return H.to_bct(name, t)


def main():
    argv = sys.argv[1:]
    token_only = "--token" in argv
    nseeds = int(argv[argv.index("--seeds") + 1]) if "--seeds" in argv else 1
    Ts = [int(a) for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] != "--seeds")] or [8, 64, 128]
    Ts = [T for T in Ts for _ in range(nseeds)]
    seen = {}
    sds = spec.synthetic_state_dicts(0)
    m32 = R.build(spec.DEFAULT_CONFIG, sds)
    m64 = R.build(spec.DEFAULT_CONFIG, sds, torch.float64)
    eng = Engine(sds)
    eng.debug_enable(True)
    for T in Ts:
        seen[T] = seen.get(T, 0) + 1
        rng = np.random.default_rng(SEEDS.get(T, 3000 + T) + 7919 * (seen[T] - 1))
        n = T - 2
        ids = np.concatenate([[0], rng.integers(1, 178, size=n), [0]]).astype(np.int64)
        ref_s = spec.synthetic_voice_pack(1)[n - 1, 0].astype(np.float32)
        x, rs = torch.from_numpy(ids)[None], torch.from_numpy(ref_s)[None]
        pd32, _, _ = m32.durations(x, rs, 1.0)
        F = int(pd32.sum())
        noise = torch.from_numpy(R.counter_noise(7, F * 600))[None]
        t32, t64 = {}, {}
        if token_only:
            m32.durations(x, rs, 1.0, t32)
            m64.durations(x, rs.double(), 1.0, t64)
        else:
            m32.forward_with_tokens(x, rs, 1.0, noise=noise, pred_dur=pd32, taps=t32)
            m64.forward_with_tokens(x, rs.double(), 1.0, noise=noise, pred_dur=pd32, taps=t64)
        lens = np.array([T], np.int32)
        with eng.session():
            pred, nfr = eng.durations(ids[None], lens, ref_s[None])
            got = {k: eng.debug_get(k)[0] for k in STAGES_T}
            if not np.array_equal(pred[0], pd32.numpy()):
                print(f"T{T}: engine durations differ from the fp32 oracle's at {int((pred[0] != pd32.numpy()).sum())} tokens; forcing")
                eng.durations(ids[None], lens, ref_s[None], forced_dur=pd32.numpy()[None].astype(np.int32))
            if not token_only:
                eng.decode(noise=noise[0].numpy())
                got.update({k: eng.debug_get(k)[0] for k in STAGES_F})
        print(f"\nT = {T} tokens, F = {F} frames   (errors relative to max |fp64| of the stage)")
        print(f"{'stage':10s} {'|eng-f64|':>11s} {'|o32-f64|':>11s} {'ratio':>7s}   {'rms eng':>10s} {'rms o32':>10s} {'ratio':>6s}")
        for name in STAGES_T + ([] if token_only else STAGES_F):
            r64 = bct(name, t64[name]).astype(np.float64)
            r32 = bct(name, t32[name]).astype(np.float64)
            g = got[name][:r64.shape[0], :r64.shape[1]].astype(np.float64)
            sc = max(np.abs(r64).max(), 1e-30)
            ee, eo = np.abs(g - r64).max() / sc, np.abs(r32 - r64).max() / sc
            re_, ro = np.sqrt(((g - r64) ** 2).mean()) / sc, np.sqrt(((r32 - r64) ** 2).mean()) / sc
            print(f"{name:10s} {ee:11.3e} {eo:11.3e} {ee / max(eo, 1e-30):7.2f}   {re_:10.3e} {ro:10.3e} {re_ / max(ro, 1e-30):6.2f}")
    eng.close()


if __name__ == "__main__":
    main()
