"""Generates tests/golden/synthetic_T{8,14,64,256,512}.npz from the CPU oracle with seeded synthetic weights.

There are no reference fixtures for this path, so these vectors pin the ORACLE against itself
across machines/versions and give the GPU tests committed numbers to meet; they are not evidence of parity with
kokoro_onnx.  Every vector carries TWO evaluations of the same fp32 parameters and inputs:
  *32  the fp32 oracle (what an fp32 implementation is compared with), and
  *64  the same model in float64 (oracle/kokoro_ref.py build(..., dtype=torch.float64)) -- the arbiter where the
       model is ill-conditioned: the tests assert that the engine is as close to the fp64 values as the fp32
       oracle itself is (durations before rounding, F0, N, free-running waveform).
Short vectors (T = 8, 64) also carry the fp32 oracle's F0 / N / har tensors and full waveform, which the engine is
teacher-forced with (k82_debug_force) and compared to sample by sample; long ones (T = 256, 512) carry every 8th
sample.  T = 64 and T = 512 decode with forced durations of 2 frames per token to keep the files small; the
model's own durations are stored and checked either way.
Run: python tools/make_golden.py [T ...]        (a few minutes of CPU for the long ones)
"""
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from tts_amd import spec  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CHECKSUM_STAGES = ("bert_out", "d_en", "d", "dur_lstm", "en", "shared", "t_en", "asr", "dec_encode", "dec_out",
                   "gen0_noise_conv", "gen0_xsource", "gen0_out", "gen1_xsource", "gen1_out", "conv_post")
# T (tokens incl. pads) -> (rng seed, noise seed, forced frames per token for the decode or None, audio stride)
CASES = {8: (2101, 7, None, 1), 64: (2102, 8, 2, 1), 256: (2103, 9, None, 8), 512: (2104, 10, 2, 8)}


def checksums(taps):
    """stage -> (sum, sum |x|, max |x|) in float64 of the fp32 oracle's tensor."""
    out = {}
    for k in CHECKSUM_STAGES:
        a = taps[k].detach().numpy().astype(np.float64)
        out[k] = np.array([a.sum(), np.abs(a).sum(), np.abs(a).max()])
    return out


def legacy_T14(model):
    rng = np.random.default_rng(2024)
    ids = np.concatenate([[0], rng.integers(1, 178, size=12), [0]]).astype(np.int64)
    ref_s = spec.synthetic_voice_pack(1)[11, 0]
    pd, _, _ = model.durations(torch.from_numpy(ids)[None], torch.from_numpy(ref_s)[None], 1.0)
    noise = torch.from_numpy(R.counter_noise(99, int(pd.sum()) * 600))[None]
    taps = {}
    audio, pd = model.forward_with_tokens(torch.from_numpy(ids)[None], torch.from_numpy(ref_s)[None], 1.0, noise=noise, taps=taps)
    np.savez_compressed(os.path.join(GOLD, "synthetic_T14.npz"),
                        ids=ids, ref_s=ref_s, noise_seed=np.int64(99), pred_dur=pd.numpy(),
                        duration=taps["duration"].numpy(), d=taps["d"].numpy()[0].astype(np.float32),
                        t_en=taps["t_en"].numpy()[0].astype(np.float32), F0=taps["F0"].numpy()[0].astype(np.float32),
                        audio_len=np.int64(audio.numel()), audio_head=audio.numpy()[:2400].astype(np.float32))
    print("T14: pred_dur", pd.tolist(), "audio", audio.numel())


def make(T, m32, m64):
    seed, nseed, fpt, stride = CASES[T]
    rng = np.random.default_rng(seed)
    n = T - 2
    ids = np.concatenate([[0], rng.integers(1, 178, size=n), [0]]).astype(np.int64)
    ref_s = spec.synthetic_voice_pack(1)[n - 1, 0].astype(np.float32)             # KPipeline row: pack[len(ps) - 1]
    x, rs = torch.from_numpy(ids)[None], torch.from_numpy(ref_s)[None]
    t0 = time.time()
    pd32, _, raw32 = m32.durations(x, rs, 1.0)
    pd64, _, raw64 = m64.durations(x, rs.double(), 1.0)
    dec = pd32.clone() if fpt is None else torch.full_like(pd32, fpt)
    F = int(dec.sum())
    noise = torch.from_numpy(R.counter_noise(nseed, F * 600))[None]
    t32, t64 = {}, {}
    a32, _ = m32.forward_with_tokens(x, rs, 1.0, noise=noise, pred_dur=dec, taps=t32)
    a64, _ = m64.forward_with_tokens(x, rs.double(), 1.0, noise=noise, pred_dur=dec, taps=t64)
    out = dict(ids=ids, ref_s=ref_s, noise_seed=np.int64(nseed), pred_dur=pd32.numpy().astype(np.int32),
               pred_dur64=pd64.numpy().astype(np.int32), raw32=raw32.numpy().astype(np.float32),
               raw64=raw64.numpy().astype(np.float64), decode_dur=dec.numpy().astype(np.int32),
               F0_32=t32["F0"].numpy()[0].astype(np.float32), N_32=t32["N"].numpy()[0].astype(np.float32),
               F0_64=t64["F0"].numpy()[0].astype(np.float64), N_64=t64["N"].numpy()[0].astype(np.float64),
               audio_stride=np.int64(stride), audio_len=np.int64(a32.numel()),
               audio32=a32.numpy()[::stride].astype(np.float32), audio64=a64.numpy()[::stride].astype(np.float32))
    for k, v in checksums(t32).items():
        out["sum_" + k] = v
    if stride == 1:                                                 # teacher-forcing tensors for the short vectors
        out["har32"] = t32["har"].numpy()[0].astype(np.float32)    # [22, 120F+1]: the oracle's own STFT of its source
    np.savez_compressed(os.path.join(GOLD, f"synthetic_T{T}.npz"), **out)
    e = (a32.double() - a64).abs().max().item()
    print(f"T{T}: F={F} frames, pred_dur32==pred_dur64: {bool((pd32 == pd64).all())}, "
          f"max|raw32-raw64|={float((raw32.double() - raw64).abs().max()):.2e}, "
          f"min margin64={float(((raw64 % 1.0) - 0.5).abs().min()):.2e}, "
          f"free-running max|audio32-audio64|={e:.3f}, {time.time() - t0:.0f} s, "
          f"{os.path.getsize(os.path.join(GOLD, f'synthetic_T{T}.npz')) / 1e6:.2f} MB")


STAGE_NOISE = ("bert_out", "d_en", "d", "dur_lstm", "t_en", "shared", "f0_blk0", "f0_blk1", "f0_blk2", "n_blk0", "n_blk1", "n_blk2")
STAGE_KEEP = 2048            # elements kept per stage (a fixed stride over the flattened [C, L] tensor)


def stage_index(n):
    return np.arange(n)[::max(1, n // STAGE_KEEP)][:STAGE_KEEP]


def make_stages(T, m32, m64):
    """tests/golden/synthetic_T{T}_stages.npz: strided samples of the token-rate and prosody-branch stage tensors, from
    the fp32 oracle AND its float64 evaluation, on the inputs / decode durations of synthetic_T{T}.npz.  The GPU test
    compares |engine - fp64| with |oracle32 - fp64| stage by stage (tests/test_gpu_golden.py): an engine kernel that
    rounds more than torch's CPU kernel shows as a ratio that jumps at its stage."""
    g = np.load(os.path.join(GOLD, f"synthetic_T{T}.npz"))
    x, rs = torch.from_numpy(g["ids"])[None], torch.from_numpy(g["ref_s"])[None]
    dec = torch.from_numpy(g["decode_dur"].astype(np.int64))
    idx = torch.repeat_interleave(torch.arange(T), dec)
    out = {}
    for tag, m, r in (("32", m32, rs), ("64", m64, rs.double())):
        taps = {}
        _, d, _ = m.durations(x, r, 1.0, taps)
        en = d.transpose(-1, -2)[:, :, idx]                       # == d^T @ one-hot alignment
        with torch.no_grad():
            _, _, taps["shared"] = m.predictor.F0Ntrain(en, r[:, 128:], taps)
            taps["t_en"] = m.text_encoder(x)
        for k in STAGE_NOISE:
            a = taps[k].detach().numpy()
            a = a[0].T if k in ("bert_out", "d", "dur_lstm") else a[0]            # -> [C, L]
            flat = np.ascontiguousarray(a).reshape(-1)
            out[f"{k}_{tag}"] = flat[stage_index(flat.size)].astype(np.float32 if tag == "32" else np.float64)
            out[f"{k}_shape"] = np.array(a.shape, np.int64)
            if tag == "64":
                out[f"{k}_absmax"] = np.float64(np.abs(a).max())
    path = os.path.join(GOLD, f"synthetic_T{T}_stages.npz")
    np.savez_compressed(path, **out)
    print(f"T{T} stages: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    sds = spec.synthetic_state_dicts(0)
    m32 = R.build(spec.DEFAULT_CONFIG, sds)
    stages_only = "--stages" in sys.argv[1:]                      # only the *_stages.npz files (needs the main vectors)
    which = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [14, 8, 64, 256, 512]
    m64 = R.build(spec.DEFAULT_CONFIG, sds, torch.float64) if any(t != 14 for t in which) else None
    for T in which:
        if T == 14:
            if not stages_only:
                legacy_T14(m32)
        else:
            if not stages_only:
                make(T, m32, m64)
            make_stages(T, m32, m64)
