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


if __name__ == "__main__":
    sds = spec.synthetic_state_dicts(0)
    m32 = R.build(spec.DEFAULT_CONFIG, sds)
    which = [int(a) for a in sys.argv[1:]] or [14, 8, 64, 256, 512]
    m64 = R.build(spec.DEFAULT_CONFIG, sds, torch.float64) if any(t != 14 for t in which) else None
    for T in which:
        if T == 14:
            legacy_T14(m32)
        else:
            make(T, m32, m64)
