"""Generates tests/golden/synthetic_T14.npz from the CPU oracle with seeded synthetic weights.

There are no reference fixtures for this path, so this golden vector pins the
ORACLE against itself across machines/versions (and the engine's token-rate stages against it on
the GPU box); it is not evidence of parity with kokoro_onnx.  Run: python tools/make_golden.py
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from tts_amd import spec  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402

model = R.build(spec.DEFAULT_CONFIG, spec.synthetic_state_dicts(0))
rng = np.random.default_rng(2024)
ids = np.concatenate([[0], rng.integers(1, 178, size=12), [0]]).astype(np.int64)
ref_s = spec.synthetic_voice_pack(1)[11, 0]
pd, _, _ = model.durations(torch.from_numpy(ids)[None], torch.from_numpy(ref_s)[None], 1.0)
noise = torch.from_numpy(R.counter_noise(99, int(pd.sum()) * 600))[None]
taps = {}
audio, pd = model.forward_with_tokens(torch.from_numpy(ids)[None], torch.from_numpy(ref_s)[None], 1.0, noise=noise, taps=taps)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "synthetic_T14.npz"),
                    ids=ids, ref_s=ref_s, noise_seed=np.int64(99), pred_dur=pd.numpy(),
                    duration=taps["duration"].numpy(), d=taps["d"].numpy()[0].astype(np.float32),
                    t_en=taps["t_en"].numpy()[0].astype(np.float32), F0=taps["F0"].numpy()[0].astype(np.float32),
                    audio_len=np.int64(audio.numel()), audio_head=audio.numpy()[:2400].astype(np.float32))
print("pred_dur", pd.tolist(), "audio", audio.numel())
