"""Rounding noise of the fp32 conv kernel against torch's CPU kernel, per reduction length K = cin * taps (GPU box).
For random x, w: y_engine (k82_op_conv1d, v_mfma_f32_32x32x2_f32), y_torch (F.conv1d fp32 on the CPU), y_64 (float64);
prints rms |y - y_64| / rms |y_64| for both and the ratio.  usage: python tools/conv_noise.py"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402

eng = Engine(spec.synthetic_state_dicts(0))
rng = np.random.default_rng(3)
print(f"{'cin':>5s} {'taps':>4s} {'K':>6s} {'L':>5s} {'eng rms':>10s} {'torch rms':>10s} {'ratio':>6s} {'eng max':>10s} {'torch max':>10s}")
for cin, taps, L in [(16, 1, 512), (64, 1, 512), (128, 1, 512), (256, 1, 512), (768, 1, 512), (2048, 1, 512), (512, 5, 512), (1090, 3, 512),
                     (128, 11, 4096), (256, 7, 4096), (768, 1, 64), (768, 1, 8)]:
    cout = 128
    x = rng.standard_normal((1, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    y, _ = eng.op_conv1d(x, w, b)
    xt, wt, bt = torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b)
    y32 = F.conv1d(xt, wt, bt, padding=(taps - 1) // 2).numpy()
    y64 = F.conv1d(xt.double(), wt.double(), bt.double(), padding=(taps - 1) // 2).numpy()
    sc = np.sqrt((y64 ** 2).mean())
    ee, et = np.sqrt(((y - y64) ** 2).mean()) / sc, np.sqrt(((y32 - y64) ** 2).mean()) / sc
    print(f"{cin:5d} {taps:4d} {cin * taps:6d} {L:5d} {ee:10.3e} {et:10.3e} {ee / et:6.2f} {np.abs(y - y64).max() / sc:10.3e} {np.abs(y32 - y64).max() / sc:10.3e}")
eng.close()
