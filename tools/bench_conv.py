"""Micro-benchmark of the MFMA conv kernel through k82_op_conv1d (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402

eng = Engine(spec.synthetic_state_dicts(0), precision=os.environ.get("K82_PRECISION", "fp32"))
print("precision", eng.precision)
rng = np.random.default_rng(0)
shapes = [  # (cin, cout, L, B, taps, dil)
    (128, 128, 92161, 8, 3, 1), (128, 128, 92161, 8, 7, 1), (128, 128, 92161, 8, 7, 3), (128, 128, 92161, 8, 11, 5),
    (256, 256, 15360, 16, 7, 1), (256, 256, 15360, 16, 11, 5), (1090, 1024, 768, 32, 3, 1), (768, 2304, 256, 64, 1, 1),
]
if len(sys.argv) > 1 and sys.argv[1] == "post":      # conv_post (128 -> 22, k = 7) and the F0 / N heads (256 -> 1, k = 1)
    shapes = [(128, 22, 92161, 8, 7, 1), (256, 1, 30720, 16, 1, 1)]
elif len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for cin, cout, L, B, taps, dil in shapes:
    x = rng.standard_normal((B, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    alpha = (1 + 0.5 * rng.random(cin)).astype(np.float32)
    fl = 2.0 * cin * cout * taps * L * B
    for act, name in ((0, "none"), (1, "leaky"), (2, "snake")):
        if act == 2 and cout <= 32:              # no Snake conv has <= 32 output channels (not built)
            continue
        _, ms = eng.op_conv1d(x, w, b, dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None, iters=10)
        print(f"cin {cin:4d} cout {cout:4d} L {L:6d} B {B:2d} k {taps:2d} d {dil} act {name:5s}: {ms:8.3f} ms "
              f"{fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
