import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tts_amd import spec
from tts_amd.engine import Engine
eng = Engine(spec.synthetic_state_dicts(0))
rng = np.random.default_rng(0)
for cin, cout, L, B, taps, dil in [(128,128,92161,1,3,1),(128,128,92161,2,3,1),(128,128,92161,4,3,1),(128,128,92161,8,3,1),(128,128,92161,1,7,1),(128,128,92161,2,7,1),(128,128,92161,8,7,1),(128,128,23040,8,3,1),(128,128,23040,8,7,1)]:
    x = rng.standard_normal((B, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32)
    fl = 2.0 * cin * cout * taps * L * B
    _, ms = eng.op_conv1d(x, w, None, dil=dil, act=0, iters=20)
    print(f"L {L} B {B} k {taps}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s  bytes {2*B*cin*L*4/1e6:.0f} MB", flush=True)
