"""One conv shape through k82_op_conv1d, repeated: the workload of a rocprofv3 --pmc pass (tools/pmc_conv.sh).
usage: pmc_conv.py <precision> <cin> <cout> <L> <B> <taps> <dil> <act>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402

prec = sys.argv[1]
cin, cout, L, B, taps, dil, act = (int(a) for a in sys.argv[2:9])
eng = Engine(spec.synthetic_state_dicts(0), precision=prec)
rng = np.random.default_rng(0)
x = rng.standard_normal((B, cin, L)).astype(np.float32)
w = (rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32)
alpha = (1 + 0.5 * rng.random(cin)).astype(np.float32)
_, ms = eng.op_conv1d(x, w, None, dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None, iters=8)
print(prec, sys.argv[2:9], f"{ms:.3f} ms {2.0 * cin * cout * taps * L * B / ms / 1e9:.1f} TFLOP/s")
