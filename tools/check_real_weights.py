"""Parity suite on REAL Kokoro weights, for whoever has them.  Offline there is no
checkpoint, so the committed tests use seeded random weights of the same architecture ("parity unpinned").
Given the hexgrad checkpoint this runs the SAME GPU parity tests (bit-exact durations, stage-wise and waveform
parity against oracle/kokoro_ref.py, padding invisibility, ...) with those weights:

    python tools/check_real_weights.py /path/to/kokoro-v1_0.pth

The golden-fixture test is deselected (it pins the synthetic weights).  If the `kokoro` package is importable the
script says so: comparing with `kokoro.KModel` itself is then one call on the same ids / ref_s (not automated).
Needs a GPU (the engine has no CPU path).  Exit status = pytest's."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) != 2 or not os.path.exists(sys.argv[1]):
    sys.exit(__doc__)
try:
    import kokoro  # noqa: F401
    print("[info] the kokoro package is importable here: KModel can serve as the true reference")
except Exception:
    print("[info] kokoro package not importable: checking against the in-repo restatement only")
env = dict(os.environ, K82_REAL_PTH=os.path.abspath(sys.argv[1]))
sys.exit(subprocess.call([sys.executable, "-u", "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu",
                          "-x", "-q", "-k", "not golden", "--timeout", "300", "--timeout-method", "thread"], env=env, cwd=ROOT))
