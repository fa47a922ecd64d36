import sys, numpy as np
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
from tts_amd import spec
from tts_amd.engine import Engine
pack = spec.synthetic_voice_pack(1)
for (B, n) in [(1, 6), (1, 100), (4, 62), (16, 126), (64, 254), (8, 510)]:
    eng = Engine(spec.synthetic_state_dicts(0))
    rng = np.random.default_rng(B)
    id_lists = [rng.integers(1, 178, size=n).tolist() for _ in range(B)]
    ref = np.stack([pack[n - 1, 0]] * B)
    ids, lens = eng.pack(id_lists)
    with eng.session():
        _, nfr = eng.durations(ids, lens, ref)
        eng.decode(seeds=np.arange(B, dtype=np.uint64))
    _, capT, bT = eng.prof_read("arena_token")
    _, capF, bF = eng.prof_read("arena_frame")
    Tp = (n + 2 + 31) // 32 * 32
    print(f"B {B} T {n+2} Fmax {int(nfr.max())}: token pass {bT/1e6:.1f} MB = {bT/(B*Tp)/1e3:.1f} KB/padded token; frame pass {bF/1e6:.1f} MB = {bF/(B*int(nfr.max()))/1e6:.3f} MB/padded frame", flush=True)
    eng.close()
