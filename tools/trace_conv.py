"""Phase timeline of the fp32 conv kernel from a -DK82_TRACE=1 build (GPU box).

  CPU side:  tools/ab_conv.sh build "-DK82_TRACE=1"            (-> tts_amd/libk82_v0.so)
  GPU box:   K82_LIB_PATH=$PWD/tts_amd/libk82_v0.so python tools/trace_conv.py [cin cout L B taps dil act]

Every wave stamps s_memtime at: 1 tile start, 2 ci-tile start, 3 first tap's MFMAs issued, 4 staging written to LDS,
5 last tap issued (before the barrier), 6 epilogue start, 7 epilogue end.  The report gives the mean time a wave spends
between stamps, and, for the two waves that share a SIMD (one of each workgroup on the CU), how much of the kernel
BOTH spend outside the multiply phases -- time in which that SIMD's matrix pipe has nothing to issue.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
cin, cout, L, B, taps, dil, act = (args + [128, 128, 92161, 8, 3, 1, 2][len(args):])[:7]
path = os.environ.pop("K82_TRACE_FILE", "/tmp/k82_trace.bin")
eng = Engine(spec.synthetic_state_dicts(0), precision=os.environ.get("K82_PRECISION", "fp32"))
rng = np.random.default_rng(0)
x = rng.standard_normal((B, cin, L)).astype(np.float32)
w = (rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32)
b = rng.standard_normal(cout).astype(np.float32)
alpha = (1 + 0.5 * rng.random(cin)).astype(np.float32)
_, ms = eng.op_conv1d(x, w, b, dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None, iters=5)   # stamps on, no dump
os.environ["K82_TRACE_FILE"] = path
eng.op_conv1d(x, w, b, dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None, iters=1)
print(f"cin {cin} cout {cout} L {L} B {B} k {taps} d {dil} act {act}: {ms:.3f} ms per launch (trace build)")

CAP = 1024
buf = np.fromfile(path, dtype=np.uint64).reshape(-1, CAP)
MASK = np.uint64((1 << 56) - 1)
waves = []
for row in buf:
    if row[1] == 0:
        continue
    n = int(np.argmax(row[1:] == 0)) + 1 if (row[1:] == 0).any() else CAP
    ids = (row[1:n] >> np.uint64(56)).astype(np.int64)
    ts = (row[1:n] & MASK).astype(np.int64)
    waves.append((int(row[0]), ids, ts))
print(f"{len(waves)} waves traced, {sum(len(w[1]) for w in waves)} stamps")
firsts = np.array([w[2][0] for w in waves]); lasts = np.array([w[2][-1] for w in waves])
t0 = int(np.median(firsts)); t1 = int(np.median(lasts))
tick_us = (t1 - t0) / (ms * 1e3)
print(f"median first stamp -> median last stamp: {t1 - t0} ticks ~ {tick_us:.1f} ticks/us (kernel {ms * 1e3:.0f} us); "
      f"first stamps spread {int(firsts.max() - firsts.min())}, last stamps spread {int(lasts.max() - lasts.min())}")

names = {(3, 8): "ci-tile: wait for the x tile", (8, 4): "ci-tile: transform + LDS write", (1, 2): "tile start -> first ci-tile", (2, 3): "ci-tile: loads + first tap issued", (3, 4): "ci-tile: staging (wait x, transform, LDS write)",
         (4, 5): "ci-tile: remaining taps issued", (2, 5): "ci-tile: all taps (+ threaded staging, ring pipeline)", (5, 2): "ci-tile: barrier (wait for the workgroup)", (5, 6): "last barrier-less gap",
         (6, 7): "epilogue", (7, 1): "tile switch (next tile, bias, zero acc)", (4, 2): "(one-tap) -> next ci-tile"}
acc = {}
for hw, ids, ts in waves:
    d = np.diff(ts)
    for a, bb, dt in zip(ids[:-1], ids[1:], d):
        k = (int(a), int(bb))
        s = acc.setdefault(k, [0, 0])
        s[0] += int(dt)
        s[1] += 1
tot = sum(v[0] for v in acc.values())
print("\nper-wave time between stamps (all waves):")
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {names.get(k, str(k)):52s} {100.0 * s / tot:5.1f} %   mean {s / n / tick_us:7.3f} us  x {n}")

# the two waves of one SIMD: intervals in which a wave is NOT in a multiply phase
def idle_intervals(ids, ts):
    out = []
    for a, s, e in zip(ids[:-1], ts[:-1], ts[1:]):
        if a in (3, 8, 5, 6, 7, 1):       # staging, barrier, epilogue, tile switch
            out.append((s, e))
    return out

by_simd = {}
for hw, ids, ts in waves:
    key = (hw >> 32, (hw >> 8) & 0xff, (hw >> 4) & 3)       # (xcc, se/sh/cu bits, simd)
    by_simd.setdefault(key, []).append((ids, ts))
both = span = single = 0
pairs = 0
for key, ws in by_simd.items():
    if len(ws) != 2:
        continue
    pairs += 1
    s0 = max(ws[0][1][0], ws[1][1][0])
    e0 = min(ws[0][1][-1], ws[1][1][-1])
    if e0 <= s0:
        continue
    span += e0 - s0
    ia, ib = idle_intervals(*ws[0]), idle_intervals(*ws[1])
    single += sum(min(e, e0) - max(s, s0) for s, e in ia if min(e, e0) > max(s, s0))
    single += sum(min(e, e0) - max(s, s0) for s, e in ib if min(e, e0) > max(s, s0))
    i = j = 0
    while i < len(ia) and j < len(ib):
        s = max(ia[i][0], ib[j][0], s0)
        e = min(ia[i][1], ib[j][1], e0)
        if e > s:
            both += e - s
        if ia[i][1] < ib[j][1]:
            i += 1
        else:
            j += 1
print(f"\n{pairs} SIMDs with two traced waves ({len(by_simd)} SIMD keys)")
if span:
    print(f"  a wave is outside the multiply phases {100.0 * single / (2 * span):.1f} % of the time")
    print(f"  BOTH waves of a SIMD are outside them {100.0 * both / span:.1f} % of the time (if independent: {100.0 * (single / (2 * span)) ** 2:.1f} %)")
