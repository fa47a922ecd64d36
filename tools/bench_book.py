"""cfg3 / cfg5 of SURVEY.md 8d on one GPU (GPU box; reads only committed fixtures).

cfg3: every sentence of The Art of War (tests/golden/epub_art_of_war.json.gz, produced by tts_amd.epub) ->
      surrogate g2p (no phonemizer offline; flagged) -> ids -> length-bucketed batches -> audio-s/s, fp32.
cfg5: sentences one at a time (B=1) through the same front door the read-aloud path uses; p50/p95 of
      (call -> float32 PCM on the host).  Lengths follow Sound-of-Sirens' first chapters (fixture holds the
      character lengths only); the text is the Art of War sentence nearest in length.
Prints one JSON object per config.
"""
import argparse
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec, text as T  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402
from tts_amd.scheduler import BatchSynthesizer, audio_seconds  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-sentences", type=int, default=0)
ap.add_argument("--latency-sentences", type=int, default=200)
ap.add_argument("--max-batch", type=int, default=256)
ap.add_argument("--frame-pad-limit", type=float, default=None, help="BatchSynthesizer.frame_pad_limit (default: the class default)")
ap.add_argument("--skip-latency", action="store_true")
ap.add_argument("--prof", action="store_true", help="per-kernel-family GPU time of the cfg3 pass (HIP events; slows it a little)")
ap.add_argument("--dtype", choices=("fp32", "bf16", "bf16x3"), default="fp32")
ap.add_argument("--contexts", type=int, default=1, help="execution contexts on the one set of weights (k82_fork), a host thread each")
args = ap.parse_args()

G = os.path.join(ROOT, "tests", "golden")
aow = json.load(gzip.open(os.path.join(G, "epub_art_of_war.json.gz")))
sir = json.load(gzip.open(os.path.join(G, "epub_sound_of_sirens.json.gz")))
sents = [t for _, t in aow["sentences"]]
if args.max_sentences:
    sents = sents[:args.max_sentences]
vocab = spec.DEFAULT_VOCAB
ids = [T.tokenize(T.surrogate_g2p(s), vocab)[:T.MAX_PHONEME_LENGTH] for s in sents]
ids = [x for x in ids if x]
pack = spec.synthetic_voice_pack(1)
rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]

eng = Engine(spec.synthetic_state_dicts(0), precision=args.dtype)
syn = BatchSynthesizer(eng, max_batch=args.max_batch, contexts=args.contexts, **({} if args.frame_pad_limit is None else {"frame_pad_limit": args.frame_pad_limit}))
syn.synthesize(ids[:64 * args.contexts], rows[:64 * args.contexts])   # warm-up (arena growth, first launches)
t0 = time.perf_counter()
syn.stats = {k: 0 for k in syn.stats}
eng.host_alloc_s = 0.0
if args.prof:
    eng.prof_enable(True)
res = syn.synthesize(ids, rows)
dt = time.perf_counter() - t0
fam = {}
if args.prof:
    for f in ("conv_mfma_2x2", "conv_mfma_1x4", "lstm", "instnorm", "attention", "istft", "gather", "source", "other"):
        ms, n, _ = eng.prof_read(f)
        fam[f] = {"s": ms / 1e3, "launches": n}
    eng.prof_enable(False)
a = audio_seconds([res[i][0] for i in range(len(ids))])
tok = sum(len(x) + 2 for x in ids)
print(json.dumps({"config": f"cfg3 (1 GPU, {args.dtype}, {args.contexts} context(s)): The Art of War, every sentence", "sentences": len(ids), "tokens": tok,
                  "mean_tokens": tok / len(ids), "audio_s": a, "wall_s": dt, "audio_s_per_s": a / dt,
                  "phonemes": "surrogate g2p (no phonemizer offline)", "weights": "seeded random",
                  "includes": "host packing, H2D ids, D2H audio copies per batch", "gpu_seconds_by_family": fam, "scheduler": dict(syn.stats, frame_pad_limit=syn.frame_pad_limit, host_alloc_s=eng.host_alloc_s)}), flush=True)
syn.close()
if args.skip_latency:
    eng.close()
    sys.exit(0)

# cfg5: one sentence at a time
lens = sir["char_lengths"][:args.latency_sentences]
by_len = sorted(range(len(sents)), key=lambda i: len(sents[i]))
sl = np.array([len(sents[i]) for i in by_len])
pick = [by_len[int(np.searchsorted(sl, L).clip(0, len(by_len) - 1))] for L in lens]
lat, aud = [], []
for k, i in enumerate(pick):
    x = T.tokenize(T.surrogate_g2p(sents[i]), vocab)[:T.MAX_PHONEME_LENGTH] or [16]
    r = T.style_row(pack, len(x), "kokoro_onnx")[None]
    t0 = time.perf_counter()
    audios, _ = eng.synthesize([x], r, speed=1.0, seeds=[k])
    lat.append((time.perf_counter() - t0) * 1e3)
    aud.append(len(audios[0]) / 24000.0)
lat = np.array(lat[5:])                                    # first calls grow the arena
print(json.dumps({"config": f"cfg5 (B=1, {args.dtype}): sentence lengths of Sound-of-Sirens, one at a time", "sentences": len(lat),
                  "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)),
                  "mean_ms": float(lat.mean()), "mean_audio_s": float(np.mean(aud[5:])),
                  "realtime_factor_p50": float(np.mean(aud[5:]) / (np.percentile(lat, 50) / 1e3))}), flush=True)
eng.close()
