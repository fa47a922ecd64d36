"""First-audio latency, B = 1 (the bench's configs[4] loop) -- prints p50 / p95 / mean of (call -> float32 PCM on the host).
usage: python tools/latency_ab.py [dtype] [sentences]     (environment switches such as K82_SPLITK=0 select the variant)"""
import gzip, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec, text as T
from tts_amd.engine import Engine
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 85
with gzip.open(os.path.join(ROOT, "tests", "golden", "epub_sound_of_sirens.json.gz"), "rt", encoding="utf-8") as fh:
    lens = [min(max(int(v), 1), T.MAX_PHONEME_LENGTH) for v in json.load(fh)["char_lengths"][:n]]
rng = np.random.default_rng(4242)
sent = [rng.integers(1, 178, size=k).tolist() for k in lens]
pack = spec.synthetic_voice_pack(1)
eng = Engine(spec.synthetic_state_dicts(0), precision=dtype)
lat, aud, d_ms = [], [], []
for rep in range(2):
    for k, x in enumerate(sent):
        r = T.style_row(pack, len(x), "kokoro_onnx")[None]
        ids, ln = eng.pack([x])
        t0 = time.perf_counter()
        with eng.session():
            eng.durations(ids, ln, r)
            t1 = time.perf_counter()
            audio, off = eng.decode(seeds=np.array([k], np.uint64))
        t2 = time.perf_counter()
        if rep == 1:
            lat.append((t2 - t0) * 1e3); d_ms.append((t1 - t0) * 1e3); aud.append(audio.size / 24000.0)
lat, d_ms = np.array(lat), np.array(d_ms)
print(json.dumps({"dtype": dtype, "switches": {k: v for k, v in os.environ.items() if k.startswith("K82_")}, "sentences": len(lat),
                  "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)), "mean_ms": float(lat.mean()),
                  "durations_p50_ms": float(np.percentile(d_ms, 50)), "mean_audio_s": float(np.mean(aud)), "mean_tokens": float(np.mean([len(x) + 2 for x in sent])),
                  "splitk_launches": eng.prof_read("conv_shape_splitk")[1]}))
eng.close()
