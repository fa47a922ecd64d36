"""Where the wall time of the cfg3 run goes (host packing / phase 1 / phase 2 + D2H / python copies)."""
import gzip, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd import spec, text as T
from tts_amd.engine import Engine
from tts_amd.scheduler import BatchSynthesizer, audio_seconds
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
aow = json.load(gzip.open(os.path.join(ROOT, "tests", "golden", "epub_art_of_war.json.gz")))
sents = [t for _, t in aow["sentences"]]
ids = [x for x in (T.tokenize(T.surrogate_g2p(s), spec.DEFAULT_VOCAB)[:510] for s in sents) if x]
pack = spec.synthetic_voice_pack(1)
rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
eng = Engine(spec.synthetic_state_dicts(0), precision=dtype)
acc = {"durations": 0.0, "decode": 0.0, "n": 0}
d0, c0 = eng.durations, eng.decode
def dur(*a, **k):
    t = time.perf_counter(); r = d0(*a, **k); acc["durations"] += time.perf_counter() - t; return r
def dec(*a, **k):
    t = time.perf_counter(); r = c0(*a, **k); acc["decode"] += time.perf_counter() - t; acc["n"] += 1; return r
eng.durations, eng.decode = dur, dec
syn = BatchSynthesizer(eng, max_batch=256)
syn.synthesize(ids[:64], rows[:64])
acc.update(durations=0.0, decode=0.0, n=0)
eng.prof_enable(True)
t0 = time.perf_counter(); res = syn.synthesize(ids, rows); dt = time.perf_counter() - t0
gpu = sum(eng.prof_read(n)[0] for n in ("conv_mfma_2x2", "conv_mfma_1x4", "lstm", "instnorm", "attention", "istft", "gather", "source", "other"))
print(json.dumps({"dtype": dtype, "wall_s": dt, "durations_s": acc["durations"], "decode_s": acc["decode"], "batches": acc["n"],
                  "python_rest_s": dt - acc["durations"] - acc["decode"], "gpu_kernel_s_in_decode": gpu / 1e3,
                  "audio_s": audio_seconds([res[i][0] for i in range(len(ids))])}))
