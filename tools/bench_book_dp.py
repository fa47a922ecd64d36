"""cfg3 of SURVEY.md 8d on N GPUs of one node: The Art of War, sentence-sharded.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      tools/bench_book_dp.py --dtype bf16          (N = 1 works without the launcher)

One process per GPU; every rank computes the same deterministic partition (scheduler.shard_sentences: greedy
longest-processing-time over a FLOP estimate), synthesizes its shard in length-bucketed batches and keeps the audio in
host memory (--gather object) or, by default, as 16-bit PCM in device memory (k82_pcm16 on the GPU right behind the
decoder); the only exchange is the ordered gather of the audio to rank 0 at the end (scheduler.ordered_gather_pcm16:
one int16 tensor per rank over RCCL, device to device), timed separately.  No collective touches the forward pass.  Rank 0 prints one JSON line; wall time is the maximum over
ranks between two barriers.  Reads only committed fixtures.
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

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", choices=("fp32", "bf16", "bf16x3"), default="bf16")      # cfg3 is quoted on the bf16 decoder
ap.add_argument("--max-batch", type=int, default=256)
ap.add_argument("--max-sentences", type=int, default=0)
ap.add_argument("--gather", choices=("pcm16", "object", "none"), default="pcm16",
                help="final ordered gather of the audio to rank 0: one 16-bit PCM tensor per rank over RCCL (default), "
                     "pickled float32 objects, or none")
ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                help="all ranks share GPU 0 and talk over gloo: exercises sharding and the ordered gather on a 1-GPU box")
args = ap.parse_args()

import torch  # noqa: E402
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    if args.rehearse_on_one_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
red_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"

from tts_amd import spec, text as T  # noqa: E402
from tts_amd.engine import Engine  # noqa: E402
from tts_amd.scheduler import (BatchSynthesizer, audio_seconds, ordered_gather, ordered_gather_pcm16,  # noqa: E402
                               shard_sentences)

aow = json.load(gzip.open(os.path.join(ROOT, "tests", "golden", "epub_art_of_war.json.gz")))
sents = [t for _, t in aow["sentences"]]
if args.max_sentences:
    sents = sents[:args.max_sentences]
ids = [x for x in (T.tokenize(T.surrogate_g2p(s), spec.DEFAULT_VOCAB)[:T.MAX_PHONEME_LENGTH] for s in sents) if x]
pack = spec.synthetic_voice_pack(1)
rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
counts = [len(x) + 2 for x in ids]
mine = shard_sentences(counts, world)[rank]

eng = Engine(spec.synthetic_state_dicts(0), device=local, precision=args.dtype)
# --gather pcm16: the waveform is converted to 16-bit PCM ON THE GPU (k82_pcm16) and stays there until the gather
syn = BatchSynthesizer(eng, max_batch=args.max_batch, output="pcm16_device" if args.gather == "pcm16" else "host")
syn.synthesize(ids[:64], rows[:64])                                    # warm-up: arena growth, first launches


def barrier():
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()


barrier()
t0 = time.perf_counter()
res = syn.synthesize(ids, rows, indices=mine)
barrier()
dt = time.perf_counter() - t0
local_audio = {i: res[i][0] for i in mine}
t1 = time.perf_counter()
if args.gather == "pcm16":
    gathered = ordered_gather_pcm16(local_audio, len(ids), device=None if args.rehearse_on_one_gpu else f"cuda:{local}")
elif args.gather == "object":
    gathered = ordered_gather(local_audio, len(ids))
else:
    gathered = None
barrier()
dg = time.perf_counter() - t1
a_local = audio_seconds(list(local_audio.values()))
if dist is not None:
    v = torch.tensor([dt, a_local], dtype=torch.float64, device=red_dev)
    mx = v.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = v.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    dt, a_all = float(mx[0]), float(sm[1])
else:
    a_all = a_local
if rank == 0:
    out = {"config": f"cfg3 ({world} GPU, {args.dtype}): The Art of War, every sentence, sentence-sharded", "n_gpus": world,
           "sentences": len(ids), "tokens": sum(counts), "audio_s": a_all, "wall_s": dt, "audio_s_per_s": a_all / dt,
           "ordered_gather": args.gather, "ordered_gather_s": None if args.gather == "none" else dg,
           "gathered_ok": None if gathered is None else bool(len(gathered) == len(ids) and all(g.size > 0 for g in gathered)),
           "phonemes": "surrogate g2p (no phonemizer offline)", "weights": "seeded random",
           "includes": "host packing, H2D ids, D2H audio copies per batch; the final gather is timed separately"}
    print(json.dumps(out), flush=True)
eng.close()
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
