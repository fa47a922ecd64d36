"""Sentence-level data parallelism: bucketing, sharding, batching, ordered gather.

The reference synthesizes one sentence per ``create`` call, in any order, and re-assembles by
index (basic.tts_kokoro_2_play_pause_works.py:85-124 keeps a reorder buffer keyed by sentence
number).  Sentences are independent, so the node-level strategy is pure data parallelism: every
GPU gets whole sentences, length-bucketed so padding stays small, and the only exchange is the
final ordered gather of variable-length audio.  No collective touches the forward pass.
"""
from __future__ import annotations

import threading
import time

import numpy as np

from . import spec

BUCKET_EDGES = (32, 64, 128, 256, 512)       # token-count buckets


def estimate_cost(n_tokens, frames_per_token: float = 3.5):
    """Relative FLOP estimate of one sentence (SURVEY.md 8a totals): 79 M MAC per token plus
    attention, 660 M MAC per predicted frame."""
    t = np.asarray(n_tokens, dtype=np.float64)
    return t * (79.0e6 + 18.4e3 * t) + t * frames_per_token * 660.0e6


def bucket_of(n_tokens: int) -> int:
    for e in BUCKET_EDGES:
        if n_tokens <= e:
            return e
    raise ValueError(f"{n_tokens} tokens exceed the 512-token context")


def shard_sentences(token_counts, world: int):
    """Greedy longest-processing-time assignment: returns ``world`` index lists.  Deterministic
    (ties broken by index), every rank computes the same partition without communicating."""
    cost = estimate_cost(token_counts)
    order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(cost[i])
    for lst in out:
        lst.sort()
    return out


def make_batches(indices, token_counts, max_batch: int = 64, max_tokens: int = 64 * 256):
    """Group a rank's sentences into batches of similar length: sort by bucket then length, cut
    when the PADDED token count (batch x longest) or the batch size would exceed the limits."""
    idx = sorted(indices, key=lambda i: (bucket_of(token_counts[i]), token_counts[i], i))
    batches, cur, cur_max = [], [], 0
    for i in idx:
        t = token_counts[i]
        new_max = max(cur_max, t)
        if cur and (len(cur) + 1 > max_batch or new_max * (len(cur) + 1) > max_tokens
                    or bucket_of(t) != bucket_of(token_counts[cur[0]])):
            batches.append(cur)
            cur, new_max = [], t
        cur.append(i)
        cur_max = new_max
    if cur:
        batches.append(cur)
    # heaviest batch first: the engine's workspace arenas are sized by the largest pass seen, so they are
    # grown once instead of once per bucket (each regrowth is a device-wide sync plus tens of GB of hipMalloc)
    batches.sort(key=lambda bt: -sum(estimate_cost(token_counts[i]) for i in bt))
    return batches


class BatchSynthesizer:
    """Runs lists of phoneme-id sentences through one Engine in length-bucketed batches and
    returns audio per sentence, in input order."""

    def __init__(self, engine, max_batch: int = 64, max_tokens: int = 64 * 256, frame_budget: int = 160_000,
                 frame_pad_limit: float = 3.0, output: str = "host", contexts: int = 1):
        """Two-phase execution: batches are formed by token count, phase 1 predicts the
        durations, and a batch whose PREDICTED frame counts are too ragged -- padded frames B * Fmax above
        ``frame_pad_limit`` x the real frames, or more real frames than ``frame_budget`` -- is re-bucketed by F: its
        sentences are ordered by predicted frames and the two halves go through phase 1 + phase 2 on their own (phase 1
        is ~2 % of a sentence's work, and padding is bit-invisible, so the audio does not change).  The frame-rate
        workspace is B * Fmax wide (0.6 MB per padded frame), so this bounds MEMORY; it does not buy time: the conv
        kernels skip the tiles beyond a sentence's own length, and on The Art of War (padded / real frames = 1.5 when
        bucketed by tokens alone) a limit of 1.5 measured 5-9 % SLOWER than no limit (17 more, smaller batches and
        their repeated phase 1) -- hence the loose default, which only catches pathological mixes such as speed 0.5
        next to speed 2.0 (tests/test_gpu_epub.py)."""
        if output not in ("host", "pcm16_device"):
            raise ValueError("output must be 'host' (float32 numpy per sentence) or 'pcm16_device' (int16 torch tensors on the GPU)")
        # "pcm16_device": the bulk form for N GPUs -- a batch's waveform never leaves the GPU as float32: k82_decode writes
        # it into a device buffer, k82_pcm16 converts it there (clip(x, -1, 1) * 32767 -> int16, the reference's playback
        # format: basic.tts_kokoro_2_play_pause_works.py:25), and the per-sentence results are int16 views of that device
        # buffer, ready for ordered_gather_pcm16 to send over RCCL or copy to the host once, at half the bytes.
        # contexts > 1: that many execution contexts on the engine's weights (Engine.fork: own stream, workspace and
        # phase-1 state each), one host thread per context, batches handed out in order.  While one context sits in a
        # batch's token-rate phase (a hundred small launches that cannot fill the chip, then a device -> host sync for
        # the predicted durations) or in host code (packing, page-locked buffers), the other one's frame-rate kernels
        # run: phase 1 of batch i+1 overlaps phase 2 of batch i ("two-phase execution ...
        # per-GPU queues"; the look-ahead intent of speech7.py:218-258).  Batch composition is the same as with one
        # context and a batch gives the same bits on any context, so the audio is bit-identical.
        if contexts < 1:
            raise ValueError("contexts must be >= 1")
        self.output = output
        self.engine = engine
        self.engines = [engine] + [engine.fork() for _ in range(contexts - 1)]
        self._stats_lock = threading.Lock()
        self.max_batch, self.max_tokens, self.frame_budget = max_batch, max_tokens, frame_budget
        self.frame_pad_limit = frame_pad_limit
        self.stats = {"batches": 0, "rebucketed": 0, "frames": 0, "padded_frames": 0,
                      "s_pack": 0.0, "s_durations": 0.0, "s_decode": 0.0}

    def _add(self, **kw):
        with self._stats_lock:
            for k, v in kw.items():
                self.stats[k] += v

    def close(self):
        """Destroys the forked contexts (the engine itself stays the caller's)."""
        for e in self.engines[1:]:
            e.close()
        self.engines = self.engines[:1]

    def _run(self, id_lists, ref_rows, speeds, seeds, out, where, engine=None):
        engine = engine or self.engine
        t0 = time.perf_counter()
        ids, lens = engine.pack(id_lists)
        refs = np.stack(ref_rows)
        self._add(s_pack=time.perf_counter() - t0)
        # the durations -> decode pair shares state inside the handle: one critical section per batch, so two threads
        # on one engine (epubviewer_06_latest_1_nov_2025_theme.py:407,692,715) interleave whole batches, never halves
        with engine.session():
            t0 = time.perf_counter()
            pred, nfr = engine.durations(ids, lens, refs, speed=np.asarray(speeds, np.float32))
            self._add(s_durations=time.perf_counter() - t0)
            real, padded = int(nfr.sum()), int(nfr.max()) * len(id_lists)
            split = len(id_lists) > 1 and (real > self.frame_budget or padded > self.frame_pad_limit * real)
            if not split:
                t0 = time.perf_counter()
                if self.output == "pcm16_device":
                    import torch
                    dev = torch.device("cuda", engine.device)
                    n = real * spec.FRAME_SAMPLES
                    f32 = torch.empty((n,), dtype=torch.float32, device=dev)
                    audio = torch.empty((n,), dtype=torch.int16, device=dev)
                    _, off = engine.decode(seeds=np.asarray(seeds, dtype=np.uint64), out=(f32.data_ptr(), n))
                    engine.pcm16_device(f32.data_ptr(), audio.data_ptr(), n)
                    del f32
                else:
                    audio, off = engine.decode(seeds=np.asarray(seeds, dtype=np.uint64))
                self._add(s_decode=time.perf_counter() - t0, batches=1, frames=real, padded_frames=padded)
        if split:                                                # second-phase re-bucketing by predicted F
            self._add(rebucketed=1)
            order = sorted(range(len(id_lists)), key=lambda k: (int(nfr[k]), k))
            h = len(order) // 2
            for part in (order[:h], order[h:]):
                self._run([id_lists[k] for k in part], [ref_rows[k] for k in part], [speeds[k] for k in part],
                          [seeds[k] for k in part], out, [where[k] for k in part], engine)
            return
        for k, w in enumerate(where):
            out[w] = (audio[off[k]:off[k + 1]], pred[k, :lens[k]].copy())     # views of the batch buffer: no second copy

    def synthesize(self, id_lists, ref_rows, speeds=None, seeds=None, indices=None):
        """-> {index: (audio f32[600 F], pred_dur i32[T])} for ``indices`` (default: all)."""
        n = len(id_lists)
        indices = list(range(n)) if indices is None else list(indices)
        speeds = [1.0] * n if speeds is None else list(speeds)
        seeds = list(range(n)) if seeds is None else list(seeds)
        counts = [len(x) + 2 for x in id_lists]
        out = {}
        batches = make_batches(indices, counts, self.max_batch, self.max_tokens)

        def run(batch, engine):
            self._run([id_lists[i] for i in batch], [ref_rows[i] for i in batch], [speeds[i] for i in batch],
                      [seeds[i] for i in batch], out, batch, engine)

        if len(self.engines) == 1 or len(batches) < 2:
            for batch in batches:
                run(batch, self.engine)
            return out
        # one host thread per context; batches are handed out in order (heaviest first), results land in `out` by index
        nxt, lock, errors = [0], threading.Lock(), []

        def worker(engine):
            try:
                while not errors:
                    with lock:
                        k = nxt[0]
                        nxt[0] += 1
                    if k >= len(batches):
                        return
                    run(batches[k], engine)
            except BaseException as e:                            # re-raised in the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(e,), name=f"k82-context-{k}") for k, e in enumerate(self.engines)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out


def ordered_gather(local: dict, n_total: int, group=None, dst: int = 0):
    """The one exchange of the path: rank ``dst`` receives every rank's {index: payload} and returns
    the payloads as a list in sentence order (other ranks return None).  Host-side gather through
    torch.distributed (gloo or RCCL-backed object gather); audio volume is ~96 kB per audio-second,
    far below any link's capacity."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local[i] for i in range(n_total)]
    rank = dist.get_rank(group)
    bucket = [None] * dist.get_world_size(group) if rank == dst else None
    dist.gather_object(local, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    missing = [i for i in range(n_total) if i not in merged]
    if missing:
        raise RuntimeError(f"ordered_gather: sentences {missing[:5]} were synthesized by no rank")
    return [merged[i] for i in range(n_total)]


def ordered_gather_pcm16(local: dict, n_total: int, group=None, dst: int = 0, device=None):
    """The same exchange as ``ordered_gather`` for bulk audio, as ONE point-to-point tensor transfer per rank instead
    of pickled objects: every rank concatenates its sentences as 16-bit PCM (what the reference plays:
    ``clip(x, -1, 1) * 32767 -> '<i2'``, basic.tts_kokoro_2_play_pause_works.py:25 -- 48 kB per audio-second) in index
    order, the sample counts travel in one small all_gather, and rank ``dst`` receives each rank's block with
    ``recv`` (RCCL over xGMI when ``device`` is a CUDA device and the backend is nccl; gloo on CPU tensors otherwise).
    ``local``: {index: float32 or int16 samples}.  Returns the list of int16 arrays in sentence order on ``dst``,
    None elsewhere."""
    import torch
    import torch.distributed as dist
    from . import wav

    dev = torch.device("cpu") if device is None else torch.device(device)

    def pcm(a):
        """-> int16 torch tensor on ``dev``.  Device-resident int16 (BatchSynthesizer(output="pcm16_device"): converted
        by k82_pcm16 on the GPU) passes through untouched; host float32 is converted on the host with the same
        expression; host int16 is taken as it is."""
        if torch.is_tensor(a):
            if a.dtype != torch.int16:
                raise TypeError("ordered_gather_pcm16: tensors must be int16 (convert on the device with Engine.pcm16_device)")
            return a.to(dev)
        a = np.asarray(a)
        return torch.from_numpy(a if a.dtype == np.int16 else wav.f32_to_s16le_array(a)).to(dev)

    idx = sorted(local)
    blocks = [pcm(local[i]) for i in idx]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if idx != list(range(n_total)):
            raise RuntimeError("ordered_gather_pcm16: a single rank must hold every sentence")
        return [b.cpu().numpy() for b in blocks]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = [None] * world
    dist.all_gather_object(meta, (idx, [int(b.shape[0]) for b in blocks]), group=group)
    mine = torch.cat(blocks) if blocks else torch.zeros((0,), dtype=torch.int16, device=dev)
    if rank != dst:
        if mine.numel():
            dist.send(mine, dst=dst, group=group)
        return None
    out = [None] * n_total
    for r in range(world):
        r_idx, r_len = meta[r]
        total = int(sum(r_len))
        if r == dst:
            buf = mine
        else:
            buf = torch.empty((total,), dtype=torch.int16, device=dev)
            if total:
                dist.recv(buf, src=r, group=group)
        host = buf.cpu().numpy()
        pos = 0
        for i, n in zip(r_idx, r_len):
            if out[i] is not None:
                raise RuntimeError(f"ordered_gather_pcm16: sentence {i} was synthesized by two ranks")
            out[i] = host[pos:pos + n]
            pos += n
    missing = [i for i in range(n_total) if out[i] is None]
    if missing:
        raise RuntimeError(f"ordered_gather_pcm16: sentences {missing[:5]} were synthesized by no rank")
    return out


def audio_seconds(audios) -> float:
    return sum(a.shape[0] for a in audios) / float(spec.SAMPLE_RATE)
