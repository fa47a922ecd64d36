"""GPU tests on the book workloads of BASELINE.json configs[2] and configs[4].

configs[2]: "Kokoro-82M bf16, full The-Art-of-War.epub sentence-bucketed and sharded across 8 x MI355X" -- here the
first 300 sentences of the committed extraction (tests/golden/epub_art_of_war.json.gz, made by tts_amd.epub from the
book the reference ships) through the same BatchSynthesizer / shard_sentences path tools/bench_book_dp.py runs on N
GPUs.  configs[4]: "sentence-at-a-time batch=1 latency for the epub reader play/next path (Sound-of-Sirens.epub)" --
that book's text is not redistributed; the fixture holds its sentence lengths, and the sentences synthesized are the
Art-of-War sentences nearest in length (as tools/bench_book.py does).  No phonemizer exists offline: ids come from
the flagged grapheme surrogate (tts_amd.text.surrogate_g2p), which only shapes the workload.
"""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from tts_amd import spec, text as T
from tts_amd.scheduler import BatchSynthesizer, shard_sentences

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _book(name):
    return json.load(gzip.open(os.path.join(GOLD, name)))


def _ids_of(sentences):
    ids = [T.tokenize(T.surrogate_g2p(s), spec.DEFAULT_VOCAB)[:T.MAX_PHONEME_LENGTH] for s in sentences]
    return [x for x in ids if x]


def _snr(ref, x):
    ref, x = ref.astype(np.float64), x.astype(np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((ref - x) ** 2).sum(), 1e-300))


def test_art_of_war_bf16_sharded(engine, oracle_model):
    sents = [t for _, t in _book("epub_art_of_war.json.gz")["sentences"]][:300]
    ids = _ids_of(sents)
    n = len(ids)
    assert n >= 280
    pack = spec.synthetic_voice_pack(1)
    rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
    seeds = [1000 + i for i in range(n)]
    syn = BatchSynthesizer(engine, max_batch=64)
    with engine.session():
        engine.set_precision("fp32")
        r32 = syn.synthesize(ids, rows, seeds=seeds)
        engine.set_precision("bf16")
        try:
            r16 = syn.synthesize(ids, rows, seeds=seeds)
            # the 2-way shard of tools/bench_book_dp.py: each "rank" synthesizes only its index list (its own batches,
            # its own padding); the union must reproduce the unsharded audio bit for bit
            parts = shard_sentences([len(x) + 2 for x in ids], 2)
            assert sorted(parts[0] + parts[1]) == list(range(n)) and min(len(p) for p in parts) > n // 4
            sharded = {}
            for p in parts:
                sharded.update(syn.synthesize(ids, rows, seeds=seeds, indices=p))
        finally:
            engine.set_precision("fp32")
    assert sorted(r16) == sorted(sharded) == list(range(n))
    snrs = []
    for i in range(n):
        a32, d32 = r32[i]
        a16, d16 = r16[i]
        assert np.array_equal(d32, d16)                       # the duration branch never leaves fp32
        assert a16.size == 600 * int(d16.sum()) and np.isfinite(a16).all()
        assert np.array_equal(a16, sharded[i][0]) and np.array_equal(d16, sharded[i][1]), i
        snrs.append(_snr(a32, a16))
    snrs = np.array(snrs)
    print(f"Art of War, {n} sentences, bf16 vs fp32 mode: SNR min {snrs.min():.1f} / median {np.median(snrs):.1f} dB; "
          f"{sum(a.size for a, _ in r16.values()) / 24000:.0f} audio-s")
    assert snrs.min() >= 40.0, (int(snrs.argmin()), float(snrs.min()))
    # durations against the CPU oracle on a sample of 8 (short, long, and in between)
    order = np.argsort([len(x) for x in ids])
    for i in [int(order[k]) for k in np.linspace(0, n - 1, 8).astype(int)]:
        t = torch.tensor([[0] + ids[i] + [0]], dtype=torch.long)
        pd, _, _ = oracle_model.durations(t, torch.from_numpy(rows[i][None].copy()), 1.0)
        assert np.array_equal(pd.numpy(), r16[i][1]), (i, len(ids[i]))


def test_sound_of_sirens_lengths_one_at_a_time(engine, oracle_model):
    """configs[4]: B = 1, one call per sentence, as the reader's play / next path does (speech7.py:124-133)."""
    sents = [t for _, t in _book("epub_art_of_war.json.gz")["sentences"]]
    lens = _book("epub_sound_of_sirens.json.gz")["char_lengths"]
    lens = [L for L in lens if L >= 8][30:50]                  # 20 sentences past the front matter
    by_len = sorted(range(len(sents)), key=lambda i: len(sents[i]))
    sl = np.array([len(sents[i]) for i in by_len])
    pick = [by_len[int(np.searchsorted(sl, L).clip(0, len(by_len) - 1))] for L in lens]
    pack = spec.synthetic_voice_pack(1)
    total = 0.0
    for k, i in enumerate(pick):
        x = _ids_of([sents[i]])[0]
        row = T.style_row(pack, len(x), "kokoro_onnx")[None]
        audios, pred = engine.synthesize([x], row, speed=1.0, seeds=[k])
        t = torch.tensor([[0] + x + [0]], dtype=torch.long)
        pd, _, _ = oracle_model.durations(t, torch.from_numpy(row.copy()), 1.0)
        assert np.array_equal(pd.numpy(), pred[0, :len(x) + 2]), (k, len(x))
        assert audios[0].size == 600 * int(pd.sum()) and np.isfinite(audios[0]).all()
        total += audios[0].size / 24000
    print(f"Sound-of-Sirens lengths, {len(pick)} sentences at B=1: {total:.0f} audio-s, durations == oracle")


def test_rebucketing_by_predicted_frames_does_not_change_the_audio(engine):
    """scheduler.BatchSynthesizer re-buckets a ragged batch by predicted F (second phase): with slow and fast speech
    in one token bucket the frame counts differ 4x, the batch is split, and every sentence's samples are the bits the
    unsplit batch gives (padding and batch composition are bit-invisible)."""
    sents = [t for _, t in _book("epub_art_of_war.json.gz")["sentences"]][300:340]
    ids = _ids_of(sents)
    pack = spec.synthetic_voice_pack(1)
    rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
    speeds = [0.5 if i % 3 == 0 else 2.0 for i in range(len(ids))]
    seeds = [5000 + i for i in range(len(ids))]
    plain = BatchSynthesizer(engine, frame_pad_limit=float("inf"))
    tight = BatchSynthesizer(engine, frame_pad_limit=1.3)
    a = plain.synthesize(ids, rows, speeds=speeds, seeds=seeds)
    b = tight.synthesize(ids, rows, speeds=speeds, seeds=seeds)
    assert tight.stats["rebucketed"] >= 1 and plain.stats["rebucketed"] == 0
    assert tight.stats["padded_frames"] < plain.stats["padded_frames"]
    for i in range(len(ids)):
        assert np.array_equal(a[i][1], b[i][1]) and np.array_equal(a[i][0], b[i][0]), i
    print(f"padded/real frames: {plain.stats['padded_frames'] / plain.stats['frames']:.2f} -> "
          f"{tight.stats['padded_frames'] / tight.stats['frames']:.2f} ({tight.stats['rebucketed']} batches re-bucketed)")


def test_device_side_pcm16_gather_gives_the_host_conversions_bytes(engine):
    """The bulk N-GPU form of the one exchange step (ordering contract
    basic.tts_kokoro_2_play_pause_works.py:85-124): BatchSynthesizer(output="pcm16_device") converts a batch's waveform
    to 16-bit PCM on the GPU (k82_pcm16, device to device) and ordered_gather_pcm16 takes those device tensors as they
    are.  Single rank here (the exchange itself is covered by the world-size-2 gloo test): the gathered bytes must equal
    the reference expression clip(x, -1, 1) * 32767 -> '<i2' (:25) applied on the host to the float32 path's audio."""
    from tts_amd import wav
    from tts_amd.scheduler import ordered_gather_pcm16
    sents = [t for _, t in _book("epub_art_of_war.json.gz")["sentences"]][100:140]
    ids = _ids_of(sents)
    pack = spec.synthetic_voice_pack(1)
    rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
    seeds = [900 + i for i in range(len(ids))]
    host = BatchSynthesizer(engine).synthesize(ids, rows, seeds=seeds)
    dev = BatchSynthesizer(engine, output="pcm16_device").synthesize(ids, rows, seeds=seeds)
    assert all(torch.is_tensor(dev[i][0]) and dev[i][0].is_cuda and dev[i][0].dtype == torch.int16 for i in dev)
    got = ordered_gather_pcm16({i: dev[i][0] for i in dev}, len(ids), device="cuda")
    want = ordered_gather_pcm16({i: host[i][0] for i in host}, len(ids))
    assert len(got) == len(want) == len(ids)
    for i in range(len(ids)):
        assert np.array_equal(host[i][1], dev[i][1])
        assert got[i].dtype == np.int16 and np.array_equal(got[i], want[i]), i
        assert np.array_equal(want[i], wav.f32_to_s16le_array(host[i][0]))


def test_two_contexts_overlap_phases_and_keep_every_bit(engine):
    """BatchSynthesizer(contexts=2): two execution contexts on one set of weights (k82_fork), a host thread each, batches
    handed out in order -- phase 1 of one batch overlaps phase 2 of another (the reference's
    two pool threads on one Kokoro object, epubviewer_06_latest_1_nov_2025_theme.py:407,692,715).  Every sentence must
    carry the bits of the single-context pass, in fp32 and in the bf16 mode, and both contexts must have worked."""
    sents = [t for _, t in _book("epub_art_of_war.json.gz")["sentences"]][400:560]
    ids = _ids_of(sents)
    pack = spec.synthetic_voice_pack(1)
    rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
    seeds = [7000 + i for i in range(len(ids))]
    one = BatchSynthesizer(engine, max_batch=16)
    two = BatchSynthesizer(engine, max_batch=16, contexts=2)
    try:
        assert len(two.engines) == 2 and two.engines[1] is not engine
        for mode in ("fp32", "bf16"):
            engine.set_precision(mode)                            # (reaches the fork too)
            assert two.engines[1].precision == mode
            try:
                a = one.synthesize(ids, rows, seeds=seeds)
                before = [e.prof_read("arena_frame")[1] for e in two.engines]
                b = two.synthesize(ids, rows, seeds=seeds)
            finally:
                engine.set_precision("fp32")
            assert sorted(a) == sorted(b) == list(range(len(ids)))
            for i in range(len(ids)):
                assert np.array_equal(a[i][1], b[i][1]) and np.array_equal(a[i][0], b[i][0]), (mode, i)
            if mode == "fp32":                                    # the fork started without a workspace: it must have grown one
                assert before[1] == 0 and two.engines[1].prof_read("arena_frame")[1] > 0
        assert two.stats["batches"] == one.stats["batches"] >= 16
        # a parent cannot go while its fork lives
        assert engine.lib.k82_destroy(engine.h) == -1 and b"k82_fork" in engine.lib.k82_last_error(engine.h)
    finally:
        two.close()
    assert len(two.engines) == 1 and not engine._forks


def test_art_of_war_whole_book_bf16_sharded_properties(engine):
    """BASELINE configs[2] at FULL size on one GPU: every sentence of The Art of War (5.8 k sentences, ~14 h of audio) in
    the bf16 mode through BatchSynthesizer, then again as the 2-way shard tools/bench_book_dp.py would run on two GPUs.
    Size-independent properties: every waveform finite and exactly 600 x sum(pred_dur) samples, every duration >= 1,
    and the union of the two shards reproduces the unsharded pass bit for bit (other batches, other padding, other
    launch shapes -- same samples).  The oracle comparison lives in test_art_of_war_bf16_sharded (first 300 sentences)."""
    sents = [t for _, t in _book("epub_art_of_war.json.gz")["sentences"]]
    ids = _ids_of(sents)
    n = len(ids)
    assert n > 5000
    pack = spec.synthetic_voice_pack(1)
    rows = [T.style_row(pack, len(x), "kokoro_onnx") for x in ids]
    seeds = [20000 + i for i in range(n)]
    syn = BatchSynthesizer(engine, max_batch=256)
    engine.set_precision("bf16")
    try:
        whole = syn.synthesize(ids, rows, seeds=seeds)
        parts = shard_sentences([len(x) + 2 for x in ids], 2)
        assert sorted(parts[0] + parts[1]) == list(range(n))
        total = 0
        for p in parts:
            shard = syn.synthesize(ids, rows, seeds=seeds, indices=p)
            assert sorted(shard) == p
            for i in p:
                a, d = whole[i]
                assert d.min() >= 1 and a.size == 600 * int(d.sum()) and np.isfinite(a).all(), i
                assert np.array_equal(d, shard[i][1]) and np.array_equal(a, shard[i][0]), i
                total += a.size
            del shard
    finally:
        engine.set_precision("fp32")
    print(f"The Art of War, {n} sentences, bf16: {total / 24000 / 3600:.1f} h of audio; 2-way shard bit-identical")
