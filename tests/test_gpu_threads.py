"""GPU tests of the boundary's concurrency contract (SURVEY.md 8b "Threading").

The reference's newest harness calls ``kokoro.create`` from two pool threads on ONE object
(epubviewer_06_latest_1_nov_2025_theme.py:407,692,715) and from a Timer thread (speech7.py:178); a producer
process may share the GPU with its parent (basic.tts_kokoro_2_play_pause_works.py:34-41,136-146).  These tests make
a swap of audio between callers, or a dependence on who else is using the chip, visible as a bit difference.
"""
import threading

import numpy as np
import pytest

import helpers as H
from tts_amd import spec

pytestmark = pytest.mark.gpu

SENTENCES = [
    "Short one.",
    "This sentence is a good deal longer than the first one, so its audio cannot be mistaken for it.",
    "Numbers like 1234 and 5678 are read aloud too!",
    "Why?",
    "The art of war is of vital importance to the state; it is a matter of life and death.",
    "Sound of sirens in the distance, then silence.",
    "A medium sentence with some commas, pauses, and a final stop.",
    "Tiny.",
]


def test_concurrent_create_on_one_object_is_bit_identical():
    """4 threads x several different-length sentences on one Kokoro: every result must equal, bit for bit, the same
    sentence synthesized single-threaded (the noise seed is a hash of the chunk's tokens, so ``create`` is a pure
    function of its arguments).  On the round-1 code thread A could receive thread B's audio."""
    from tts_amd import Kokoro
    k = Kokoro("synthetic", None)
    want = {t: k.create(t, voice="af_synthetic", trim=False)[0].copy() for t in SENTENCES}
    assert len({a.size for a in want.values()}) >= 6                 # lengths differ: a swap cannot go unnoticed
    got, errors = {}, []
    start = threading.Barrier(4)

    def worker(w):
        try:
            start.wait()
            for rep in range(3):
                for i in range(w, len(SENTENCES), 4):
                    t = SENTENCES[(i + rep) % len(SENTENCES)]
                    got[(w, rep, t)] = k.create(t, voice="af_synthetic", trim=False)[0].copy()
        except Exception as e:                                      # pragma: no cover - reported below
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert len(got) == 4 * 3 * 2
    for (w, rep, t), a in got.items():
        assert a.size == want[t].size, (w, rep, t, a.size, want[t].size)
        assert np.array_equal(a, want[t]), (w, rep, t, float(np.abs(a - want[t]).max()))


def test_unpaired_decode_is_refused_not_swapped(engine):
    """Engine level: a decode() whose durations() state was replaced by another thread raises instead of decoding the
    other thread's batch; inside engine.session() the pair is atomic and exact."""
    ids_a, ref_a = H.make_batch(3, [12])
    ids_b, ref_b = H.make_batch(4, [40])
    pa, la = engine.pack(ids_a)
    pb, lb = engine.pack(ids_b)
    engine.durations(pa, la, ref_a)
    t = threading.Thread(target=lambda: engine.durations(pb, lb, ref_b))
    t.start()
    t.join()
    with pytest.raises(RuntimeError):
        engine.decode(seeds=np.array([1], np.uint64))
    with engine.session():
        _, nfr = engine.durations(pa, la, ref_a)
        audio, off = engine.decode(seeds=np.array([1], np.uint64))
    assert audio.size == 600 * int(nfr[0])


def test_lstm_falls_back_when_partners_do_not_arrive(engine, monkeypatch):
    """lstm_split_kernel needs its four workgroups per recurrence co-resident.  With the wait bound shrunk to zero
    polls (K82_LSTM_SPIN_LIMIT=0: every wait that is not already satisfied gives up, which is what a GPU shared with
    another process looks like to the kernel) the engine must notice, re-run the phase on the streaming kernel and
    return the SAME bits as an undisturbed engine -- not an error, not garbage -- and keep working afterwards."""
    from tts_amd.engine import Engine
    id_lists, ref_s = H.make_batch(9, [37, 90, 12, 64, 5])
    ids, lens = engine.pack(id_lists)
    seeds = np.arange(5, dtype=np.uint64) + 50
    with engine.session():
        pred_a, nfr_a = engine.durations(ids, lens, ref_s)
        audio_a, off_a = engine.decode(seeds=seeds)
        audio_a = audio_a.copy()
    assert engine.prof_read("lstm_fallbacks")[1] == 0                # the undisturbed engine never fell back
    monkeypatch.setenv("K82_LSTM_SPIN_LIMIT", "0")
    shaky = Engine(spec.synthetic_state_dicts(0))
    try:
        pred_s, nfr_s = shaky.durations(ids, lens, ref_s)
        audio_s, off_s = shaky.decode(seeds=seeds)
        n_fb = shaky.prof_read("lstm_fallbacks")[1]
        assert n_fb >= 1, "the zero-poll bound did not trigger the fallback: the test exercises nothing"
        assert np.array_equal(pred_a, pred_s) and np.array_equal(off_a, off_s)
        assert np.array_equal(audio_a, audio_s), float(np.abs(audio_a - audio_s).max())
        # the handle stays usable (flag cleared) and stays on the streaming kernel for a while: no new fallback
        pred_2, _ = shaky.durations(ids, lens, ref_s)
        audio_2, _ = shaky.decode(seeds=seeds)
        assert np.array_equal(pred_2, pred_a) and np.array_equal(audio_2, audio_a)
        assert shaky.prof_read("lstm_fallbacks")[1] == n_fb
    finally:
        shaky.close()


def test_two_engines_on_one_gpu_from_two_threads(engine):
    """Two handles decoding at the same time on two streams of one GPU (the parent / producer pattern inside one
    process): whatever the hardware does about co-residency, both must return the bits of a solo run."""
    from tts_amd.engine import Engine
    other = Engine(spec.synthetic_state_dicts(0))
    try:
        batches = [H.make_batch(31, [50, 20, 77]), H.make_batch(32, [33, 120])]
        solo = []
        for (id_lists, ref_s) in batches:
            ids, lens = engine.pack(id_lists)
            with engine.session():
                engine.durations(ids, lens, ref_s)
                a, _ = engine.decode(seeds=np.arange(len(id_lists), dtype=np.uint64))
            solo.append(a.copy())
        out, errors = [None, None], []
        go = threading.Barrier(2)

        def run(k, eng):
            try:
                id_lists, ref_s = batches[k]
                ids, lens = eng.pack(id_lists)
                go.wait()
                for _ in range(3):
                    with eng.session():
                        eng.durations(ids, lens, ref_s)
                        a, _ = eng.decode(seeds=np.arange(len(id_lists), dtype=np.uint64))
                    out[k] = a.copy()
            except Exception as e:                                  # pragma: no cover
                errors.append(repr(e))

        ts = [threading.Thread(target=run, args=(0, engine)), threading.Thread(target=run, args=(1, other))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
        assert np.array_equal(out[0], solo[0]) and np.array_equal(out[1], solo[1])
        print(f"fallbacks while sharing the GPU: {engine.prof_read('lstm_fallbacks')[1]} / {other.prof_read('lstm_fallbacks')[1]}")
    finally:
        other.close()


def test_out_of_range_ids_are_an_error_not_a_fault(engine):
    """Token ids index the embedding tables (ops.hip albert_embed_kernel / embed_kernel): host ids outside
    [0, n_token) -> ValueError (K82_EINVAL); device-resident ids cannot be validated without a copy and are clamped
    inside the kernels -- no out-of-bounds read either way, and the handle keeps working."""
    import torch
    ids = np.zeros((1, 6), dtype=np.int64)
    lens = np.array([6], dtype=np.int32)
    ref = np.zeros((1, 256), np.float32)
    for bad in (178, 10_000, -1):
        ids[0, 2] = bad
        with pytest.raises(ValueError):
            engine.durations(ids, lens, ref)
    ids[0, 2] = 177
    with engine.session():
        p_ok, _ = engine.durations(ids, lens, ref)
    ids_d = torch.tensor([[0, 5, 1 << 40, 7, 9, 0]], dtype=torch.int64, device="cuda")
    with engine.session():
        p_dev, nfr = engine.durations((ids_d.data_ptr(), 6), lens, ref)
        audio, _ = engine.decode(seeds=np.array([0], np.uint64))
    assert nfr[0] >= 6 and np.isfinite(audio).all()
    # clamped to the last row: the same durations as that id given honestly
    ids2 = np.array([[0, 5, 177, 7, 9, 0]], dtype=np.int64)
    p_host, _ = engine.durations(ids2, lens, ref)
    assert np.array_equal(p_dev, p_host)


def test_reserved_workspace_is_not_grown_by_passes_within_the_reservation():
    """k82_reserve (workspace sized up front, no hidden allocations on the hot path): after reserving for
    (B, Tmax, frames) every pass within that size runs in the reserved chunks -- the capacity the handle holds does not
    change -- and gives the bits an unreserved engine gives; a larger pass still works (grows on demand)."""
    from tts_amd.engine import Engine
    sds = spec.synthetic_state_dicts(0)
    id_lists, ref_s = H.make_batch(17, [30, 61, 12, 45])
    seeds = np.arange(4, dtype=np.uint64)
    plain = Engine(sds)
    eng = Engine(sds)
    try:
        ids, lens = plain.pack(id_lists)
        with plain.session():
            _, nfr = plain.durations(ids, lens, ref_s)
            want, _ = plain.decode(seeds=seeds)
        eng.reserve(4, 64, int(nfr.max()) + 50)
        capT, capF = eng.prof_read("arena_token")[1], eng.prof_read("arena_frame")[1]
        assert capT > 0 and capF > 0
        for sub in ([0, 1, 2, 3], [2], [1, 3]):
            i2, l2 = eng.pack([id_lists[k] for k in sub])
            with eng.session():
                eng.durations(i2, l2, ref_s[sub])
                got, off = eng.decode(seeds=seeds[sub])
            assert (eng.prof_read("arena_token")[1], eng.prof_read("arena_frame")[1]) == (capT, capF)
            if sub == [0, 1, 2, 3]:
                assert np.array_equal(got, want)
        big, big_ref = H.make_batch(18, [200] * 6)
        i3, l3 = eng.pack(big)
        with eng.session():
            eng.durations(i3, l3, big_ref)
            a3, _ = eng.decode(seeds=np.arange(6, dtype=np.uint64))
        assert np.isfinite(a3).all() and eng.prof_read("arena_frame")[1] > capF
    finally:
        plain.close()
        eng.close()


def _child_synthesize(texts, q):
    """Producer process (basic.tts_kokoro_2_play_pause_works.py:34-41): builds its OWN Kokoro -- its own HIP context on
    the same GPU -- and synthesizes while the parent does the same."""
    import hashlib
    from tts_amd import Kokoro
    try:
        k = Kokoro("synthetic", None)
        out = []
        for rep in range(6):
            for t in texts:
                a = k.create(t, voice="af_synthetic", trim=False)[0]
                out.append((t, a.size, hashlib.sha256(a.tobytes()).hexdigest()))
        fb = k.engine.prof_read("lstm_fallbacks")[1]
        q.put(("ok", out, fb))
    except Exception as e:                                          # pragma: no cover
        q.put(("error", repr(e), 0))


def test_parent_and_producer_process_share_one_gpu():
    """The reference pre-rolls in the parent while a producer PROCESS synthesizes the rest on the same device
    (basic.tts_kokoro_2_play_pause_works.py:34-41,136-146).  Two processes, two HIP contexts, one GPU: whatever the
    co-residency of the split BiLSTM kernel turns out to be, both must get the samples a solo run gives (the
    streaming fallback computes the same bits)."""
    import hashlib
    import multiprocessing as mp
    from tts_amd import Kokoro
    k = Kokoro("synthetic", None)
    solo = {t: hashlib.sha256(k.create(t, voice="af_synthetic", trim=False)[0].tobytes()).hexdigest() for t in SENTENCES}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    child = ctx.Process(target=_child_synthesize, args=(SENTENCES, q))
    child.start()
    mine = []
    try:
        import queue as _q
        import time
        t_end, res = time.time() + 120, None
        while res is None and time.time() < t_end:                  # keep the GPU busy until the producer is done
            for t in SENTENCES:
                mine.append((t, hashlib.sha256(k.create(t, voice="af_synthetic", trim=False)[0].tobytes()).hexdigest()))
            try:
                res = q.get_nowait()
            except _q.Empty:
                pass
        assert res is not None, "the producer process did not report"
        status, out, fb_child = res
    finally:
        child.join(timeout=60)
        if child.is_alive():                                        # pragma: no cover
            child.kill()
    assert status == "ok", out
    assert all(h == solo[t] for t, h in mine)
    assert len(out) == 6 * len(SENTENCES) and all(h == solo[t] for t, _, h in out)
    assert len(mine) >= len(SENTENCES)
    print(f"BiLSTM fallbacks while two processes shared the GPU: parent {k.engine.prof_read('lstm_fallbacks')[1]}, child {fb_child}")
