"""Headless read-aloud scheduler (tts_amd/readaloud.py) against the behaviour of the reference's TTSEngine
(speech7.py:40-512).  A fake Kokoro stands in for the engine: these tests are about scheduling, on CPU."""
import threading
import time

import numpy as np
import pytest

from tts_amd.readaloud import AudioSink, NullSink, PcmChunkSink, Playback, TTSEngine


class FakeKokoro:
    def __init__(self, fail=(), delay=0.0, batch=True):
        self.calls, self.fail, self.delay, self.lock = [], set(fail), delay, threading.Lock()
        if not batch:
            self.create_batch = None

    def create(self, text, voice, speed=1.0, lang="en-us"):
        with self.lock:
            self.calls.append((text,))
        if self.delay:
            time.sleep(self.delay)
        if text in self.fail:
            raise RuntimeError("boom")
        n = 2400 + 10 * len(text)
        return np.full(n, (len(text) % 7) / 10.0, np.float32), 24000

    def create_batch(self, texts, voice, speed=1.0, lang="en-us"):
        with self.lock:
            self.calls.append(tuple(texts))
        if self.delay:
            time.sleep(self.delay)
        if any(t in self.fail for t in texts):
            raise RuntimeError("boom")
        return [(np.full(2400 + 10 * len(t), (len(t) % 7) / 10.0, np.float32), 24000) for t in texts]

    def texts(self):
        return [t for c in self.calls for t in c]


class ManualPlayback(Playback):
    def __init__(self):
        self.ev, self.stopped = threading.Event(), False

    def done(self):
        return self.ev.is_set()

    def stop(self):
        self.stopped = True


class ManualSink(AudioSink):
    """Playback ends only when the test says so."""

    def __init__(self):
        self.started, self.cv = [], threading.Condition()

    def play(self, samples, sr, index=-1):
        pb = ManualPlayback()
        with self.cv:
            self.started.append((index, pb))
            self.cv.notify_all()
        return pb

    def wait_started(self, n, timeout=3.0):
        with self.cv:
            return self.cv.wait_for(lambda: len(self.started) >= n, timeout)


S = [f"Sentence number {i}." for i in range(10)]


def wait_for(pred, timeout=3.0):
    t0 = time.monotonic()
    while time.monotonic() - t0 < timeout:
        if pred():
            return True
        time.sleep(0.005)
    return False


def test_split_sentences_matches_reference_rule():
    e = TTSEngine()
    assert e.split_sentences("Hello there. How are you?  Fine!") == ["Hello there.", "How are you?", "Fine!"]
    assert e.split_sentences("No terminal punctuation") == ["No terminal punctuation"]
    assert e.split_sentences("Wait... what?! ok") == ["Wait...", "what?!", "ok"]
    assert e.split_sentences("") == []


def test_plays_everything_in_order_and_reports():
    k, sink = FakeKokoro(), NullSink()
    hl, fin = [], []
    e = TTSEngine(k, sink, debounce=0.05)
    e.speak_sentences_list(S, highlight_callback=lambda i, t: hl.append((i, t)), finished_callback=lambda: fin.append(1))
    assert e.wait(5.0)
    assert [i for i, _ in sink.played] == list(range(10))
    assert [i for i, _ in hl] == list(range(10)) + [-1] and hl[-1][1] == "" and hl[3][1] == S[3]
    assert fin == [1] and not e.is_playing
    assert sorted(k.texts()) == sorted(S)                      # every sentence synthesized exactly once
    assert k.calls[0] == (S[0],)                               # the awaited sentence goes first, alone
    assert any(len(c) > 1 for c in k.calls)                    # the rest of the window is batched


def test_no_kokoro_reports_finished_immediately():
    fin = []
    TTSEngine(None).speak_sentences_list(S, finished_callback=lambda: fin.append(1))
    assert fin == [1]


def test_lookahead_window_and_pause_limit():
    k, sink = FakeKokoro(), ManualSink()
    e = TTSEngine(k, sink, debounce=0.05)
    e.speak_sentences_list(S)
    assert sink.wait_started(1)
    assert wait_for(lambda: len(k.texts()) == 4)
    time.sleep(0.1)
    assert sorted(k.texts()) == S[:4]                          # cur .. cur+3, nothing further (speech7.py:233)
    e.pause()
    assert wait_for(lambda: sink.started[0][1].stopped)
    sink.started[0][1].ev.set()                                 # the cut playback must not advance the index
    time.sleep(0.1)
    assert e._current_play_index == 0 and len(k.texts()) == 4
    e.resume()                                                  # replays sentence 0 from its start
    assert sink.wait_started(2) and sink.started[1][0] == 0
    sink.started[1][1].ev.set()
    assert sink.wait_started(3) and sink.started[2][0] == 1
    assert wait_for(lambda: S[4] in k.texts())                  # window moved with the index
    e.pause()
    time.sleep(0.1)
    n = len(k.texts())
    assert S[5] not in k.texts()
    e.stop()
    assert len(k.texts()) == n and e._audio == {}


def test_navigation_debounce_and_cancel():
    k, sink = FakeKokoro(), ManualSink()
    hl = []
    e = TTSEngine(k, sink, debounce=0.25)
    e.speak_sentences_list(S, highlight_callback=lambda i, t: hl.append(i))
    assert sink.wait_started(1) and wait_for(lambda: len(k.texts()) == 4)
    for _ in range(6):                                          # 0 -> 6, quickly
        e.next_sentence()
    assert hl[-1] == 6                                          # highlight follows at once
    time.sleep(0.1)
    assert len(k.texts()) == 4                                  # nothing synthesized during the debounce
    assert sink.wait_started(2, timeout=3.0) and sink.started[1][0] == 6
    assert S[4] not in k.texts() and S[5] not in k.texts()      # skipped positions are never synthesized
    assert k.texts()[4] == S[6]
    e.prev_sentence()                                           # back to 5: not cached, comes after the debounce
    t0 = time.monotonic()
    assert sink.wait_started(3, timeout=3.0) and sink.started[2][0] == 5
    assert time.monotonic() - t0 >= 0.2
    e.stop()


def test_stop_midway_calls_finished_but_not_end_highlight():
    k, sink = FakeKokoro(), ManualSink()
    hl, fin = [], []
    e = TTSEngine(k, sink, debounce=0.05)
    e.speak_sentences_list(S, highlight_callback=lambda i, t: hl.append(i), finished_callback=lambda: fin.append(1))
    assert sink.wait_started(1)
    e.stop()
    assert wait_for(lambda: fin == [1])
    assert -1 not in hl and not e.is_playing and sink.started[0][1].stopped


def test_failed_sentence_is_skipped():
    k, sink = FakeKokoro(fail={S[2]}), NullSink()
    e = TTSEngine(k, sink, debounce=0.05)
    e.speak_sentences_list(S[:5])
    assert e.wait(5.0)
    assert [i for i, _ in sink.played] == [0, 1, 3, 4]


def test_dict_sentences_and_engine_without_batch_and_restart():
    k, sink = FakeKokoro(batch=False), NullSink()
    hl = []
    e = TTSEngine(k, sink, debounce=0.05)
    items = [{"sid": 100 + i, "text": t} for i, t in enumerate(S[:3])]
    e.speak_sentences_list(items, highlight_callback=lambda i, p: hl.append((i, p)))
    assert e.wait(5.0)
    assert hl[0] == (0, {"sid": 100, "text": S[0]}) and hl[-1] == (-1, {"sid": None, "text": ""})
    assert all(len(c) == 1 for c in k.calls)
    e.speak_sentences_list(S[5:7])                              # a new list on the same engine
    assert e.wait(5.0)
    assert [i for i, _ in sink.played][-2:] == [0, 1]


def test_pcm_chunk_sink_and_wav_dir(tmp_path):
    got = []
    k, sink = FakeKokoro(), PcmChunkSink(got.append)
    e = TTSEngine(k, sink, debounce=0.05, wav_dir=str(tmp_path))
    e.speak_sentences_list(S[:2])
    assert e.wait(5.0)
    assert all(len(c) <= 2400 * 2 for c in got) and len(got[0]) == 4800
    total = sum(len(c) for c in got) // 2
    # audio went through PCM_16 WAV files: sample counts survive
    assert total == sum(2400 + 10 * len(t) for t in S[:2])
    assert list(tmp_path.iterdir()) == []                       # played files are removed (speech7.py:373-388)


# ------------------------------------------------------------------ the newest harness's policy
def test_parallel_policy_two_workers_five_ahead_one_sentence_per_task():
    """epubviewer_06_latest_1_nov_2025_theme.py:406-412,685-756: SYNTHESIS_BUFFER_SIZE = 5 (cur .. cur+4),
    MAX_PARALLEL_WORKERS = 2, every task is one ``create`` of one sentence; never more than two calls in flight and
    nothing beyond cur+4 is requested while sentence 0 plays."""
    sents = [f"Sentence {i}." for i in range(12)]
    inflight, peak, lock = [0], [0], threading.Lock()

    class Slow(FakeKokoro):
        def create(self, text, voice, speed=1.0, lang="en-us"):
            with lock:
                inflight[0] += 1
                peak[0] = max(peak[0], inflight[0])
            try:
                return super().create(text, voice, speed, lang)
            finally:
                with lock:
                    inflight[0] -= 1

    k = Slow(delay=0.03)
    sink = ManualSink()
    e = TTSEngine.parallel(k, sink)
    assert (e.workers, e.lookahead, e.max_wait, e.keep_behind, e.batch_window) == (2, 4, 5.0, 8, False)
    e.speak_sentences_list(sents, voice="v")
    t0 = time.time()
    while len(k.texts()) < 5 and time.time() - t0 < 5:
        time.sleep(0.01)
    time.sleep(0.15)
    assert sorted(k.texts()) == sorted(sents[:5])                 # the window, and only the window
    assert all(len(c) == 1 for c in k.calls)                      # one sentence per task, never a batch
    assert 1 <= peak[0] <= 2
    e.stop()


def test_player_synthesizes_itself_after_max_wait():
    """:967-993: the player waits at most max_wait (5 s there) for a sentence, then synthesizes it synchronously
    ("[fallback] Synthesizing idx immediately") and plays what it got."""
    sents = ["First one.", "Second one."]

    class Stuck(FakeKokoro):
        def __init__(self):
            super().__init__()
            self.release = threading.Event()
            self.n = 0

        def create(self, text, voice, speed=1.0, lang="en-us"):
            with self.lock:
                self.n += 1
                first = self.n <= 2                               # the two pool workers hang ...
            if first:
                self.release.wait(5.0)
            return super().create(text, voice, speed, lang)       # ... the player's own call goes through

    k = Stuck()
    sink = NullSink()
    e = TTSEngine.parallel(k, sink, max_wait=0.2)
    done = threading.Event()
    e.speak_sentences_list(sents, voice="v", finished_callback=done.set)
    t0 = time.time()
    while not sink.played and time.time() - t0 < 3:
        time.sleep(0.01)
    assert sink.played and sink.played[0][0] == 0 and e.fallbacks >= 1
    k.release.set()
    assert done.wait(5.0)
    assert [i for i, _ in sink.played] == [0, 1]


def test_worker_exception_fails_only_its_own_sentences_and_the_worker_goes_on():
    """A task that raises outside create() (here: _synthesize_many itself) marks ITS sentences failed -- the player skips
    them as after a failed create (speech7.py:336-340) -- leaves the other worker's in-flight entries alone, and both
    workers keep serving the list: no sentence waits for max_wait, none is synthesized twice."""
    k, sink = FakeKokoro(delay=0.01), NullSink()
    e = TTSEngine.parallel(k, sink, debounce=0.01, max_wait=None)
    real = e._synthesize_many
    boom = {3}

    def flaky(idxs):
        if boom & set(idxs):
            raise OSError("disk full")
        return real(idxs)

    e._synthesize_many = flaky
    e.speak_sentences_list(S)
    assert e.wait(5.0)                                         # max_wait=None: would hang if index 3 were never failed
    assert [i for i, _ in sink.played] == [0, 1, 2, 4, 5, 6, 7, 8, 9]
    assert sorted(k.texts()) == sorted(s for i, s in enumerate(S) if i != 3)      # exactly once each
    assert e.fallbacks == 0
