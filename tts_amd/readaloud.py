"""Headless read-aloud scheduler: the reference's ``TTSEngine`` without GTK/GStreamer.

Same surface and observable behaviour as ``speech7.py:40-512`` (and the dict-sentence variant of
``epubviewer_06_latest_1_nov_2025_theme.py:758-1105``):

* ``speak_sentences_list(sentences, voice, speed, lang, highlight_callback, finished_callback)``,
  ``next_sentence()``, ``prev_sentence()``, ``pause()``, ``resume()``, ``stop()``, ``split_sentences(text)``,
  ``synthesize_sentence(sentence, voice, speed, lang)``.
* look-ahead window: sentences ``cur .. cur+3`` are synthesized ahead of playback, ``cur .. cur+1`` while
  paused; nothing behind ``cur`` and nothing beyond the window (speech7.py:218-258).
* navigation: the highlight moves at once, playback of the old sentence is cut, and synthesis for the new
  position starts only if the position is still the same 0.5 s later (speech7.py:145-182, 415-450);
  results that arrive for a position the user has left are dropped.
* a sentence that was played to its end is released and the index advances (speech7.py:373-390);
  ``highlight_callback(-1, "")`` is sent at the natural end, not after ``stop()``; ``finished_callback`` always.
* a sentence whose synthesis fails is skipped (speech7.py:336-340).

What is different on purpose (MI355X-first, not a translation):

* the audio stays in memory as float32 arrays (``wav_dir=...`` restores the reference's one-WAV-per-sentence
  files for sinks that want a path); the sink is an interface (``AudioSink``), not a GStreamer playbin;
* the look-ahead window is ONE batched engine call (``Kokoro.create_batch``): on the GPU a batch of four
  sentences costs about what one does, so the window fills in one step instead of four; the sentence the
  listener is waiting for is always synthesized first, alone, to keep first-audio latency at the B=1 figure;
* one condition variable instead of 20-50 ms polling loops, so navigation and ``stop()`` take effect
  immediately; callbacks go through ``dispatch`` (pass ``GLib.idle_add`` in a GTK program; default: call directly).
"""
from __future__ import annotations

import os
import re
import threading
import time

import numpy as np

from . import wav as _wav

LOOKAHEAD = 3               # speech7.py:233
PAUSED_LOOKAHEAD = 1        # speech7.py:231
NAV_DEBOUNCE_S = 0.5        # speech7.py:145, 299, 432, 450


class Playback:
    """Handle returned by ``AudioSink.play``."""

    def done(self) -> bool:
        raise NotImplementedError

    def stop(self) -> None:
        raise NotImplementedError


class AudioSink:
    """Where finished sentences go.  ``play`` must return at once."""

    def play(self, samples: np.ndarray, sample_rate: int, index: int = -1) -> Playback:
        raise NotImplementedError


class _TimedPlayback(Playback):
    def __init__(self, seconds):
        self._end = time.monotonic() + seconds
        self._stopped = False

    def done(self):
        return self._stopped or time.monotonic() >= self._end

    def stop(self):
        self._stopped = True


class NullSink(AudioSink):
    """Plays nothing; a sentence "lasts" ``duration / rate`` seconds (rate=inf: ends at once)."""

    def __init__(self, rate: float = float("inf")):
        self.rate = rate
        self.played = []                     # (index, n_samples) in play order

    def play(self, samples, sample_rate, index=-1):
        self.played.append((index, len(samples)))
        secs = 0.0 if self.rate == float("inf") else len(samples) / float(sample_rate) / self.rate
        return _TimedPlayback(secs)


class PcmChunkSink(NullSink):
    """Feeds ``write(bytes)`` with 2400-frame s16le chunks (basic.tts_kokoro_2_play_pause_works.py:17,25,90)."""

    def __init__(self, write, rate: float = float("inf")):
        super().__init__(rate)
        self._write = write

    def play(self, samples, sample_rate, index=-1):
        for chunk in _wav.iter_chunks(samples):
            self._write(chunk)
        return super().play(samples, sample_rate, index)


class TTSEngine:
    def __init__(self, kokoro=None, sink: AudioSink = None, dispatch=None, *, lookahead: int = LOOKAHEAD,
                 paused_lookahead: int = PAUSED_LOOKAHEAD, debounce: float = NAV_DEBOUNCE_S, keep_behind: int = 0,
                 wav_dir: str = None, poll: float = 0.02, workers: int = 1, batch_window: bool = True,
                 max_wait: float = None):
        """Defaults = speech7.py's policy (window cur .. cur+3, cur+1 while paused, one synthesis thread) with the
        window filled by one batched engine call.  ``TTSEngine.parallel(...)`` gives the newest harness's policy
        (epubviewer_06_latest_1_nov_2025_theme.py:406-412,685-756,967-993): SYNTHESIS_BUFFER_SIZE = 5 sentences
        (cur .. cur+4), MAX_PARALLEL_WORKERS = 2 threads each synthesizing ONE sentence per task, the player waits at
        most 5 s for a sentence and then synthesizes it itself, the last CLEANUP_THRESHOLD = 8 sentences stay cached.
        ``max_wait=None``: wait until the worker delivers or fails (speech7's loop has no time limit while its
        worker is alive, speech7.py:303-315)."""
        self.workers, self.batch_window, self.max_wait = max(1, int(workers)), bool(batch_window), max_wait
        self.kokoro = kokoro
        self.sink = sink or NullSink()
        self._dispatch = dispatch or (lambda fn, *a: fn(*a))
        self.lookahead, self.paused_lookahead = int(lookahead), int(paused_lookahead)
        self.debounce, self.keep_behind, self.wav_dir, self.poll = float(debounce), int(keep_behind), wav_dir, poll
        self.is_playing = False
        self.should_stop = False
        self.paused = False
        self.current_thread = None
        self._cv = threading.Condition()
        self._gen = 0                        # bumps on every speak/stop: stale workers see it and leave
        self._tts_sentences, self._tts_sids, self._as_dict = [], [], False
        self._tts_voice, self._tts_speed, self._tts_lang = None, 1.0, "en-us"
        self._tts_finished_callback = self._tts_highlight_callback = None
        self._current_play_index = 0
        self._audio = {}                     # idx -> float32 samples (or wav path with wav_dir)
        self._failed = set()
        self._inflight = set()
        self._nav_time = None                # monotonic time of the last next/prev
        self._synth_thread = None
        self.synth_log = []                  # [(tuple(indices), seconds)]: one entry per engine call (tests, tuning)
        self.fallbacks = 0                   # sentences the player had to synthesize itself after max_wait
        self._file_seq = 0                   # unique WAV names (the reference's ms timestamp alone can collide)

    @classmethod
    def parallel(cls, kokoro=None, sink: AudioSink = None, dispatch=None, **kw):
        """The look-ahead policy of epubviewer_06_latest_1_nov_2025_theme.py (see ``__init__``)."""
        opts = dict(lookahead=4, paused_lookahead=4, workers=2, batch_window=False, max_wait=5.0, keep_behind=8)
        opts.update(kw)
        return cls(kokoro, sink, dispatch, **opts)

    # ------------------------------------------------------------------ pieces with reference names
    def split_sentences(self, text):
        """speech7.py:108-122: split on [.!?]+ followed by whitespace or end; keep the punctuation."""
        parts = re.split(r"([.!?]+(?:\s+|$))", text)
        out = []
        for i in range(0, len(parts) - 1, 2):
            s = (parts[i] + parts[i + 1]).strip()
            if s:
                out.append(s)
        if len(parts) % 2 == 1 and parts[-1].strip():
            out.append(parts[-1].strip())
        return out

    def _store(self, idx, samples, sr):
        if self.wav_dir:
            os.makedirs(self.wav_dir, exist_ok=True)
            with self._cv:
                self._file_seq += 1
                seq = self._file_seq
            path = os.path.join(self.wav_dir, f"tts_{int(time.time() * 1000)}_{os.getpid()}_{seq}.wav")
            _wav.write_wav(path, samples, sr)
            return path
        return np.asarray(samples, dtype=np.float32)

    def synthesize_sentence(self, sentence, voice, speed, lang):
        """speech7.py:124-133: one sentence -> audio (array, or WAV path with ``wav_dir``); ``None`` on failure."""
        try:
            samples, sr = self.kokoro.create(sentence, voice=voice, speed=speed, lang=lang)
            return self._store(-1, samples, sr)
        except Exception as e:                                   # the reference swallows and reports
            print(f"[error] Synthesis error for sentence: {e}")
            return None

    def _synthesize_many(self, idxs):
        """-> {idx: audio or None}.  One batched call when the engine offers it."""
        texts = [self._tts_sentences[i] for i in idxs]
        t0 = time.monotonic()
        out = {}
        batch = getattr(self.kokoro, "create_batch", None)
        if batch is not None and len(idxs) > 1:
            try:
                res = batch(texts, voice=self._tts_voice, speed=self._tts_speed, lang=self._tts_lang)
                for i, (samples, sr) in zip(idxs, res):
                    out[i] = self._store(i, samples, sr)
            except Exception as e:
                print(f"[error] batched synthesis failed ({e}); falling back to single sentences")
                out = {}
        for i, tx in zip(idxs, texts):
            if i not in out:
                out[i] = self.synthesize_sentence(tx, self._tts_voice, self._tts_speed, self._tts_lang)
        self.synth_log.append((tuple(idxs), time.monotonic() - t0))
        return out

    # ------------------------------------------------------------------ callbacks
    def _highlight(self, idx):
        cb = self._tts_highlight_callback
        if not cb:
            return
        if idx < 0:
            payload = {"sid": None, "text": ""} if self._as_dict else ""
        elif self._as_dict:
            payload = {"sid": self._tts_sids[idx], "text": self._tts_sentences[idx]}
        else:
            payload = self._tts_sentences[idx]
        self._dispatch(cb, idx, payload)

    # ------------------------------------------------------------------ workers
    def _window(self):
        cur = self._current_play_index
        la = self.paused_lookahead if self.paused else self.lookahead
        return cur, min(len(self._tts_sentences) - 1, cur + la)

    def _synthesis_worker(self, gen):
        while True:
            with self._cv:
                while True:
                    if self.should_stop or gen != self._gen or self._current_play_index >= len(self._tts_sentences):
                        return
                    cur, hi = self._window()
                    now = time.monotonic()
                    hold = 0.0
                    if self._nav_time is not None:
                        hold = self._nav_time + self.debounce - now
                        if hold <= 0:
                            self._nav_time, hold = None, 0.0
                    missing = [i for i in range(cur, hi + 1)
                               if i not in self._audio and i not in self._failed and i not in self._inflight]
                    if missing and hold <= 0:
                        # the sentence being waited for goes first and alone; the rest of the window in one batch
                        # (or, under the newest harness's policy, one sentence per task as its pool threads do)
                        todo = [cur] if missing[0] == cur or not self.batch_window else missing
                        if not self.batch_window:
                            todo = [missing[0]]
                        self._inflight |= set(todo)
                        break
                    self._cv.wait(timeout=hold if (missing and hold > 0) else None)
            try:
                res = self._synthesize_many(todo)
            except Exception as e:                               # keep the player alive (speech7.py:261-265): THIS task's
                print(f"[error] Synthesis worker error: {e}")    # sentences count as failed (the player skips them, as it
                res = {i: None for i in todo}                    # does after a failed create), the worker goes on, and
            with self._cv:                                       # the other worker's in-flight entries stay untouched
                self._inflight -= set(todo)
                if gen != self._gen or self.should_stop:
                    self._discard(res.values())
                    self._cv.notify_all()
                    return
                cur, hi = self._window()
                for i, a in res.items():
                    if a is None:
                        self._failed.add(i)
                    elif cur - self.keep_behind <= i <= cur + self.lookahead and i not in self._audio:
                        self._audio[i] = a
                    else:
                        self._discard([a])               # the user has moved on: cancel-on-navigate
                self._cv.notify_all()

    def is_cached(self, idx):
        return idx in self._audio

    def _discard(self, items):
        for a in items:
            if isinstance(a, str):
                try:
                    os.remove(a)
                except OSError:
                    pass

    def _load(self, a):
        if isinstance(a, str):
            return _wav.read_wav(a)
        return a, 24000

    def _tts_thread(self, gen):
        finished_cb = self._tts_finished_callback        # this list's callback, whatever starts after it
        try:
            total = len(self._tts_sentences)
            self._synth_threads = [threading.Thread(target=self._synthesis_worker, args=(gen,), daemon=True)
                                   for _ in range(self.workers)]
            for t in self._synth_threads:
                t.start()
            self._synth_thread = self._synth_threads[0]
            self.is_playing = True
            while True:
                with self._cv:
                    if self.should_stop or gen != self._gen or self._current_play_index >= total:
                        break
                    idx = self._current_play_index
                self._highlight(idx)                              # at once, so navigation feels responsive
                with self._cv:
                    while self.paused and not self.should_stop and gen == self._gen:
                        self._cv.wait(timeout=0.1)
                    # wait for this sentence's audio, a navigation, a failure or stop -- at most max_wait seconds
                    t_wait = time.monotonic()
                    timed_out = False
                    while (not self.should_stop and gen == self._gen and self._current_play_index == idx
                           and idx not in self._audio and idx not in self._failed and not self.paused):
                        if self.max_wait is not None and time.monotonic() - t_wait >= self.max_wait:
                            timed_out = True
                            break
                        self._cv.wait(timeout=0.05 if self.max_wait is not None else 0.1)
                    if self.should_stop or gen != self._gen:
                        break
                    if self._current_play_index != idx or self.paused:
                        continue
                if timed_out:
                    # epubviewer_06...:982-993 "[fallback] Synthesizing idx immediately": the player synthesizes the
                    # sentence itself, synchronously, and keeps what it got
                    print(f"[fallback] Synthesizing {idx} immediately")
                    a = self.synthesize_sentence(self._tts_sentences[idx], self._tts_voice, self._tts_speed, self._tts_lang)
                    self.fallbacks += 1
                    with self._cv:
                        if a is None:
                            self._failed.add(idx)
                        elif idx not in self._audio:
                            self._audio[idx] = a
                        else:
                            self._discard([a])
                with self._cv:
                    if self.should_stop or gen != self._gen:
                        break
                    if self._current_play_index != idx or self.paused:
                        continue
                    if idx in self._failed:
                        print(f"[warn] No audio for index {idx}, skipping forward")
                        self._current_play_index = idx + 1
                        self._cv.notify_all()
                        continue
                    audio = self._audio[idx]
                samples, sr = self._load(audio)
                pb = self.sink.play(samples, sr, idx)
                with self._cv:
                    while (not pb.done() and not self.should_stop and gen == self._gen
                           and self._current_play_index == idx and not self.paused):
                        self._cv.wait(timeout=self.poll)
                    finished = pb.done()
                pb.stop()
                with self._cv:
                    if gen != self._gen or self.should_stop:
                        break
                    if finished and self._current_play_index == idx and not self.paused:
                        for i in [i for i in self._audio if i <= idx - self.keep_behind]:
                            self._discard([self._audio.pop(i)])
                        self._current_play_index = idx + 1
                        self._cv.notify_all()
            natural = not self.should_stop and gen == self._gen
            if gen == self._gen:
                self.is_playing = False
            if natural:
                self._highlight(-1)
            if finished_cb:
                self._dispatch(finished_cb)
        except Exception as e:
            print(f"[error] TTS thread error: {e}")
            if finished_cb:
                self._dispatch(finished_cb)
        finally:
            with self._cv:
                self._cv.notify_all()

    # ------------------------------------------------------------------ public control surface
    def speak_sentences_list(self, sentences, voice="af_sarah", speed=1.0, lang="en-us",
                             highlight_callback=None, finished_callback=None):
        if not self.kokoro:
            print("[warn] TTS not available")
            if finished_callback:
                self._dispatch(finished_callback)
            return
        self.stop()
        with self._cv:
            self._gen += 1
            gen = self._gen
            self.should_stop = False
            self._tts_sentences, self._tts_sids = [], []
            self._as_dict = any(isinstance(s, dict) for s in sentences)
            for s in sentences:
                if isinstance(s, dict):
                    self._tts_sids.append(s.get("sid"))
                    self._tts_sentences.append(s.get("text"))
                else:
                    self._tts_sids.append(None)
                    self._tts_sentences.append(str(s))
            self._tts_voice, self._tts_speed, self._tts_lang = voice, speed, lang
            self._tts_finished_callback, self._tts_highlight_callback = finished_callback, highlight_callback
            self._audio, self._failed, self._inflight = {}, set(), set()
            self._current_play_index = 0
            self._nav_time = None
            self.paused = False
            self.synth_log = []
        self.current_thread = threading.Thread(target=self._tts_thread, args=(gen,), daemon=True)
        self.current_thread.start()

    def _navigate(self, delta):
        if not self._tts_sentences:
            return
        with self._cv:
            n = len(self._tts_sentences)
            self._current_play_index = max(0, min(n - 1, self._current_play_index + delta))
            idx = self._current_play_index
            self._nav_time = time.monotonic()                    # (re)arms the 0.5 s debounce
            self._cv.notify_all()
        self._highlight(idx)

    def next_sentence(self):
        """speech7.py:415-432."""
        self._navigate(+1)

    def prev_sentence(self):
        """speech7.py:434-450."""
        self._navigate(-1)

    def pause(self):
        """speech7.py:452-463: playback stops; look-ahead shrinks to the next sentence."""
        with self._cv:
            self.paused = True
            self._cv.notify_all()

    def resume(self):
        """speech7.py:465-472: the interrupted sentence starts again from its beginning."""
        with self._cv:
            self.paused = False
            self._nav_time = None
            self._cv.notify_all()

    def stop(self):
        """speech7.py:474-512."""
        with self._cv:
            self.should_stop = True
            self.paused = False
            self._nav_time = None
            self._cv.notify_all()
        t = self.current_thread
        if t is not None and t is not threading.current_thread():
            t.join(timeout=1.0)
        with self._cv:
            self.is_playing = False
            self._discard(self._audio.values())
            self._audio.clear()

    # ------------------------------------------------------------------ conveniences for headless use
    def wait(self, timeout=None) -> bool:
        """Block until the current list has finished (or ``timeout``); True when finished."""
        t = self.current_thread
        if t is None:
            return True
        t.join(timeout)
        return not t.is_alive()
