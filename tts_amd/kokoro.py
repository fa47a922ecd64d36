"""The two Python front doors over the one HIP engine.

A. ``Kokoro(model_path, voices_path).create(text, voice, speed, lang)`` -- what the reference
   really calls (speech7.py:82,127; basic.tts_kokoro.py:17,30; 107 call sites), same names,
   argument meaning, return value ``(float32[N], 24000)`` and error behaviour (exceptions; the
   callers wrap the call in ``try/except Exception``: speech7.py:126-133).
B. ``KPipeline(lang_code, ...)(text, voice, speed) -> (graphemes, phonemes, audio)`` and
   ``KModel(config, model)`` -- the API north_star names.

Both accept the hexgrad ``.pth`` + ``config.json`` + ``voices/*.pt`` layout and the
``voices-v1.0.bin`` npz (weights.py).  ``kokoro-v1.0.onnx`` itself cannot be parsed here (no
onnx library): when ``model_path`` names an ``.onnx`` file the ``.pth`` beside it is used.
With ``model_path="synthetic"`` seeded random weights of the exact architecture are used
(benchmarks / tests; there is no checkpoint offline).
"""
from __future__ import annotations

import os
import re
import threading
import warnings
import zlib
from dataclasses import dataclass

import numpy as np

from . import spec, text as T, weights as W
from .engine import Engine
from .scheduler import BatchSynthesizer

SAMPLE_RATE = spec.SAMPLE_RATE
# model ids that need no checkpoint: seeded random weights of the exact architecture (spec.synthetic_state_dicts).
# "synthetic-calm" is the set on which two fp32 implementations stay sample-comparable free-running (every frame
# unvoiced, duration LSTMs at default scale): the front-door parity tests run on it.
SYNTHETIC_SETS = {"synthetic": {}, "synthetic-calm": {"lstm_gain": 1.0, "f0_gain": 1.0, "f0_bias": 0.0}}
_engine_lock = threading.Lock()
_engines = {}


def _content_seed(ids) -> int:
    """Seed of the harmonic source's noise for one chunk: a hash of its token ids (FNV-1a, 64 bit).  kokoro_onnx
    draws the noise inside the graph (Random* nodes) and exposes no seed; keying ours by content makes ``create``
    a pure function of its arguments -- the same sentence gives the same samples whichever thread, batch or call
    order it is synthesized in (what the concurrent-create test pins)."""
    h = 0xcbf29ce484222325
    for v in ids:
        h = ((h ^ (int(v) & 0xffff)) * 0x100000001b3) & 0xffffffffffffffff
    return h


def _shared_engine(key, state_dicts_fn, config, device, precision="fp32"):
    """One Engine (one copy of the weights) per (checkpoint, device) per process, built lazily in
    the process that first synthesizes (never at import): child processes that construct their
    own Kokoro (basic.tts_kokoro_2_play_pause_works.py:34-41) get their own HIP context."""
    k = (key, device, os.getpid(), precision)      # one engine per precision mode: the mode is handle state, and two
    with _engine_lock:                             # front-door objects must not change each other's arithmetic
        if k not in _engines:
            _engines[k] = Engine(state_dicts_fn(), config, device, precision=precision)
        return _engines[k]


def _load_model(model_path, voices_path, device, precision="fp32"):
    if model_path in SYNTHETIC_SETS:
        cfg = dict(spec.DEFAULT_CONFIG)
        kw = SYNTHETIC_SETS[model_path]
        eng = _shared_engine(model_path, lambda: spec.synthetic_state_dicts(0, **kw), cfg, device, precision)
        voices = {"af_synthetic": spec.synthetic_voice_pack(1)}
        if voices_path and os.path.exists(voices_path):
            voices = W.load_voices(voices_path)
        return eng, cfg, dict(spec.DEFAULT_VOCAB), voices
    pth, cfgp, vo = W.find_model_files(model_path, voices_path)
    if pth is None:
        raise FileNotFoundError(
            f"no Kokoro .pth checkpoint found for model_path={model_path!r} (searched beside it, $KOKORO_ONNX_PATH, "
            "/app/share/kokoro-models, ~/.local/share/kokoro-models); the .onnx file cannot be parsed offline")
    cfg = W.load_config(cfgp) if cfgp else dict(spec.DEFAULT_CONFIG)
    vocab = cfg.get("vocab") or dict(spec.DEFAULT_VOCAB)
    if vo is None:
        raise FileNotFoundError(f"no voices file found (voices_path={voices_path!r})")
    eng = _shared_engine(os.path.abspath(pth), lambda: W.load_pth(pth), cfg, device, precision)
    return eng, cfg, vocab, W.load_voices(vo)


class Kokoro:
    """Drop-in for ``kokoro_onnx.Kokoro`` (pinned 0.4.9, io.github.fastrizwaan.tts.yaml:890-901)."""

    def __init__(self, model_path: str, voices_path: str, device: int = 0, precision: str = "fp32",
                 reserve=(4, 256, 2000)):
        """``reserve``: (sentences, tokens, frames per sentence) the workspace is sized for up front (k82_reserve), so
        the first sentences of a read-aloud session do not pay for workspace growth; None skips it.  The default
        covers the look-ahead window of the readers (4-5 sentences of up to 50 s each) in 5.2 GB of the 288 GB
        (k82_reserve: 0.63 MB per padded frame + 90 KB per padded token).  Best effort: every process that builds a
        Kokoro asks for it (the reference's parent + producer child both do), so on a GPU that cannot spare the
        memory the constructor carries on and the workspace grows on demand instead."""
        self.engine, self.config, self.vocab, self.voices = _load_model(model_path, voices_path, device, precision)
        if reserve:
            try:
                self.engine.reserve(*reserve)
            except (RuntimeError, ValueError) as e:
                warnings.warn(f"tts_amd.Kokoro: workspace reservation {tuple(reserve)} skipped ({e}); it grows on demand")
        self.synth = BatchSynthesizer(self.engine)

    @classmethod
    def from_session(cls, session, voices_path: str):
        """basic2.py:17 builds Kokoro from an onnxruntime session; there is no ORT session to adopt
        here, so this raises clearly instead of silently ignoring the argument."""
        raise NotImplementedError("tts_amd.Kokoro runs on its own HIP engine; construct it with "
                                  "Kokoro(model_path, voices_path) instead of from_session()")

    def get_voices(self):
        return sorted(self.voices.keys())

    def get_voice_style(self, name: str) -> np.ndarray:
        if name not in self.voices:
            raise KeyError(f"voice '{name}' not found; available: {self.get_voices()[:8]}...")
        return self.voices[name]

    def _prepare(self, text, voice, lang, is_phonemes):
        """text -> (token-id lists of its <= 510-phoneme chunks, style rows)."""
        phonemes = text if is_phonemes else T.phonemize(text, lang)[0]
        chunks = T.split_phonemes(phonemes) or [""]
        id_lists = [T.tokenize(ch[:T.MAX_PHONEME_LENGTH], self.vocab)[:T.MAX_PHONEME_LENGTH] for ch in chunks]
        rows = [T.style_row(voice, len(ids), "kokoro_onnx") for ids in id_lists]
        return id_lists, rows

    def create_batch(self, texts, voice, speed: float = 1.0, lang: str = "en-us", is_phonemes: bool = False,
                     trim: bool = True):
        """``[create(t, ...) for t in texts]`` as ONE scheduled engine pass (all chunks of all texts are
        bucketed and batched together).  Not in kokoro_onnx: it is what the read-aloud look-ahead window
        (tts_amd/readaloud.py) and whole-chapter export use, because on the GPU a batch costs about what one
        sentence does.  Same validation, chunking, style-row and trim rules as ``create``."""
        assert 0.5 <= speed <= 2.0, "Speed should be between 0.5 and 2.0"
        if isinstance(voice, str):
            voice = self.get_voice_style(voice)
        voice = np.asarray(voice, dtype=np.float32)
        id_lists, rows, owner = [], [], []
        for ti, text in enumerate(texts):
            ids, rs = self._prepare(text, voice, lang, is_phonemes)
            id_lists += ids
            rows += rs
            owner += [ti] * len(ids)
        seeds = [_content_seed(ids) for ids in id_lists]
        res = self.synth.synthesize(id_lists, rows, speeds=[float(speed)] * len(id_lists), seeds=seeds)
        parts = [[] for _ in texts]
        for i, ti in enumerate(owner):
            a = res[i][0]
            if trim:
                a, _ = T.trim_audio(a)
            parts[ti].append(a)
        return [(np.concatenate(p).astype(np.float32), SAMPLE_RATE) for p in parts]

    def create(self, text, voice, speed: float = 1.0, lang: str = "en-us", is_phonemes: bool = False,
               trim: bool = True):
        """-> (samples float32[N] in [-1, 1], 24000).  Same contract as the reference boundary
        (speech7.py:127): phonemize, chunk to <= 510 phonemes, synthesize each chunk with the
        style row ``voice[len(tokens)]``, optionally trim silence, concatenate."""
        return self.create_batch([text], voice, speed, lang, is_phonemes, trim)[0]


# ------------------------------------------------------------------------------ surface B
@dataclass
class Output:
    audio: "object"
    pred_dur: "object" = None


class KModel:
    """hexgrad ``KModel`` [EXT]: local paths only (no hub download offline)."""

    def __init__(self, repo_id=None, config=None, model=None, device: int = 0):
        self.synthetic = model is None or model in SYNTHETIC_SETS
        if self.synthetic:
            self.engine, self.config, self.vocab, _ = _load_model(model or "synthetic", None, device)
        else:
            cfg = W.load_config(config) if isinstance(config, str) else (config or dict(spec.DEFAULT_CONFIG))
            self.config, self.vocab = cfg, cfg.get("vocab") or dict(spec.DEFAULT_VOCAB)
            self.engine = _shared_engine(os.path.abspath(model), lambda: W.load_pth(model), cfg, device)
        self.context_length = self.config["plbert"]["max_position_embeddings"]

    def forward(self, phonemes: str, ref_s, speed: float = 1, return_output: bool = False):
        import torch
        ids = [self.vocab[p] for p in phonemes if p in self.vocab]
        assert len(ids) + 2 <= self.context_length, (len(ids) + 2, self.context_length)
        rs = np.asarray(ref_s.detach().cpu().numpy() if hasattr(ref_s, "detach") else ref_s, np.float32).reshape(1, 256)
        audios, pred = self.engine.synthesize([ids], rs, speed=float(speed), seeds=[_content_seed(ids)])
        audio = torch.from_numpy(audios[0].copy())
        pd = torch.from_numpy(pred[0, :len(ids) + 2].astype(np.int64))
        return Output(audio=audio, pred_dur=pd) if return_output else audio

    __call__ = forward


@dataclass
class Result:
    graphemes: str
    phonemes: str
    audio: "object" = None
    pred_dur: "object" = None

    def __iter__(self):                       # unpacks to (graphemes, phonemes, audio)
        yield self.graphemes
        yield self.phonemes
        yield self.audio

    def __getitem__(self, i):
        return (self.graphemes, self.phonemes, self.audio)[i]

    def __len__(self):
        return 3


class KPipeline:
    """hexgrad ``KPipeline`` [EXT]: ``for gs, ps, audio in KPipeline('a')(text, voice='af_heart')``."""

    def __init__(self, lang_code: str = "a", repo_id=None, model=True, voices_dir=None, device: int = 0):
        lang_code = T.LANG_TO_CODE.get(lang_code, lang_code)
        assert lang_code in T.LANG_CODES, (lang_code, sorted(T.LANG_CODES))
        self.lang_code = lang_code
        self.voices_dir = voices_dir
        self.voices = {}
        self.synthetic = False
        if model is True:
            pth, cfgp, vo = W.find_model_files(None, voices_dir)
            if pth is None:                    # as Kokoro(...) does: never noise in place of a missing checkpoint
                raise FileNotFoundError(
                    "KPipeline(model=True): no Kokoro .pth checkpoint found ($KOKORO_ONNX_PATH, /app/share/kokoro-models, "
                    "~/.local/share/kokoro-models; there is no hub download offline).  Pass model=KModel(config=..., "
                    "model=<path>) -- or model='synthetic' for seeded random weights (benchmarks / tests only)")
            self.model = KModel(config=cfgp, model=pth, device=device)
            if vo:
                self.voices = W.load_voices(vo)
        elif isinstance(model, str) and model in SYNTHETIC_SETS:
            self.model = KModel(model=model, device=device)
            self.synthetic = True
        elif isinstance(model, KModel):
            self.model = model
            self.synthetic = model.synthetic
        else:
            self.model = None                  # "quiet" pipeline: phonemes only

    def load_voice(self, voice):
        if not isinstance(voice, str):
            return np.asarray(voice.detach().cpu().numpy() if hasattr(voice, "detach") else voice, np.float32)
        if voice in self.voices:
            return self.voices[voice]
        if voice.endswith(".pt") and os.path.exists(voice):
            self.voices[voice] = next(iter(W.load_voices(voice).values()))
            return self.voices[voice]
        if self.voices_dir and os.path.exists(os.path.join(self.voices_dir, voice + ".pt")):
            self.voices[voice] = next(iter(W.load_voices(os.path.join(self.voices_dir, voice + ".pt")).values()))
            return self.voices[voice]
        if self.model is not None and getattr(self, "synthetic", False):
            # synthetic weights only: a voice pack seeded by a STABLE hash of the name (str.__hash__ is salted per
            # process: two ranks of a sharded run would draw different voices)
            self.voices[voice] = spec.synthetic_voice_pack(zlib.crc32(voice.encode("utf-8")) % 1000)
            return self.voices[voice]
        raise KeyError(f"voice '{voice}' not found")

    def __call__(self, text, voice=None, speed: float = 1, split_pattern=r"\n+"):
        import torch
        if isinstance(text, str):
            segs = re.split(split_pattern, text.strip()) if split_pattern else [text]
        else:
            segs = list(text)
        items = []
        for gs in segs:
            if not gs.strip():
                continue
            ps, _ = T.phonemize(gs, T.LANG_CODES[self.lang_code])
            for chunk in (T.split_phonemes(ps) or [""]):
                items.append((gs, chunk))
        if self.model is None:
            for gs, ps in items:
                yield Result(gs, ps, None)
            return
        if voice is None:
            raise ValueError('Specify a voice: pipeline(text="Hello world!", voice="af_heart")')
        pack = self.load_voice(voice)
        id_lists = [T.tokenize(ps, self.model.vocab)[:T.MAX_PHONEME_LENGTH] for _, ps in items]
        rows = [T.style_row(pack, len(ids), "kpipeline") for ids in id_lists]
        synth = BatchSynthesizer(self.model.engine)
        res = synth.synthesize(id_lists, rows, speeds=[float(speed)] * len(items),
                               seeds=[_content_seed(ids) for ids in id_lists])
        for i, (gs, ps) in enumerate(items):
            yield Result(gs, ps, torch.from_numpy(res[i][0]), torch.from_numpy(res[i][1].astype(np.int64)))
