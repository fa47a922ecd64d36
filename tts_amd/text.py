"""Host-side text front end of the synthesis path.

String work stays on host cores (north_star).  Everything here mirrors what the reference's
callers and the pinned ``kokoro_onnx 0.4.9`` wrapper do around ``sess.run`` [EXT where the
code is not under /root/reference]:

* sentence splitters of the reader harnesses (four variants, cited per function);
* phoneme string -> token ids with the config.json vocab, ``[0, *ids, 0]`` framing;
* splitting phoneme strings into chunks of <= 510 at punctuation (``Kokoro._split_phonemes``);
* style-row selection: ``voice[len(tokens)]`` (kokoro_onnx) vs ``pack[len(ps) - 1]`` (KPipeline);
* trailing/leading-silence trim (``kokoro_onnx.trim.trim``, a port of ``librosa.effects.trim``
  with top_db = 60, frame_length = 2048, hop_length = 512).

No phonemizer (misaki / espeak-ng) is installed in this image or on the GPU box:
``phonemize`` uses one if importable and otherwise a deterministic grapheme surrogate, flagged in
its return value -- never parity-relevant, only a way to turn the bundled epubs into phoneme-id
workloads of realistic length.
"""
from __future__ import annotations

import re

import numpy as np

from . import spec

MAX_PHONEME_LENGTH = spec.MAX_PHONEMES

# language strings of kokoro_onnx <-> KPipeline lang codes (kq3_1_f.py:288-298)
LANG_CODES = {"a": "en-us", "b": "en-gb", "j": "ja", "z": "zh", "e": "es", "f": "fr-fr", "h": "hi", "i": "it",
              "p": "pt-br"}
LANG_TO_CODE = {v: k for k, v in LANG_CODES.items()}


# ------------------------------------------------------------------ sentence splitters (a20)
def split_sentences_speech7(text: str):
    """TTSEngine.split_sentences (speech7.py:108-122): split on runs of . ! ? followed by
    whitespace or end of text, keeping the punctuation with its sentence."""
    parts = re.split(r"([.!?]+(?:\s+|$))", text)
    out = []
    for i in range(0, len(parts) - 1, 2):
        s = (parts[i] + parts[i + 1]).strip()
        if s:
            out.append(s)
    if len(parts) % 2 == 1 and parts[-1].strip():
        out.append(parts[-1].strip())
    return out


def split_sentences_basic(text: str):
    """tokenize() of basic.tts_kokoro_2_play_pause_works.py:24."""
    return [p.strip() for p in re.split(r"(?<=[.!?])\s+", text.strip()) if p.strip()]


def split_sentences_newline(text: str):
    """split_sentences of tts_16_3.py:27-29 (also breaks at newlines)."""
    return [p.strip() for p in re.split(r"(?<=[.!?])\s+|\n+", text) if p and p.strip()]


_ABBREV = (r"\b(?:Mr|Mrs|Ms|Dr|Prof|Sr|Jr|St|Mt|vs|etc|Fig|fig|Eq|eq|Dept|No|pp|Rev|Lt|Col|Gen|Sgt|Capt|Sen|Rep|Gov|"
           r"Pres|Ave|Rd|Blvd|Jan|Feb|Mar|Apr|Jun|Jul|Aug|Sep|Sept|Oct|Nov|Dec|Inc|Corp|Ltd|Co|PhD|MD|BA|MA|Bros|viz|"
           r"al|cf|e\.g|i\.e|et|al)\b")


def split_sentences_abbrev(text: str):
    """_split_text_into_sentences (epubviewer_06_latest_1_nov_2025_theme.py:2458-2499):
    abbreviation dots are protected, then split at [.!?] + whitespace + capital/digit/quote."""
    if not text or not text.strip():
        return []
    # :2465-2474 AS SHIPPED: the four "smart quote" replacements there have plain ASCII quotes as their search strings
    # (the typographic characters were lost in the file), so they are no-ops and typographic quotes pass through
    # untouched -- which also means a sentence that opens with a typographic quote is not a split point for the
    # look-ahead below.  What does act: grave and acute accents become apostrophes, a doubled ASCII double quote
    # collapses ('``' can no longer occur after the grave replacement).  Identical results on identical input is the bar.
    for a, b in (("`", "'"), ("´", "'"), ('""', '"')):
        text = text.replace(a, b)
    mark = "∯"
    protected = re.sub(f"({_ABBREV})\\.", lambda m: m.group(0).replace(".", mark), text, flags=re.IGNORECASE)
    parts = re.split(r"(?<=[.!?])\s+(?=[A-Z0-9'\"`(\[])", protected)
    return [p.replace(mark, ".").strip() for p in parts if p.replace(mark, ".").strip()]


def stable_id_for_text(text: str) -> str:
    """Sentence id of the reader harnesses (tts_16_3.py:32-35): the first 12 hex digits of the sha1 of the UTF-8 text --
    the key their highlight spans and audio files are matched by."""
    import hashlib
    return hashlib.sha1(text.encode("utf-8")).hexdigest()[:12]


# ------------------------------------------------------------------ phonemes -> ids (a3)
def tokenize(phonemes: str, vocab=None):
    """Characters missing from the vocab are dropped, as both front ends do [EXT]."""
    vocab = vocab or spec.DEFAULT_VOCAB
    return [vocab[c] for c in phonemes if c in vocab]


def split_phonemes(phonemes: str, max_len: int = MAX_PHONEME_LENGTH):
    """Kokoro._split_phonemes [EXT kokoro_onnx 0.4.9]: pack punctuation-delimited pieces into
    chunks shorter than ``max_len``."""
    pieces = re.split(r"([.,!?;])", phonemes)
    out, cur = [], ""
    for part in pieces:
        part = part.strip()
        if not part:
            continue
        if len(cur) + len(part) + 1 >= max_len:
            if cur.strip():
                out.append(cur.strip())
            while len(part) >= max_len:              # a single piece longer than the context
                out.append(part[:max_len - 1])
                part = part[max_len - 1:]
            cur = part
        elif part in ".,!?;":
            cur += part
        else:
            cur = (cur + " " + part) if cur else part
    if cur.strip():
        out.append(cur.strip())
    return out


def style_row(voice_pack: np.ndarray, n_tokens: int, convention: str) -> np.ndarray:
    """voice_pack f32[510,1,256] -> f32[256].  kokoro_onnx: ``voice[len(tokens)]``;
    KPipeline: ``pack[len(ps) - 1]``."""
    v = np.asarray(voice_pack, dtype=np.float32).reshape(voice_pack.shape[0], -1)
    idx = n_tokens if convention == "kokoro_onnx" else n_tokens - 1
    idx = min(max(idx, 0), v.shape[0] - 1)
    return v[idx]


# ------------------------------------------------------------------ phonemizer bridge (H6)
def phonemize(text: str, lang: str = "en-us"):
    """-> (phoneme string, backend) with backend in {"misaki", "phonemizer", "surrogate"}."""
    code = LANG_TO_CODE.get(lang, lang if lang in LANG_CODES else "a")
    try:                                              # pragma: no cover - absent offline
        from misaki import en
        g2p = en.G2P(trf=False, british=(code == "b"))
        return g2p(text)[0], "misaki"
    except Exception:
        pass
    try:                                              # pragma: no cover - absent offline
        from phonemizer import phonemize as _ph
        return _ph(text, language=LANG_CODES.get(code, lang), backend="espeak", preserve_punctuation=True,
                   with_stress=True).strip(), "phonemizer"
    except Exception:
        pass
    return surrogate_g2p(text), "surrogate"


_SURR = {"c": "k", "q": "k", "x": "ks", "y": "j"}


def surrogate_g2p(text: str) -> str:
    """Deterministic grapheme surrogate (NOT a phonemizer): lower-case letters that exist in the
    vocab map to themselves, a few to near IPA symbols, punctuation and spaces are kept."""
    out = []
    for ch in text.lower():
        if ch in _SURR:
            out.append(_SURR[ch])
        elif ch in spec.DEFAULT_VOCAB:
            out.append(ch)
        elif ch.isspace() or ch in "-—":
            out.append(" ")
    return re.sub(r" +", " ", "".join(out)).strip()


# ------------------------------------------------------------------ trim (a19)
def trim_audio(y: np.ndarray, top_db: float = 60.0, frame_length: int = 2048, hop_length: int = 512):
    """kokoro_onnx.trim.trim [EXT] == librosa.effects.trim: frames whose RMS is within ``top_db`` of
    the loudest frame are non-silent; returns (y[start:end], (start, end))."""
    y = np.asarray(y, dtype=np.float32)
    if y.size == 0:
        return y, (0, 0)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")       # librosa.feature.rms(center=True) zero-pads
    n_frames = 1 + (yp.size - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    rms = np.sqrt(np.mean(yp[idx].astype(np.float64) ** 2, axis=1))
    ref = rms.max()
    if ref <= 0:
        return y[:0], (0, 0)
    db = 10.0 * np.log10(np.maximum(1e-10, rms ** 2)) - 10.0 * np.log10(np.maximum(1e-10, ref ** 2))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[:0], (0, 0)
    start = int(nz[0]) * hop_length
    end = min(y.size, (int(nz[-1]) + 1) * hop_length)
    return y[start:end], (start, end)
