"""Output formats on the host side of the boundary.

Three things the reference does with the float32 samples ``Kokoro.create`` returns:

* ``sf.write(path, samples, 24000)`` (speech7.py:129; basic.tts_kokoro.py:33): soundfile's default subtype
  for a ``.wav`` name is PCM_16; libsndfile scales by 0x7FFF, rounds to nearest-even (``lrintf``) and, because
  python-soundfile switches clipping on, saturates.  ``write_wav`` reproduces that file byte for byte
  (44-byte canonical RIFF header, little-endian int16 payload).  *parity unpinned*: soundfile is not installed
  here and the reference holds no .wav fixture, so the header/rounding follow libsndfile's published behaviour.
* ``f32_to_s16le``: ``(clip(x,-1,1) * 32767).astype('<i2').tobytes()`` -- truncation toward zero
  (basic.tts_kokoro_2_play_pause_works.py:25).  On the GPU this is ``k82_pcm16``; ``f32_to_s16le`` here is
  the host form used when the samples are already on the host.
* fixed 2400-frame chunks fed to the player (basic.tts_kokoro_2_play_pause_works.py:17,90): ``iter_chunks``.
"""
from __future__ import annotations

import io
import struct

import numpy as np

CHUNK_FRAMES = 2400          # basic.tts_kokoro_2_play_pause_works.py:17


def f32_to_s16le(x) -> bytes:
    """basic.tts_kokoro_2_play_pause_works.py:25 (truncating conversion)."""
    x = np.asarray(x, dtype=np.float32)
    return (np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2").tobytes()


def f32_to_s16le_array(x) -> np.ndarray:
    """The same conversion as ``f32_to_s16le``, as an int16 array."""
    x = np.asarray(x, dtype=np.float32)
    return (np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2")


def float_to_pcm16_sndfile(x) -> np.ndarray:
    """libsndfile float -> PCM_16 with normalisation and clipping on: lrint(x * 32767) saturated."""
    x = np.asarray(x)
    if x.dtype == np.float64:
        s = np.rint(x * 32767.0)
    else:
        s = np.rint(x.astype(np.float32) * np.float32(32767.0))
    return np.clip(s, -32768, 32767).astype("<i2")


def wav_bytes(samples, sample_rate: int = 24000) -> bytes:
    """One RIFF/WAVE PCM_16 file image; ``samples`` float [N] or [N, channels] (or already int16)."""
    a = np.asarray(samples)
    ch = 1 if a.ndim == 1 else a.shape[1]
    pcm = a.astype("<i2") if a.dtype == np.int16 else float_to_pcm16_sndfile(a)
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE"
    hdr += b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, int(sample_rate), int(sample_rate) * ch * 2, ch * 2, 16)
    hdr += b"data" + struct.pack("<I", len(data))
    return hdr + data


def write_wav(path, samples, sample_rate: int = 24000):
    """Stand-in for ``sf.write(path, samples, sample_rate)`` with its defaults for a .wav name."""
    b = wav_bytes(samples, sample_rate)
    if isinstance(path, (str, bytes)) or hasattr(path, "__fspath__"):
        with open(path, "wb") as f:
            f.write(b)
    else:
        path.write(b)


def read_wav(path):
    """-> (float32 samples in [-1, 1), sample_rate); PCM_16 only (what write_wav produces)."""
    raw = open(path, "rb").read() if not isinstance(path, (bytes, bytearray)) else bytes(path)
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE", "not a RIFF/WAVE file"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, sz = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        body = raw[pos + 8:pos + 8 + sz]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
        pos += 8 + sz + (sz & 1)
    assert fmt and data is not None and fmt[0] == 1 and fmt[5] == 16, "only PCM_16 is supported"
    a = np.frombuffer(data, "<i2").astype(np.float32) / np.float32(32768.0)
    return (a if fmt[1] == 1 else a.reshape(-1, fmt[1])), fmt[2]


def iter_chunks(samples, frames: int = CHUNK_FRAMES, as_bytes: bool = True):
    """Fixed-size playback chunks (the last one short), as s16le bytes or float views."""
    a = np.asarray(samples, dtype=np.float32)
    for i in range(0, len(a), frames):
        c = a[i:i + frames]
        yield f32_to_s16le(c) if as_bytes else c


def concat_wav(parts, sample_rate: int = 24000) -> bytes:
    """Whole-book export: sentences in order -> one WAV image (kq3_1_f.py-style "save as")."""
    buf = io.BytesIO()
    pcm = [float_to_pcm16_sndfile(p) for p in parts]
    write_wav(buf, np.concatenate(pcm) if pcm else np.zeros(0, "<i2"), sample_rate)
    return buf.getvalue()
