"""WAV / PCM16 output helpers, EPUB front end, voice tables."""
import gzip
import io
import json
import os
import struct
import wave
import zipfile

import numpy as np
import pytest

from tts_amd import epub, text as T, voices, wav

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_wav_header_and_rounding():
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1.5, -1.5, 0.5 / 32767, 1.5 / 32767, 2.5 / 32767], np.float32)
    b = wav.wav_bytes(x, 24000)
    assert len(b) == 44 + 2 * len(x) and b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt "
    assert struct.unpack("<IHHIIHH", b[16:36]) == (16, 1, 1, 24000, 48000, 2, 16)
    pcm = np.frombuffer(b[44:], "<i2")
    # libsndfile: lrint(x * 32767) (ties to even), saturating
    assert pcm.tolist() == [0, 16384, -16384, 32767, -32767, 32767, -32768, 0, 2, 2]
    with wave.open(io.BytesIO(b)) as w:                         # the stdlib reader accepts the file
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 24000, len(x))


def test_wav_roundtrip_and_f32_to_s16le(tmp_path):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(5000) * 0.3).astype(np.float32)
    p = tmp_path / "a.wav"
    wav.write_wav(str(p), x)
    y, sr = wav.read_wav(str(p))
    assert sr == 24000 and y.shape == x.shape and np.abs(y - np.clip(x, -1, 1)).max() <= 2.0 / 32767   # write x32767, read /32768
    # f32_to_s16le truncates toward zero (basic.tts_kokoro_2_play_pause_works.py:25)
    assert np.frombuffer(wav.f32_to_s16le([0.99999, -0.99999, 2.0, 1e-5]), "<i2").tolist() == [32766, -32766, 32767, 0]
    chunks = list(wav.iter_chunks(x))
    assert len(chunks) == 3 and [len(c) for c in chunks] == [4800, 4800, 400]
    assert b"".join(chunks) == wav.f32_to_s16le(x)
    whole = wav.concat_wav([x[:100], x[100:]])
    assert whole == wav.wav_bytes(x)


CHAPTER = """<?xml version="1.0"?><html><head><title>T. Itle</title><style>p{}</style></head><body>
<h1>Chapter One</h1>
<p><span class="dropcap">I</span>t was Dr. Smith who came. He said: "Go!" <i>Then</i> he left.</p>
<div>Outer text. <p>Inner one. Inner two.</p> tail text</div>
<!-- a comment. Not text. -->
<script>var x = "No. Way.";</script>
<ul><li>First item</li><li>Second item. With two.</li></ul>
loose text. More loose.
<p><span style="float: left">Once</span> a longer drop stays a word.</p>
</body></html>"""


def test_sentences_from_html_rules():
    got = [s["text"] for s in epub.sentences_from_html(CHAPTER)]
    assert got == ["Chapter One", "It was Dr. Smith who came.", 'He said: "Go!" Then he left.',   # no split after !" (rule :2488)
                   "Outer text. tail text", "Inner one.", "Inner two.", "First item", "Second item.", "With two.",
                   "loose text.", "More loose.", "Once a longer drop stays a word."]
    assert [s["sid"] for s in epub.sentences_from_html(CHAPTER)] == list(range(len(got)))


def _make_epub(path):
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("mimetype", "application/epub+zip")
        z.writestr("META-INF/container.xml", '<?xml version="1.0"?><container version="1.0" '
                   'xmlns="urn:oasis:names:tc:opendocument:xmlns:container"><rootfiles><rootfile '
                   'full-path="OEBPS/content.opf" media-type="application/oebps-package+xml"/></rootfiles></container>')
        z.writestr("OEBPS/content.opf", '<?xml version="1.0"?><package xmlns="http://www.idpf.org/2007/opf" version="2.0">'
                   '<metadata xmlns:dc="http://purl.org/dc/elements/1.1/"><dc:title>Two Chapters</dc:title></metadata>'
                   '<manifest><item id="b" href="text/b.xhtml" media-type="application/xhtml+xml"/>'
                   '<item id="a" href="text/a%20one.xhtml" media-type="application/xhtml+xml"/>'
                   '<item id="css" href="s.css" media-type="text/css"/></manifest>'
                   '<spine><itemref idref="a"/><itemref idref="css"/><itemref idref="b"/></spine></package>')
        z.writestr("OEBPS/text/a one.xhtml", "<html><body><p>Alpha one. Alpha two.</p></body></html>")
        z.writestr("OEBPS/text/b.xhtml", "<html><body><p>Beta.</p></body></html>")
        z.writestr("OEBPS/s.css", "p{}")


def test_epub_spine_order(tmp_path):
    p = tmp_path / "t.epub"
    _make_epub(str(p))
    b = epub.EpubBook(str(p))
    assert b.title == "Two Chapters" and len(b) == 2
    assert [h for h, _ in b.items] == ["OEBPS/text/a one.xhtml", "OEBPS/text/b.xhtml"]
    assert b.all_sentences() == [(0, 0, "Alpha one."), (0, 1, "Alpha two."), (1, 0, "Beta.")]


@pytest.mark.parametrize("key,fn", [("art_of_war", "The-Art-of-War.epub"), ("sound_of_sirens", "Sound-of-Sirens.epub")])
def test_epub_golden_fixture(key, fn):
    g = json.load(gzip.open(os.path.join(GOLD, f"epub_{key}.json.gz")))
    assert g["n_sentences"] == len(g["char_lengths"]) == sum(g["per_item"])
    src = os.path.join("/root/reference", fn)
    if not os.path.exists(src):                                 # the books only exist in the build container
        pytest.skip("reference epub not present")
    import hashlib
    s = epub.EpubBook(src).all_sentences()
    assert len(s) == g["n_sentences"] and [len(t) for _, _, t in s] == g["char_lengths"]
    assert hashlib.sha256("\n".join(t for _, _, t in s).encode()).hexdigest() == g["sha256"]


def test_voice_tables():
    assert voices.voices_for_language("en-gb")[0] == "bf_alice" and len(voices.voices_for_language("en-us")) == 20
    assert voices.voices_for_language("xx") == voices.voices_for_language("en-us")       # kq3_1_f.py:300,378
    assert voices.language_of_voice("pf_dora") == "pt-br" and voices.gender_of_voice("zm_yunxi") == "male"
    assert len(voices.all_voices()) == 54 and voices.DEFAULT_VOICE in voices.all_voices()
    assert set(voices.VOICES_BY_PREFIX) == set(T.LANG_CODES)
