"""Host rows against outputs of the REFERENCE'S OWN functions (tests/golden/host_reference_functions.json.gz, written
by tools/make_host_fixtures.py in the build container: the reference's definitions were taken out of its scripts by
name and evaluated there; only inputs and outputs are committed).  Bit for bit:

  text.split_sentences_speech7   vs TTSEngine.split_sentences           speech7.py:108-122
  text.split_sentences_basic     vs tokenize                            basic.tts_kokoro_2_play_pause_works.py:24
  text.split_sentences_newline   vs split_sentences                     tts_16_3.py:27-29
  text.split_sentences_abbrev    vs _split_text_into_sentences          epubviewer_06_latest_1_nov_2025_theme.py:2458-2499
  text.stable_id_for_text        vs stable_id_for_text                  tts_16_3.py:32-35
  wav.f32_to_s16le               vs f32_to_s16le                        basic.tts_kokoro_2_play_pause_works.py:25
and the read-aloud scheduler's own split_sentences (readaloud.TTSEngine) against the speech7 outputs.
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from tts_amd import text as T, wav
from tts_amd.readaloud import TTSEngine

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))

SPLITTERS = {"speech7": T.split_sentences_speech7, "basic": T.split_sentences_basic, "newline": T.split_sentences_newline,
             "abbrev": T.split_sentences_abbrev}


@pytest.fixture(scope="module")
def gold():
    with gzip.open(os.path.join(GOLD, "host_reference_functions.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)


def _sha(obj):
    return hashlib.sha1(json.dumps(obj, ensure_ascii=False, separators=(",", ":")).encode("utf-8")).hexdigest()


def test_adversarial_strings_split_as_the_reference_does(gold):
    eng = TTSEngine()
    assert len(gold["adversarial"]) >= 30
    for rec in gold["adversarial"]:
        t = rec["text"]
        for k, f in SPLITTERS.items():
            assert f(t) == rec[k], (k, t, f(t), rec[k])
        assert eng.split_sentences(t) == rec["speech7"], t
        assert T.stable_id_for_text(t) == rec["sid"]


def test_art_of_war_corpus_splits_as_the_reference_does(gold):
    """294 chunks (40 sentences each, joined by ' ' and by newline) of the committed Art-of-War text: sentence count and
    sha1 of every splitter's output equal what the reference's functions produced on the same strings."""
    import make_host_fixtures as M
    chunks = M.corpus_chunks()
    assert len(chunks) == len(gold["corpus"]) > 200
    for t, rec in zip(chunks, gold["corpus"]):
        assert hashlib.sha1(t.encode("utf-8")).hexdigest() == rec["input_sha1"]
        for k, f in SPLITTERS.items():
            r = f(t)
            assert [len(r), _sha(r)] == rec[k], (k, t[:60])
        assert _sha([T.stable_id_for_text(s) for s in T.split_sentences_newline(t)]) == rec["sids"]


def test_pcm16_conversion_equals_the_reference_expression(gold):
    """clip(x, -1, 1) * 32767 -> '<i2' on the edge cases (exact +-1, just beyond, +-inf, every 16-bit step and half step)."""
    import make_host_fixtures as M
    cases = M.pcm_cases()
    assert set(cases) == set(gold["pcm"])
    for name, a in cases.items():
        with np.errstate(all="ignore"):
            b = wav.f32_to_s16le(a)
        g = gold["pcm"][name]
        assert len(b) == 2 * g["n"]
        assert np.frombuffer(b, "<i2")[:32].tolist() == g["head"], name
        assert hashlib.sha1(b).hexdigest() == g["sha1"], name
        assert np.array_equal(wav.f32_to_s16le_array(a), np.frombuffer(b, "<i2")), name
