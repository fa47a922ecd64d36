"""Oracle parity THROUGH the two front doors, free-running.

One phoneme string goes down both paths and the SAMPLES are compared:
  product   Kokoro(...).create(ph, voice=pack, is_phonemes=True, trim=False)        (speech7.py:127, basic.tts_kokoro.py:30)
            KPipeline(...)(text, voice=pack)                                         (north_star surface B)
  checker   text.split_phonemes -> text.tokenize -> [0, *ids, 0], style row voice[len(ids)] (kokoro_onnx) resp.
            pack[len(ids) - 1] (KPipeline), noise = counter_noise(content seed) -> oracle.forward_with_tokens per chunk,
            chunks concatenated.
Bar: integer durations bit-exact (lengths, and pred_dur where the front door returns it), waveform max-abs <= 1e-3.
No teacher forcing and no debug hooks: the weights are the "calm" synthetic set (spec.synthetic_state_dicts(lstm_gain=1,
f0_gain=1, f0_bias=0): every frame unvoiced, so no F0 is integrated into a phase and two fp32 implementations stay
sample-comparable end to end -- DESIGN.md section 6).  The chunk and style-row rules are [EXT] (kokoro_onnx 0.4.9 is
not installable): this pins engine + host glue == restatement, not the restatement itself.
"""
import warnings

import numpy as np
import pytest
import torch

from tts_amd import spec, text as T
from tts_amd.kokoro import SYNTHETIC_SETS, _content_seed
from oracle import kokoro_ref as R

pytestmark = pytest.mark.gpu
AUDIO_TOL = 1e-3

# IPA strings over the Kokoro vocab; the last two exceed 510 phonemes and must chunk
_BASE = "ðə kwˈɪk bɹˈaʊn fˈɑːks dʒˈʌmps ˌoʊvɚ ðə lˈeɪzi dˈɑːɡ"
PHONEMES = [
    "hɛlˈoʊ wˈɜːld",
    "ðɪs ɪz ðə fˈɜːst sˈɛntəns. ðɪs ɪz sˈɛkənd sˈɛntəns!",
    _BASE + ", ænd ðˈɛn ɪt ɹˈʌnz əwˈeɪ; wˈaɪ?",
    "ˈa",
    ". ".join([_BASE] * 11) + ".",                    # ~580 phonemes: two chunks
    ", ".join(["bˈʌt nˈoʊ wˈʌn nˈuː wˈaɪ ðə mˈuːn ɹˈoʊz sˈoʊ ˈɜːli"] * 24) + "?",     # ~1100 phonemes: three chunks
]


@pytest.fixture(scope="module")
def calm():
    warnings.filterwarnings("ignore")
    sds = spec.synthetic_state_dicts(0, **SYNTHETIC_SETS["synthetic-calm"])
    return R.build(spec.DEFAULT_CONFIG, sds)


def _oracle(model, ph, pack, convention, speed):
    """What the front door is specified to compute: (audio f32, [pred_dur per chunk], [ids per chunk])."""
    chunks = T.split_phonemes(ph) or [""]
    audio, durs, idl = [], [], []
    for ch in chunks:
        ids = T.tokenize(ch[:T.MAX_PHONEME_LENGTH])[:T.MAX_PHONEME_LENGTH]
        row = T.style_row(pack, len(ids), convention)
        x = torch.tensor([[0] + ids + [0]], dtype=torch.long)
        rs = torch.from_numpy(row[None].copy())
        pd, _, _ = model.durations(x, rs, speed)
        noise = torch.from_numpy(R.counter_noise(_content_seed(ids), int(pd.sum()) * 600))[None]
        a, pd2 = model.forward_with_tokens(x, rs, speed, noise=noise)
        assert torch.equal(pd, pd2)
        audio.append(a.numpy())
        durs.append(pd.numpy())
        idl.append(ids)
    return np.concatenate(audio), durs, idl


def test_kokoro_create_matches_the_oracle_sample_by_sample(calm):
    from tts_amd import Kokoro
    k = Kokoro("synthetic-calm", None)
    pack = k.get_voice_style("af_synthetic")
    n_chunks = []
    for ph, speed in zip(PHONEMES, (1.0, 1.0, 0.8, 1.0, 2.0, 1.6)):      # fast speech on the long ones: fewer frames for the CPU oracle
        got, sr = k.create(ph, voice=pack, speed=speed, is_phonemes=True, trim=False)
        want, durs, idl = _oracle(calm, ph, pack, "kokoro_onnx", speed)
        n_chunks.append(len(durs))
        assert sr == 24000 and got.dtype == np.float32
        assert got.size == want.size == 600 * sum(int(d.sum()) for d in durs), (len(ph), got.size, want.size)   # durations
        err = float(np.abs(got - want).max())
        print(f"Kokoro.create: {len(ph)} phonemes, {len(durs)} chunk(s), speed {speed}: {got.size / 24000:.1f} s, max-abs vs oracle {err:.2e}")
        assert err <= AUDIO_TOL, (ph[:40], err)
    assert n_chunks[-2] >= 2 and n_chunks[-1] >= 3 and all(len(i) <= 510 for i in idl)
    # the string form of the voice and the batched door give the same samples
    a1, _ = k.create(PHONEMES[1], voice="af_synthetic", is_phonemes=True, trim=False)
    a2 = k.create_batch([PHONEMES[0], PHONEMES[1]], voice=pack, is_phonemes=True, trim=False)[1][0]
    w, _, _ = _oracle(calm, PHONEMES[1], pack, "kokoro_onnx", 1.0)
    assert np.array_equal(a1, a2) and float(np.abs(a1 - w).max()) <= AUDIO_TOL


def test_kpipeline_matches_the_oracle_sample_by_sample(calm):
    from tts_amd import KModel, KPipeline
    pipe = KPipeline(lang_code="a", model=KModel(model="synthetic-calm"))
    pack = spec.synthetic_voice_pack(3)
    text = "Hello world, this is a test.\nAnd a second line, somewhat longer than the first one!"
    res = list(pipe(text, voice=pack, speed=1))
    assert len(res) == 2
    for r in res:
        want, durs, _ = _oracle(calm, r.phonemes, pack, "kpipeline", 1.0)
        assert len(durs) == 1
        assert np.array_equal(r.pred_dur.numpy(), durs[0]), (r.pred_dur, durs[0])              # bit-exact durations
        got = r.audio.numpy()
        assert got.size == want.size
        err = float(np.abs(got - want).max())
        print(f"KPipeline: {len(r.phonemes)} phonemes: max-abs vs oracle {err:.2e}")
        assert err <= AUDIO_TOL, err
    # KModel.forward, the call KPipeline wraps
    m = pipe.model
    ph = PHONEMES[2]
    ids = T.tokenize(ph)
    row = T.style_row(pack, len(ids), "kpipeline")
    out = m(ph, torch.from_numpy(row.copy()), speed=1, return_output=True)
    x = torch.tensor([[0] + ids + [0]], dtype=torch.long)
    rs = torch.from_numpy(row[None].copy())
    pd, _, _ = calm.durations(x, rs, 1.0)
    noise = torch.from_numpy(R.counter_noise(_content_seed(ids), int(pd.sum()) * 600))[None]
    a, _ = calm.forward_with_tokens(x, rs, 1.0, noise=noise)
    assert torch.equal(out.pred_dur, pd)
    assert float((out.audio - a).abs().max()) <= AUDIO_TOL
