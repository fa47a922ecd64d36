"""GPU parity tests: the HIP engine (through the C-ABI, via ctypes) against the CPU oracle.

Bars (BASELINE.json north_star): predicted integer durations bit-exact; fp32 waveform max-abs
error <= 1e-3 with injected source noise.  The waveform bar is checked with the oracle's F0/N
curves and the source STFT tensor forced to the engine's (teacher forcing) at the two places
where the model's own arithmetic is ill-conditioned: the harmonic source INTEGRATES F0 (a 1e-3 Hz
fp32 disagreement grows to ~0.5 rad at the 9th harmonic within seconds) and angle() of a
near-empty STFT bin flips by 2*pi on last-bit noise -- see DESIGN.md "numerical conditioning".
Both forced tensors are themselves compared against the oracle's own (F0/N to 5e-4 relative,
har modulo 2*pi).  The free-running waveform
is reported as an SNR.  The oracle is unpinned by the reference, so these tests
prove engine == restatement, not engine == kokoro_onnx.
"""
import numpy as np
import pytest
import torch

import helpers as H
from tts_amd import spec
from oracle import kokoro_ref as R

pytestmark = pytest.mark.gpu

AUDIO_TOL = 1e-3          # north_star: fp32 max-abs error <= 1e-3


def _noise_for(nfr, base=100):
    return np.concatenate([R.counter_noise(base + b, int(nfr[b]) * 600) for b in range(len(nfr))])


# ------------------------------------------------------------------ the hot conv kernel
@pytest.mark.parametrize("cin,cout,L,taps,dil,act", [
    (128, 128, 777, 7, 3, 2),      # generator stage-1 resblock shape, snake
    (256, 256, 300, 11, 5, 2),     # widest halo (50)
    (22, 128, 1000, 1, 1, 0),      # noise_convs[1]: cin padded 22 -> 32
    (514, 1024, 130, 3, 1, 1),     # decoder encode conv1: cin padded 514 -> 528, leaky
    (128, 22, 513, 7, 1, 1),       # conv_post: cout padded 22 -> 64 (1x4 variant)
    (512, 50, 70, 1, 1, 0),        # duration_proj
    (768, 2304, 64, 1, 1, 0),      # ALBERT fused QKV
    (16, 128, 1, 3, 1, 0),         # single position
])
def test_conv_kernel_matches_torch(engine, cin, cout, L, taps, dil, act):
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal((2, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    alpha = (1.0 + 0.5 * rng.random(cin)).astype(np.float32)
    y, _ = engine.op_conv1d(x, w, b, dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None)
    xt = torch.from_numpy(x)
    if act == 1:
        xt = torch.nn.functional.leaky_relu(xt, 0.2)
    elif act == 2:
        a = torch.from_numpy(alpha)[None, :, None]
        xt = xt + (1 / a) * torch.sin(a * xt) ** 2
    ref = torch.nn.functional.conv1d(xt.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                     padding=(taps - 1) // 2 * dil, dilation=dil).numpy()
    err = np.abs(y - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), err


# ------------------------------------------------------------------ durations: bit-exact
@pytest.mark.parametrize("lengths,speed", [
    ([1, 7, 30, 64, 2, 19], 1.0),
    ([120, 3, 254], 1.0),
    ([40, 41, 42], 0.5),
    ([40, 41, 42], 2.0),
    ([510], 1.0),                    # maximum context: 510 phonemes + 2 pads
    ([0], 1.0),                      # empty phoneme string: ids = [0, 0]
])
def test_durations_bit_exact(engine, oracle_model, lengths, speed):
    id_lists, ref_s = H.make_batch(11, [max(n, 0) for n in lengths])
    ref_s = np.stack([spec.synthetic_voice_pack(1)[max(len(x) - 1, 0), 0] for x in id_lists])
    ids, lens = engine.pack(id_lists)
    engine.debug_enable(True)
    try:
        pred, nfr = engine.durations(ids, lens, ref_s, speed=speed)
        raw_eng = engine.debug_get("duration")[:, 0, :]
    finally:
        engine.debug_enable(False)
    margin, noise = 1.0, 0.0
    for b, x in enumerate(id_lists):
        t = torch.tensor([[0] + x + [0]], dtype=torch.long)
        pd, _, raw = oracle_model.durations(t, torch.from_numpy(ref_s[b:b + 1].copy()), speed)
        raw = raw.numpy().reshape(-1)
        frac = np.abs((raw % 1.0) - 0.5)
        diff = np.abs(raw_eng[b, :lens[b]] - raw)
        margin = min(margin, float(frac.min()))
        noise = max(noise, float(diff.max()))
        assert np.array_equal(pd.numpy(), pred[b, :lens[b]]), (b, pd.numpy(), pred[b, :lens[b]])
        # what bit-exactness rests on, token by token: the engine sums the 50 sigmoids in fp64 and rounds once, torch
        # sums in fp32 in its own order, the LSTM stack before them amplifies summation-order noise -- the integers
        # agree because every pre-rounding value sits further from its rounding boundary than the two disagree
        # (clamp(min=1) makes the 0.5 boundary irrelevant: both sides of it give 1)
        assert (diff[raw >= 1.0] < frac[raw >= 1.0]).all(), (b, float(diff.max()), float(frac.min()))
        assert (pred[b, lens[b]:] == 0).all()
        assert nfr[b] == pd.sum()
    print(f"min |frac(duration) - 0.5| = {margin:.2e}; max |raw_engine - raw_oracle| = {noise:.2e} "
          "(tests/test_gpu_golden.py arbitrates this noise against float64)")
    assert noise <= 2e-3, noise


# ------------------------------------------------------------------ every stage, then the waveform
STAGE_TOL = {  # max-abs tolerances, scaled by max(1, max|ref|) below.  The synthetic duration-branch
    # LSTMs carry x3 weights (spec.synthetic_state_dicts) and sit near the chaotic regime: fp32
    # summation-order noise (~1e-6 per gate) is amplified to ~2e-4 over 35 recurrent steps x 4 layers.
    "default": 5e-4,
    # the two outputs of that recurrent stack: the same noise after 64-90 steps is ~2e-4..6e-4 depending on the
    # (equally valid) summation order of the LayerNorm / LSTM kernels; the integer durations computed from them
    # are asserted bit-exact above, with the rounding margin printed by test_durations_bit_exact
    "d": 2e-3,
    "dur_lstm": 2e-3,
    "har": None,            # phase rows flip by 2*pi on rounding noise where |X_k| ~ 0: checked via its consumers
}


@pytest.mark.parametrize("lengths", [[10, 33, 20], [64], [5, 90]])
def test_stage_and_waveform_parity(engine, oracle_model, lengths):
    id_lists, ref_s = H.make_batch(3, lengths)
    engine.debug_enable(True)
    try:
        ids, lens = engine.pack(id_lists)
        pred, nfr = engine.durations(ids, lens, ref_s)
        noise = _noise_for(nfr)
        audio, off = engine.decode(noise=noise)
        assert off[-1] == audio.size and np.isfinite(audio).all()
        for b, x in enumerate(id_lists):
            T, F = int(lens[b]), int(nfr[b])
            taps = {}
            nz = noise.reshape(-1, 9)[off[b]:off[b + 1]]
            f0g = engine.debug_get("F0")[b, 0, :2 * F]
            ng = engine.debug_get("N")[b, 0, :2 * F]
            harg = engine.debug_get("har")[b][:22, :120 * F + 1]
            a_ref, pd = H.oracle_run(oracle_model, x, ref_s[b], 1.0, nz, taps=taps, f0=f0g, n=ng, har=harg)
            assert np.array_equal(pd, pred[b, :T])
            for name, base, mul, add in H.STAGES:
                ref = H.to_bct(name, taps[name])
                L = (T if base == "T" else F) * mul + add
                got = engine.debug_get(name)[b][: ref.shape[0], :L]
                if name == "har":
                    # the oracle's OWN STFT of its own har_source; phases compared modulo 2*pi
                    mag_err, phase_err, flips, cerr = H.compare_har(got, H.to_bct("har", taps["har_own"]))
                    print(f"sentence {b}: har mag err {mag_err:.1e}, strong-bin phase err {phase_err:.1e}, "
                          f"complex err {cerr:.1e}, 2*pi flips {flips} of {got[11:].size}")
                    assert mag_err <= 1e-5 and phase_err <= 1e-3 and cerr <= 1e-5
                    continue
                tol = STAGE_TOL.get(name, STAGE_TOL["default"]) * max(1.0, float(np.abs(ref).max()))
                err = float(np.abs(got - ref[:, :L]).max())
                assert err <= tol, (name, b, err, tol)
            err = float(np.abs(audio[off[b]:off[b + 1]] - a_ref).max())
            assert err <= AUDIO_TOL, (b, err)
    finally:
        engine.debug_enable(False)


def test_free_running_waveform(engine, oracle_model):
    """No teacher forcing.  Durations stay bit-exact; the waveform decorrelates in PHASE because the
    source integrates F0 (engine and oracle F0 agree to ~3e-5 relative, i.e. ~5e-3 Hz, which is
    ~0.5 rad at the 9th harmonic after 2-3 s), so sample-wise SNR is only reported, while the
    phase-insensitive spectral-convergence error of the 1024-point magnitude STFT is asserted."""
    id_lists, ref_s = H.make_batch(5, [12, 25])
    ids, lens = engine.pack(id_lists)
    pred, nfr = engine.durations(ids, lens, ref_s)
    noise = _noise_for(nfr)
    audio, off = engine.decode(noise=noise)
    win = torch.hann_window(1024)
    for b, x in enumerate(id_lists):
        a_ref, pd = H.oracle_run(oracle_model, x, ref_s[b], 1.0, noise.reshape(-1, 9)[off[b]:off[b + 1]])
        assert np.array_equal(pd, pred[b, :lens[b]])
        got = audio[off[b]:off[b + 1]]
        e = got.astype(np.float64) - a_ref
        snr = 10 * np.log10((a_ref.astype(np.float64) ** 2).sum() / max((e ** 2).sum(), 1e-30))
        S1 = torch.stft(torch.from_numpy(got.copy()), 1024, 256, window=win, return_complex=True).abs()
        S0 = torch.stft(torch.from_numpy(a_ref.copy()), 1024, 256, window=win, return_complex=True).abs()
        sc = float(torch.linalg.norm(S1 - S0) / torch.linalg.norm(S0))
        print(f"sentence {b}: free-running SNR {snr:.1f} dB, max-abs {np.abs(e).max():.2e}, "
              f"spectral convergence error {sc:.3f}")
        assert sc <= 0.2


# ------------------------------------------------------------------ size-independent properties
def test_padding_is_bit_invisible(engine):
    """A sentence synthesized alone and inside a ragged batch gives identical bits."""
    id_lists, ref_s = H.make_batch(21, [37, 9, 150, 64])
    seeds = np.array([7, 8, 9, 10], dtype=np.uint64)
    ids, lens = engine.pack(id_lists)
    pred, nfr = engine.durations(ids, lens, ref_s)
    audio, off = engine.decode(seeds=seeds)
    audio = audio.copy()
    for b in (0, 1, 3):
        i1, l1 = engine.pack([id_lists[b]])
        p1, n1 = engine.durations(i1, l1, ref_s[b:b + 1])
        a1, o1 = engine.decode(seeds=seeds[b:b + 1])
        assert np.array_equal(p1[0, :l1[0]], pred[b, :lens[b]])
        assert np.array_equal(a1, audio[off[b]:off[b + 1]])


def test_deterministic_and_full_size_properties(engine):
    """BASELINE configs[1] shape (64 x 256 tokens): two runs are bit-identical, every sample is
    finite and within [-1.5, 1.5]-ish iSTFT range, offsets are 600 * sum(pred_dur)."""
    rng = np.random.default_rng(0)
    B, T = 64, 256
    ids = np.zeros((B, T), dtype=np.int64)
    ids[:, 1:T - 1] = rng.integers(1, 178, size=(B, T - 2))
    lens = np.full((B,), T, dtype=np.int32)
    ref_s = np.repeat(spec.synthetic_voice_pack(1)[T - 3], B, axis=0)
    seeds = np.arange(B, dtype=np.uint64)
    pred, nfr = engine.durations(ids, lens, ref_s)
    a1, off = engine.decode(seeds=seeds)
    a1 = a1.copy()
    assert (pred >= 1).all() and np.array_equal(nfr, pred.sum(axis=1))
    assert np.array_equal(np.diff(off), 600 * nfr)
    assert np.isfinite(a1).all()
    pred2, _ = engine.durations(ids, lens, ref_s)
    a2, _ = engine.decode(seeds=seeds)
    assert np.array_equal(pred, pred2) and np.array_equal(a1, a2)
    # row 5 alone == row 5 in the batch (padding / batching invisible at full size too)
    p5, _ = engine.durations(ids[5:6], lens[5:6], ref_s[5:6])
    a5, _ = engine.decode(seeds=seeds[5:6])
    assert np.array_equal(p5[0], pred[5]) and np.array_equal(a5, a1[off[5]:off[6]])


def test_forced_durations_and_max_context(engine, oracle_model):
    """configs[3] shape per sentence: 510 phonemes, forced pred_dur; checked against the oracle."""
    id_lists, ref_s = H.make_batch(31, [510])
    ids, lens = engine.pack(id_lists)
    forced = np.ones((1, 512), dtype=np.int32)
    forced[0, ::7] = 3
    engine.debug_enable(True)
    try:
        pred, nfr = engine.durations(ids, lens, ref_s, forced_dur=forced)
        assert np.array_equal(pred, forced)
        noise = _noise_for(nfr)
        audio, off = engine.decode(noise=noise)
        F = int(nfr[0])
        a_ref, _ = H.oracle_run(oracle_model, id_lists[0], ref_s[0], 1.0, noise.reshape(-1, 9), forced=forced[0],
                                f0=engine.debug_get("F0")[0, 0, :2 * F], n=engine.debug_get("N")[0, 0, :2 * F],
                                har=engine.debug_get("har")[0][:22, :120 * F + 1])
        assert np.abs(audio - a_ref).max() <= AUDIO_TOL
    finally:
        engine.debug_enable(False)


def test_in_kernel_noise_matches_counter_noise(engine):
    """seeds-mode (splitmix64 + Box-Muller in the kernel) == explicit noise from its numpy twin."""
    id_lists, ref_s = H.make_batch(41, [14, 6])
    ids, lens = engine.pack(id_lists)
    _, nfr = engine.durations(ids, lens, ref_s)
    seeds = np.array([123, 456], dtype=np.uint64)
    a_seed, off = engine.decode(seeds=seeds)
    a_seed = a_seed.copy()
    noise = np.concatenate([R.counter_noise(int(seeds[b]), int(nfr[b]) * 600) for b in range(2)])
    engine.durations(ids, lens, ref_s)
    a_exp, _ = engine.decode(noise=noise)
    assert np.abs(a_seed - a_exp).max() <= 1e-4


def test_errors_are_reported_not_fatal(engine):
    ids = np.zeros((1, 4), dtype=np.int64)
    with pytest.raises(ValueError):
        engine.durations(ids, np.array([9], dtype=np.int32), np.zeros((1, 256), np.float32))   # lens > Tmax
    with pytest.raises(ValueError):
        engine.durations(ids, np.array([4], dtype=np.int32), np.zeros((1, 256), np.float32), speed=0.0)
    # the handle still works afterwards
    pred, nfr = engine.durations(ids, np.array([4], dtype=np.int32), np.zeros((1, 256), np.float32))
    assert nfr[0] >= 4


def test_pcm16_matches_numpy(engine):
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(10007) * 0.7).astype(np.float32)
    got = engine.pcm16(x)
    ref = (np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2")     # basic.tts_kokoro_2_play_pause_works.py:25
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------ committed golden vector
def test_engine_matches_committed_golden(engine):
    """tests/golden/synthetic_T14.npz (tools/make_golden.py, CPU oracle): durations bit-exact,
    token-rate stages and F0 within fp32 noise."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "synthetic_T14.npz"))
    ids = g["ids"][None].astype(np.int64)
    lens = np.array([ids.shape[1]], dtype=np.int32)
    engine.debug_enable(True)
    try:
        pred, nfr = engine.durations(ids, lens, g["ref_s"][None])
        assert np.array_equal(pred[0], g["pred_dur"])
        audio, off = engine.decode(seeds=np.array([int(g["noise_seed"])], dtype=np.uint64))
        assert audio.size == int(g["audio_len"])
        T, F = int(lens[0]), int(nfr[0])
        assert np.abs(engine.debug_get("duration")[0, 0, :T] - g["duration"]).max() < 5e-4
        assert np.abs(engine.debug_get("d")[0][:, :T].T - g["d"]).max() < 1e-3
        assert np.abs(engine.debug_get("t_en")[0][:, :T] - g["t_en"]).max() < 1e-4
        assert np.abs(engine.debug_get("F0")[0, 0, :2 * F] - g["F0"]).max() < 5e-2
    finally:
        engine.debug_enable(False)


# ------------------------------------------------------------------ the two front doors
def test_kokoro_front_door(engine):
    import threading
    from tts_amd import Kokoro
    k = Kokoro("synthetic", None)
    assert k.get_voices() == ["af_synthetic"]
    samples, sr = k.create("This is the 1st sentence. This is 2nd sentence.", voice="af_synthetic", speed=1.0,
                           lang="en-us")                                   # basic.tts_kokoro.py:30
    assert sr == 24000 and samples.dtype == np.float32 and samples.ndim == 1 and samples.size > 2400
    assert np.isfinite(samples).all()
    # phoneme input, no trim: length is exactly 600 * sum(pred_dur), identical to the raw engine
    ph = "hɛlˈoʊ wɜːld"
    s2, _ = k.create(ph, voice=k.get_voice_style("af_synthetic"), is_phonemes=True, trim=False)
    from tts_amd import text as T
    ids = T.tokenize(ph)
    pred, nfr = engine.durations(*engine.pack([ids]), T.style_row(k.voices["af_synthetic"], len(ids), "kokoro_onnx")[None])
    assert s2.size == 600 * int(nfr[0])
    with pytest.raises(AssertionError):
        k.create("x", voice="af_synthetic", speed=3.0)
    with pytest.raises(KeyError):
        k.create("x", voice="no_such_voice")
    # two threads on one object (epubviewer_06_latest_1_nov_2025_theme.py:407,692,715)
    out = {}

    def work(i):
        out[i] = k.create(f"Sentence number {i} is spoken here.", voice="af_synthetic")[0]
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(out) == 4 and all(np.isfinite(v).all() and v.size > 0 for v in out.values())


def test_kpipeline_front_door():
    from tts_amd import KModel, KPipeline
    model = KModel(model="synthetic")
    pipe = KPipeline(lang_code="a", model=model)
    res = list(pipe("First line of text.\nSecond line!", voice="af_heart", speed=1))
    assert len(res) == 2
    for gs, ps, audio in res:
        assert isinstance(gs, str) and isinstance(ps, str) and audio.dtype == torch.float32 and audio.numel() > 0
    assert res[0].graphemes == "First line of text." and res[0].pred_dur.sum() * 600 == res[0].audio.numel()
    out = model("hɛlˈoʊ", torch.from_numpy(spec.synthetic_voice_pack(2)[4]), speed=1, return_output=True)
    assert out.audio.numel() == 600 * int(out.pred_dur.sum()) and len(out.pred_dur) == 8
    quiet = KPipeline(lang_code="a", model=False)
    assert [r.audio for r in quiet("Only phonemes.")] == [None]


@pytest.mark.gpu
def test_readaloud_on_gpu():
    """TTSEngine over the real engine: every sentence is played once, in order; the awaited sentence is
    synthesized alone and the rest of the look-ahead window arrives as one batched call whose audio equals
    what single-sentence calls give (seeded noise aside: compare lengths = durations)."""
    from tts_amd import Kokoro
    from tts_amd.readaloud import NullSink, TTSEngine
    k = Kokoro("synthetic", None)
    sents = ["The quick brown fox.", "It jumps over the lazy dog!", "Again and again?", "Then it rests.", "The end."]
    sink = NullSink()
    e = TTSEngine(k, sink, debounce=0.05)
    fin = []
    e.speak_sentences_list(sents, voice="af_synthetic", finished_callback=lambda: fin.append(1))
    assert e.wait(60.0) and fin == [1]
    assert [i for i, _ in sink.played] == [0, 1, 2, 3, 4]
    assert e.synth_log[0][0] == (0,) and any(len(ix) > 1 for ix, _ in e.synth_log)
    singles = [len(k.create(t, voice="af_synthetic", trim=False)[0]) for t in sents]
    batch = [len(a) for a, _ in k.create_batch(sents, voice="af_synthetic", trim=False)]
    assert singles == batch and all(n % 600 == 0 and n > 0 for n in batch)


@pytest.mark.gpu
def test_bf16_conv_op_matches_fp32_reference(engine):
    """bf16 MFMA mode of the conv kernel (k82_set_precision 1) against torch fp32 conv1d: operands are rounded to
    bf16 (2^-9 relative each), accumulation is fp32, so the error of a K-term dot product is ~2^-8 * |x||w| /
    sqrt(K) of the output scale: bound 2e-2 of max|ref|, and the fp32 mode must be restored afterwards."""
    import torch
    rng = np.random.default_rng(5)
    engine.set_precision("bf16")
    try:
        for (cin, cout, L, B, k, dil, act) in [(128, 128, 700, 2, 7, 3, 2), (256, 256, 333, 3, 3, 1, 1), (1090, 1024, 97, 2, 3, 1, 0),
                                               (130, 256, 260, 1, 11, 5, 2)]:
            x = rng.standard_normal((B, cin, L)).astype(np.float32)
            w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
            alpha = (1 + 0.5 * rng.random(cin)).astype(np.float32)
            y, _ = engine.op_conv1d(x, w, b, dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None)
            xt = torch.from_numpy(x)
            if act == 1:
                xt = torch.nn.functional.leaky_relu(xt, 0.2)
            elif act == 2:
                a = torch.from_numpy(alpha)[None, :, None]
                xt = xt + torch.sin(a * xt) ** 2 / a
            ref = torch.nn.functional.conv1d(xt, torch.from_numpy(w), torch.from_numpy(b), padding=dil * (k - 1) // 2,
                                             dilation=dil).numpy()
            err = float(np.abs(y - ref).max()) / float(np.abs(ref).max())
            print(f"bf16 conv cin={cin} cout={cout} k={k}: rel err {err:.2e}")
            assert 1e-5 < err < 2e-2                              # really bf16 (not silently fp32), and close
    finally:
        engine.set_precision("fp32")


@pytest.mark.gpu
def test_bf16_decoder_mode_snr(engine):
    """north_star: bf16 waveform SNR >= 40 dB.  Same batch, same noise, fp32 vs bf16 decoder/generator convs:
    durations and the F0/N curves (computed before the switch) must be IDENTICAL, so the harmonic source is the
    same signal in both runs and the SNR measures the bf16 rounding of the decoder + generator only."""
    id_lists, ref_s = H.make_batch(11, [23, 77, 150])
    ids, lens = engine.pack(id_lists)
    engine.debug_enable(True)
    try:
        pred32, nfr = engine.durations(ids, lens, ref_s)
        noise = _noise_for(nfr)
        a32, off = engine.decode(noise=noise)
        f0_32 = engine.debug_get("F0").copy()
        engine.set_precision("bf16")
        pred16, nfr16 = engine.durations(ids, lens, ref_s)
        a16, off16 = engine.decode(noise=noise)
        f0_16 = engine.debug_get("F0").copy()
    finally:
        engine.set_precision("fp32")
        engine.debug_enable(False)
    assert np.array_equal(pred32, pred16) and np.array_equal(off, off16) and np.array_equal(f0_32, f0_16)
    assert not np.array_equal(a32, a16)                           # the mode really changed the arithmetic
    for b in range(len(id_lists)):
        s, e = off[b], off[b + 1]
        snr = 10 * np.log10(float(np.sum(a32[s:e].astype(np.float64) ** 2)) / float(np.sum((a32[s:e] - a16[s:e]).astype(np.float64) ** 2)))
        print(f"sentence {b}: bf16 vs fp32 SNR {snr:.1f} dB")
        assert snr >= 40.0


@pytest.mark.gpu
def test_bf16x3_conv_op_is_fp32_grade(engine):
    """bf16x3 mode (k82_set_precision 2): every operand split as hi + lo (two bf16), three bf16 MFMAs per product,
    fp32 accumulate.  Representation error 2^-17 per operand, dropped lo*lo term 2^-16: the result must sit
    within 3e-5 of max|ref| of a float64 reference (the fp32-MFMA mode's own bar is 2e-5), far from plain bf16
    (~3e-3), and must not be bit-identical to the fp32 mode (the mode really ran)."""
    import torch
    rng = np.random.default_rng(6)
    try:
        for (cin, cout, L, B, k, dil, act) in [(128, 128, 700, 2, 7, 3, 2), (256, 256, 333, 3, 3, 1, 1), (1090, 1024, 97, 2, 3, 1, 0),
                                               (130, 256, 260, 1, 11, 5, 2), (16, 128, 1, 1, 3, 1, 0)]:
            x = rng.standard_normal((B, cin, L)).astype(np.float32)
            w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
            alpha = (1 + 0.5 * rng.random(cin)).astype(np.float32)
            kw = dict(dil=dil, act=act, slope=0.2, alpha=alpha if act == 2 else None)
            engine.set_precision("fp32")
            y32, _ = engine.op_conv1d(x, w, b, **kw)
            engine.set_precision("bf16x3")
            y, _ = engine.op_conv1d(x, w, b, **kw)
            xt = torch.from_numpy(x).double()
            if act == 1:
                xt = torch.nn.functional.leaky_relu(xt, 0.2)
            elif act == 2:
                a = torch.from_numpy(alpha).double()[None, :, None]
                xt = xt + torch.sin(a * xt) ** 2 / a
            ref = torch.nn.functional.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                             padding=dil * (k - 1) // 2, dilation=dil).numpy()
            scale = max(1.0, float(np.abs(ref).max()))
            err, err32 = float(np.abs(y - ref).max()) / scale, float(np.abs(y32 - ref).max()) / scale
            print(f"bf16x3 conv cin={cin} cout={cout} k={k} act={act}: err {err:.2e} (fp32 MFMA mode: {err32:.2e})")
            assert err <= 3e-5
            assert not np.array_equal(y, y32)
    finally:
        engine.set_precision("fp32")


@pytest.mark.gpu
def test_bf16x3_waveform_meets_fp32_bar(engine, oracle_model):
    """The bf16x3 mode against the CPU oracle at the fp32 bar (max-abs <= 1e-3, teacher-forced F0/N/har exactly as
    test_stage_and_waveform_parity): durations bit-exact, F0/N identical to the fp32 mode's (the duration and
    prosody branches never leave the fp32 matrix instruction)."""
    id_lists, ref_s = H.make_batch(3, [10, 33, 64])
    ids, lens = engine.pack(id_lists)
    engine.debug_enable(True)
    try:
        pred32, nfr = engine.durations(ids, lens, ref_s)
        noise = _noise_for(nfr)
        a32, off = engine.decode(noise=noise)
        f0_32 = engine.debug_get("F0").copy()
        engine.set_precision("bf16x3")
        pred, nfr3 = engine.durations(ids, lens, ref_s)
        audio, off3 = engine.decode(noise=noise)
        assert np.array_equal(pred32, pred) and np.array_equal(off, off3)
        assert np.array_equal(f0_32, engine.debug_get("F0")) and not np.array_equal(a32, audio)
        for b, x in enumerate(id_lists):
            T, F = int(lens[b]), int(nfr[b])
            nz = noise.reshape(-1, 9)[off[b]:off[b + 1]]
            f0g = engine.debug_get("F0")[b, 0, :2 * F]
            ng = engine.debug_get("N")[b, 0, :2 * F]
            harg = engine.debug_get("har")[b][:22, :120 * F + 1]
            a_ref, pd = H.oracle_run(oracle_model, x, ref_s[b], 1.0, nz, f0=f0g, n=ng, har=harg)
            assert np.array_equal(pd, pred[b, :T])
            err = float(np.abs(audio[off[b]:off[b + 1]] - a_ref).max())
            err32 = float(np.abs(a32[off[b]:off[b + 1]] - a_ref).max())
            d = (a32[off[b]:off[b + 1]] - audio[off[b]:off[b + 1]]).astype(np.float64)
            snr = 10 * np.log10(float(np.sum(a32[off[b]:off[b + 1]].astype(np.float64) ** 2)) / max(float(np.sum(d * d)), 1e-300))
            print(f"sentence {b}: bf16x3 vs oracle max-abs {err:.2e} (fp32 mode {err32:.2e}); vs fp32 mode SNR {snr:.1f} dB")
            assert err <= AUDIO_TOL
    finally:
        engine.set_precision("fp32")
        engine.debug_enable(False)


@pytest.mark.gpu
def test_sentence_is_bit_identical_alone_and_in_a_large_batch(engine):
    """Small launches take a different conv kernel than large ones (32 x 32 instead of 64 x 64 wave tiles): it is
    written to perform the same arithmetic in the same order, so a sentence must synthesize to the same durations and
    the same samples, bit for bit, alone and inside a batch of 40 (the reference synthesizes one sentence per call)."""
    lengths = [37, 90, 12] + [60 + (7 * i) % 50 for i in range(37)]
    id_lists, ref_s = H.make_batch(21, lengths)
    ids, lens = engine.pack(id_lists)
    pred_b, nfr_b = engine.durations(ids, lens, ref_s)
    noise_b = _noise_for(nfr_b)
    audio_b, off_b = engine.decode(noise=noise_b)
    audio_b = audio_b.copy()
    for k in range(3):
        i1, l1 = engine.pack([id_lists[k]])
        pred_1, nfr_1 = engine.durations(i1, l1, ref_s[k:k + 1])
        assert np.array_equal(pred_1[0, :l1[0]], pred_b[k, :lens[k]])
        a1, off1 = engine.decode(noise=noise_b.reshape(-1, 9)[off_b[k]:off_b[k + 1]].reshape(-1))
        assert np.array_equal(a1, audio_b[off_b[k]:off_b[k + 1]]), float(np.abs(a1 - audio_b[off_b[k]:off_b[k + 1]]).max())


@pytest.mark.gpu
def test_split_and_streaming_lstm_kernels_give_the_same_bits(engine, monkeypatch):
    """lstm_split_kernel (four workgroups per recurrence, W_hh in registers, per-step exchange of h through global
    memory) against lstm_kernel (one workgroup streaming W_hh from L2): same chains, same partial-sum tree -- a second
    engine built with K82_LSTM_SPLIT_MAX_B=0 (streaming kernel everywhere) must reproduce durations and audio exactly,
    on a 40-sentence batch (4 sentences per workgroup: 80 workgroups, one resident round)."""
    from tts_amd.engine import Engine
    lengths = [37, 90, 12, 254] + [40 + (11 * i) % 70 for i in range(36)]
    id_lists, ref_s = H.make_batch(5, lengths)
    ids, lens = engine.pack(id_lists)
    pred_a, nfr_a = engine.durations(ids, lens, ref_s)
    noise = _noise_for(nfr_a)
    audio_a, off_a = engine.decode(noise=noise)
    audio_a = audio_a.copy()
    monkeypatch.setenv("K82_LSTM_SPLIT_MAX_B", "0")
    ref_engine = Engine(spec.synthetic_state_dicts(0))
    try:
        pred_r, nfr_r = ref_engine.durations(ids, lens, ref_s)
        audio_r, off_r = ref_engine.decode(noise=noise)
        assert np.array_equal(pred_a, pred_r) and np.array_equal(off_a, off_r)
        assert np.array_equal(audio_a, audio_r), float(np.abs(audio_a - audio_r).max())
    finally:
        ref_engine.close()


@pytest.mark.gpu
def test_epilogue_statistics_match_the_sweep_kernel(engine, monkeypatch):
    """Instance-norm statistics from the conv epilogues (fp32 partials per row and 32 columns, summed in fp64;
    conv_mfma.hip ConvP::stat + adain_finalize_kernel) against the all-fp64 sweep of the stored tensor
    (adain_stats_kernel): a second engine built with K82_FUSED_STATS=0 sweeps everywhere.  Durations do not involve
    AdaIN and must be identical; the waveform may differ by the rounding of rstd only -- two orders of magnitude inside
    the 1e-3 waveform bar (istftnet.AdaIN1d -> InstanceNorm1d is what both compute).  Ragged lengths put the last
    partial of every row on a span that is cut by the sentence length."""
    from tts_amd.engine import Engine
    lengths = [37, 90, 12, 254, 65, 33]
    id_lists, ref_s = H.make_batch(7, lengths)
    ids, lens = engine.pack(id_lists)
    pred_a, nfr_a = engine.durations(ids, lens, ref_s)
    noise = _noise_for(nfr_a)
    audio_a, off_a = engine.decode(noise=noise)
    audio_a = audio_a.copy()
    monkeypatch.setenv("K82_FUSED_STATS", "0")
    ref_engine = Engine(spec.synthetic_state_dicts(0))
    try:
        pred_r, nfr_r = ref_engine.durations(ids, lens, ref_s)
        audio_r, off_r = ref_engine.decode(noise=noise)
        assert np.array_equal(pred_a, pred_r) and np.array_equal(off_a, off_r)
        err = float(np.abs(audio_a - audio_r).max())
        assert err <= 1e-4, err
    finally:
        ref_engine.close()


@pytest.mark.gpu
def test_bf16_ring_pipeline_gives_the_generic_loops_bits(monkeypatch):
    """bf16 mode: the ring pipeline of conv_mfma.hip (compile-time tap count, 4 x 1 waves of 32 x 128, one weight slot
    per tap requested a ci-tile ahead, staging threaded between the MFMAs) sums every output element in the generic
    loop's order, so an engine built with K82_RING=0 (generic loop everywhere) must reproduce durations and audio
    exactly on a ragged batch that crosses every tap count (3, 7, 11), both channel widths and all three epilogues."""
    from tts_amd.engine import Engine
    # (t1 stays fp32 in HBM on both sides: bf16 activations in HBM exist in the ring kernels only and round once more --
    # test_bf16_activations_in_hbm covers them)
    monkeypatch.setenv("K82_T1_BF16", "0")
    # the long batch runs the generator stages on the 4 x 2-wave ring (128 x 256 tiles: launches with at least two rounds
    # of them), the short one keeps every launch on the 4 x 1-wave ring
    for lengths, wide in (([37, 90, 12, 200, 65], True), ([9, 14], False)):
        id_lists, ref_s = H.make_batch(13, lengths)
        seeds = np.arange(len(lengths), dtype=np.uint64) + 70
        out = []
        for ring in ("1", "0"):
            monkeypatch.setenv("K82_RING", ring)
            eng = Engine(spec.synthetic_state_dicts(0), precision="bf16")
            try:
                n0 = [eng.prof_read(k)[1] for k in ("conv_shape_ring", "conv_shape_ring_wn2")]
                ids, lens = eng.pack(id_lists)
                with eng.session():
                    pred, _ = eng.durations(ids, lens, ref_s)
                    audio, off = eng.decode(seeds=seeds)
                out.append((pred, audio.copy(), off))
                ran = [eng.prof_read(k)[1] - a for k, a in zip(("conv_shape_ring", "conv_shape_ring_wn2"), n0)]
                # the comparison is only worth something if the "1" run really took the ring kernels (and, for the long
                # batch, the 4 x 2-wave shape) and the "0" run did not
                if ring == "1":
                    assert ran[0] > 0 and (ran[1] > 0) == wide, (lengths, ran)
                else:
                    assert ran == [0, 0], (lengths, ran)
            finally:
                eng.close()
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
        assert np.array_equal(out[0][1], out[1][1]), float(np.abs(out[0][1] - out[1][1]).max())


@pytest.mark.gpu
def test_bf16_activations_in_hbm(monkeypatch):
    """bf16 mode (north_star: "bf16 ... LDS-staged tiles"): the resblock-internal tensor t1 of every AdaINResBlock1 crosses
    HBM as bf16 channel pairs in the staged tile's own operand order (ConvP::y_b16 / x_b16; conv_mfma.hip epilogue_b16, XB
    staging) -- half the store and the load of that tensor.  Against an engine that keeps t1 in fp32 (K82_T1_BF16=0):
    durations and offsets identical (the duration branch never leaves fp32), the audio differs (t1 is rounded once more)
    by far less than the mode's own distance from fp32, and the north-star bar (SNR >= 40 dB against the fp32 mode)
    holds with margin on a ragged batch that takes all three ring shapes and all three tap counts."""
    from tts_amd.engine import Engine
    lengths = [37, 90, 12, 200, 65, 180, 150, 120]
    id_lists, ref_s = H.make_batch(29, lengths)
    seeds = np.arange(len(lengths), dtype=np.uint64) + 300
    out = {}
    for tag, env, prec in (("b16", "1", "bf16"), ("f32t1", "0", "bf16"), ("fp32", "1", "fp32")):
        monkeypatch.setenv("K82_T1_BF16", env)
        eng = Engine(spec.synthetic_state_dicts(0), precision=prec)
        try:
            ids, lens = eng.pack(id_lists)
            with eng.session():
                pred, _ = eng.durations(ids, lens, ref_s)
                audio, off = eng.decode(seeds=seeds)
            out[tag] = (pred, audio.copy(), off)
        finally:
            eng.close()
    for tag in ("f32t1", "fp32"):
        assert np.array_equal(out["b16"][0], out[tag][0]) and np.array_equal(out["b16"][2], out[tag][2])
    assert not np.array_equal(out["b16"][1], out["f32t1"][1])          # the bf16 tensor really was in the path
    off = out["b16"][2]

    def snr(ref, x):
        ref, x = ref.astype(np.float64), x.astype(np.float64)
        return 10 * np.log10((ref ** 2).sum() / max(((ref - x) ** 2).sum(), 1e-300))

    for b in range(len(lengths)):
        sl = slice(off[b], off[b + 1])
        s_mode, s_old, s_t1 = snr(out["fp32"][1][sl], out["b16"][1][sl]), snr(out["fp32"][1][sl], out["f32t1"][1][sl]), snr(out["f32t1"][1][sl], out["b16"][1][sl])
        print(f"sentence {b}: bf16 mode vs fp32 {s_mode:.1f} dB (t1 in fp32: {s_old:.1f} dB); t1 bf16 vs t1 fp32 {s_t1:.1f} dB")
        assert s_mode >= 40.0 and s_mode >= s_old - 3.0, (b, s_mode, s_old)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
def test_256_row_tiles_give_the_128_row_tiles_bits(monkeypatch, precision):
    """Large launches with >= 256 output channels run 256 x 128 workgroup tiles (4 x 2 / 8 x 1 waves, one workgroup per
    CU, the staged x tile shared by twice the rows; conv_mfma.hip launch_conv / launch_bf16 / launch_ring).  Every
    output element is still summed in the same order, so an engine built with K82_BM256=0 must reproduce durations
    and audio exactly.  The batch is long enough for the 256-channel generator stage to take the large tiles."""
    from tts_amd.engine import Engine
    lengths = [37, 90, 12, 200, 65, 180, 150, 120]      # 8 sentences: the 256-channel stage has >= 4 x CUs tiles with margin
    id_lists, ref_s = H.make_batch(17, lengths)
    seeds = np.arange(len(lengths), dtype=np.uint64) + 11
    out = []
    for sw in ("1", "0"):
        monkeypatch.setenv("K82_BM256", sw)
        eng = Engine(spec.synthetic_state_dicts(0), precision=precision)
        try:
            n0 = eng.prof_read("conv_shape_bm256")[1]
            ids, lens = eng.pack(id_lists)
            with eng.session():
                pred, _ = eng.durations(ids, lens, ref_s)
                audio, off = eng.decode(seeds=seeds)
            out.append((pred, audio.copy(), off))
            ran = eng.prof_read("conv_shape_bm256")[1] - n0
            assert (ran > 0) == (sw == "1"), (precision, sw, ran)       # the large-tile family ran in one engine only
        finally:
            eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
    assert np.array_equal(out[0][1], out[1][1]), float(np.abs(out[0][1] - out[1][1]).max())


@pytest.mark.gpu
def test_split_k_small_launches_give_the_same_bits(monkeypatch):
    """A single sentence's Linears are chains of dependent global -> LDS -> MFMA steps over K on a few dozen workgroups
    of an otherwise idle chip (the first-audio path, speech7.py:124-133).  Small fp32 launches of the token-rate and
    prosody layers are therefore split over K blocks -- the SAME blocks the blocked summation (KSPLIT) folds at -- one
    workgroup per (tile, block), and a second kernel adds the block sums in block order and applies the epilogue
    (conv_mfma.hip EPI_PART, splitk_reduce_kernel).  An engine built with K82_SPLITK=0 must reproduce durations, the
    pre-rounding durations, F0 and audio bit for bit, for one short sentence, one long one, and a small batch."""
    from tts_amd.engine import Engine
    for lengths in ([25], [200], [7, 33]):
        id_lists, ref_s = H.make_batch(31, lengths)
        seeds = np.arange(len(lengths), dtype=np.uint64) + 5
        out = []
        for sw in ("1", "0"):
            monkeypatch.setenv("K82_SPLITK", sw)
            eng = Engine(spec.synthetic_state_dicts(0))
            try:
                n0 = eng.prof_read("conv_shape_splitk")[1]
                eng.debug_enable(True)
                ids, lens = eng.pack(id_lists)
                with eng.session():
                    pred, _ = eng.durations(ids, lens, ref_s)
                    raw = eng.debug_get("duration").copy()
                    audio, off = eng.decode(seeds=seeds)
                    f0 = eng.debug_get("F0").copy()
                out.append((pred, raw, f0, audio.copy(), off))
                ran = eng.prof_read("conv_shape_splitk")[1] - n0
                assert (ran > 0) == (sw == "1"), (lengths, sw, ran)
            finally:
                eng.close()
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b), lengths


@pytest.mark.gpu
def test_conv_step_counter_timeout_fails_the_call(monkeypatch):
    """The fp32 / bf16x3 conv kernels replace the per-ci-tile barrier by an LDS step counter that a wave polls with a
    bound (conv_mfma.hip, LDS3).  A wave whose bound runs out goes on (every wave must reach the end of the grid) and may
    read an unstaged tile -- so it must say so: the engine finds the device error word behind its stream synchronize and
    the call fails with K82_EHIP instead of returning samples.  K82_CONV_SPIN_LIMIT=0 gives every wave that does not
    find the counter at its target on the FIRST poll an exhausted bound; the default (65536 polls, ~10 ms) is never met."""
    from tts_amd.engine import Engine
    id_lists, ref_s = H.make_batch(23, [60, 41])
    monkeypatch.setenv("K82_CONV_SPIN_LIMIT", "0")
    eng = Engine(spec.synthetic_state_dicts(0))
    try:
        n0 = eng.prof_read("conv_shape_lds3")[1]
        with pytest.raises(RuntimeError, match="step counter"):
            for _ in range(4):                              # one pass is thousands of polls; four to be sure
                eng.synthesize(id_lists, ref_s, seeds=[1, 2])
        assert eng.prof_read("conv_shape_lds3")[1] > n0
    finally:
        eng.close()
    monkeypatch.delenv("K82_CONV_SPIN_LIMIT")
    eng = Engine(spec.synthetic_state_dicts(0))            # the default bound: same inputs, no error, finite audio
    try:
        audios, _ = eng.synthesize(id_lists, ref_s, seeds=[1, 2])
        assert all(np.isfinite(a).all() and a.size > 0 for a in audios)
    finally:
        eng.close()
