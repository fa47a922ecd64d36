"""GPU tests against the committed golden vectors tests/golden/synthetic_T{8,64,256,512}.npz (tools/make_golden.py).

Nothing here runs the oracle: the numbers were produced on CPU and committed.  Each vector holds the fp32 oracle's
results AND the same model evaluated in float64, so where the model itself is ill-conditioned (durations just before
rounding, the F0 curve the harmonic source integrates, the free-running waveform) the test can ask the only
meaningful question -- is the engine as close to the fp64 value as the fp32 oracle is? -- instead of asking two fp32
implementations to agree on noise.  The oracle is unpinned by the reference: this is engine ==
restatement, measured properly.
"""
import os

import numpy as np
import pytest

from tts_amd import spec
from oracle import kokoro_ref as R          # counter_noise only (the numpy twin of the in-kernel generator)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
AUDIO_TOL = 1e-3


def _load(T):
    return np.load(os.path.join(GOLD, f"synthetic_T{T}.npz"))


def _snr(ref, x):
    ref, x = ref.astype(np.float64), x.astype(np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((ref - x) ** 2).sum(), 1e-300))


def _run(engine, g, forced_curves):
    ids = g["ids"][None].astype(np.int64)
    T = ids.shape[1]
    lens = np.array([T], dtype=np.int32)
    ref_s = g["ref_s"][None]
    dec = g["decode_dur"].astype(np.int32)
    use_forced_dur = not np.array_equal(dec, g["pred_dur"])
    F = int(dec.sum())
    noise = R.counter_noise(int(g["noise_seed"]), F * 600)
    out = {}
    engine.debug_enable(True)
    try:
        with engine.session():
            pred, nfr = engine.durations(ids, lens, ref_s)                       # the model's own durations first
            out["pred"], out["raw"] = pred[0].copy(), engine.debug_get("duration")[0, 0, :T].copy()
            out["stages_T"] = {k: engine.debug_get(k)[0] for k in ("bert_out", "d_en", "d", "dur_lstm")}
            if use_forced_dur:
                pred2, nfr = engine.durations(ids, lens, ref_s, forced_dur=dec[None])
                assert np.array_equal(pred2[0], dec)
            assert int(nfr[0]) == F
            if forced_curves:
                engine.debug_force("F0", g["F0_32"])
                engine.debug_force("N", g["N_32"])
                engine.debug_force("har", g["har32"])
            audio, off = engine.decode(noise=noise)
            out["audio"] = audio.copy()
            out["F0"], out["N"] = engine.debug_get("F0")[0, 0, :2 * F].copy(), engine.debug_get("N")[0, 0, :2 * F].copy()
            out["stages_F"] = {k: engine.debug_get(k)[0] for k in (
                "en", "shared", "t_en", "asr", "dec_encode", "dec_out", "gen0_noise_conv", "gen0_xsource", "gen0_out",
                "gen1_xsource", "gen1_out", "conv_post", "f0_blk0", "f0_blk1", "f0_blk2", "n_blk0", "n_blk1", "n_blk2")}
    finally:
        for k in ("F0", "N", "har"):
            engine.debug_force(k, None)
        engine.debug_enable(False)
    out["T"], out["F"] = T, F
    return out


def _check_durations(g, r, T):
    """Integer durations bit-exact; pre-rounding values no further from the fp64 arbiter than the fp32 oracle is
    (x2 + a floor of 1e-5: five fp32 ulps of a sum of 50 sigmoids ~ 25)."""
    assert np.array_equal(r["pred"], g["pred_dur"]), (r["pred"], g["pred_dur"])
    err_e = float(np.abs(r["raw"].astype(np.float64) - g["raw64"]).max())
    err_o = float(np.abs(g["raw32"].astype(np.float64) - g["raw64"]).max())
    margin = float(np.abs((g["raw64"] % 1.0) - 0.5).min())
    print(f"T{T}: raw duration |engine-fp64| {err_e:.2e}, |oracle32-fp64| {err_o:.2e}, min fp64 rounding margin {margin:.2e}")
    assert err_e <= 2.0 * err_o + 1e-5, (err_e, err_o)
    # every token whose fp64 value is further from a rounding boundary than the engine's own error must round alike
    far = np.abs((g["raw64"] % 1.0) - 0.5) > err_e
    want = np.maximum(np.rint(g["raw64"]), 1).astype(np.int64)
    assert np.array_equal(r["pred"][far], want[far])


def _lens_of(name, T, F):
    return {"bert_out": T, "d_en": T, "d": T, "dur_lstm": T, "t_en": T, "en": F, "shared": F, "asr": F, "dec_encode": F,
            "dec_out": 2 * F, "gen0_noise_conv": 20 * F, "gen0_xsource": 20 * F, "gen0_out": 20 * F,
            "gen1_xsource": 120 * F + 1, "gen1_out": 120 * F + 1, "conv_post": 120 * F + 1}[name]


ROWS = {"bert_out": 768, "d_en": 512, "d": 640, "dur_lstm": 512, "t_en": 512, "en": 640, "shared": 512, "asr": 512,
        "dec_encode": 1024, "dec_out": 512, "gen0_noise_conv": 256, "gen0_xsource": 256, "gen0_out": 256,
        "gen1_xsource": 128, "gen1_out": 128, "conv_post": 22}


def _check_checksums(g, r, names, tol):
    """Committed per-stage checksums of the fp32 oracle (sum, sum|x|, max|x|): the engine's tensors must reproduce
    sum|x| and max|x| to ``tol`` relative (plain sums cancel and are compared against sum|x|)."""
    T, F = r["T"], r["F"]
    for name in names:
        got = {**r["stages_T"], **r["stages_F"]}[name][:ROWS[name], :_lens_of(name, T, F)].astype(np.float64)
        s, sa, mx = g["sum_" + name]
        assert abs(np.abs(got).sum() - sa) <= tol * sa, (name, np.abs(got).sum(), sa)
        assert abs(got.sum() - s) <= tol * sa, (name, got.sum(), s)
        assert abs(np.abs(got).max() - mx) <= 10 * tol * max(mx, 1.0), (name, np.abs(got).max(), mx)


@pytest.mark.parametrize("T", [8, 64])
def test_golden_teacher_forced_waveform(engine, T):
    """Short vectors: durations bit-exact; with F0 / N / har pinned to the committed oracle tensors the engine's
    waveform meets the north-star bar (max-abs <= 1e-3) against the committed oracle waveform, sample by sample; every
    stage checksum downstream of the pinned tensors to 2e-4."""
    g = _load(T)
    r = _run(engine, g, forced_curves=True)
    _check_durations(g, r, T)
    assert r["audio"].size == int(g["audio_len"])
    err = float(np.abs(r["audio"] - g["audio32"]).max())
    print(f"T{T}: teacher-forced waveform max-abs vs committed oracle audio: {err:.2e}")
    assert err <= AUDIO_TOL, err
    _check_checksums(g, r, ("bert_out", "d_en", "t_en", "en", "asr", "shared", "dec_encode", "dec_out", "gen0_noise_conv",
                            "gen0_xsource", "gen0_out", "gen1_xsource", "gen1_out", "conv_post"), 2e-4)
    _check_checksums(g, r, ("d", "dur_lstm"), 2e-3)


def test_golden_free_running_against_fp64_arbiter(engine):
    """No teacher forcing, the four committed vectors.  The engine's F0 / N curves must be as close to the float64
    evaluation of the model as the fp32 oracle's own curves are, and so must the waveform (SNR against fp64 no more than
    3 dB below the fp32 oracle's SNR against fp64): the free-running sample-wise gap between two fp32 implementations is
    the model's conditioning, not an engine error -- the fp32 oracle itself sits 0.15-0.77 max-abs from the fp64 waveform.

    The criterion on the curves is r = (|engine - fp64| + floor) / (|oracle32 - fp64| + floor), floor = 2e-6 of the curve's
    range (a few fp32 ulps; round 2 had widened it tenfold to pass -- it is back at its original value since the conv
    kernel sums K in blocks, conv_mfma.hip KSPLIT).  Per vector r is a lottery: two builds of the engine that differ only
    in which multiply-adds of the layer-norm kernel the compiler contracted -- a last-bit difference -- give r = 1.38 and
    2.56 for F0 at T = 8, 1.73 and 1.94 at T = 256 (profiles/r03_layernorm_rounding_lottery.txt; the BiLSTM stacks amplify
    any rounding difference by a factor that depends on its realisation).  So the bound x2 is asserted on the GEOMETRIC
    MEAN of r over the four vectors (measured 0.8-0.9), x4 on every single vector, and 5e-4 of the range absolutely."""
    logs = {"F0": [], "N": []}
    for T in (8, 64, 256, 512):
        g = _load(T)
        r = _run(engine, g, forced_curves=False)
        _check_durations(g, r, T)
        for name in ("F0", "N"):
            ref64 = g[name + "_64"]
            e_eng = float(np.abs(r[name].astype(np.float64) - ref64).max())
            e_orc = float(np.abs(g[name + "_32"].astype(np.float64) - ref64).max())
            scale = float(np.abs(ref64).max())
            ratio = (e_eng + 2e-6 * scale) / (e_orc + 2e-6 * scale)
            logs[name].append(np.log(ratio))
            print(f"T{T}: {name} |engine-fp64| {e_eng:.2e}, |oracle32-fp64| {e_orc:.2e} (max |{name}| {scale:.1f}), ratio {ratio:.2f}")
            assert ratio <= 4.0, (T, name, e_eng, e_orc)
            assert e_eng <= 5e-4 * max(scale, 1.0), (T, name, e_eng, scale)      # relative 5e-4 against the arbiter
        st = int(g["audio_stride"])
        a = r["audio"][::st]
        assert r["audio"].size == int(g["audio_len"]) and a.size == g["audio64"].size
        snr_e, snr_o = _snr(g["audio64"], a), _snr(g["audio64"], g["audio32"])
        mx_e, mx_o = float(np.abs(a - g["audio64"]).max()), float(np.abs(g["audio32"] - g["audio64"]).max())
        print(f"T{T}: free-running vs fp64: engine SNR {snr_e:.1f} dB (max-abs {mx_e:.3f}); fp32 oracle SNR {snr_o:.1f} dB (max-abs {mx_o:.3f})")
        assert snr_e >= snr_o - 3.0, (T, snr_e, snr_o)
        # token-rate stages are upstream of every ill-conditioned step: checksums hold free-running too
        _check_checksums(g, r, ("bert_out", "d_en", "t_en"), 2e-4)
    for name, v in logs.items():
        gm = float(np.exp(np.mean(v)))
        print(f"{name}: geometric mean of the ratio over the four vectors {gm:.2f}")
        assert gm <= 2.0, (name, gm, [float(np.exp(x)) for x in v])


def test_free_running_waveform_meets_the_bar_on_calm_weights():
    """A second synthetic weight set (spec.synthetic_state_dicts(lstm_gain=1, f0_gain=1, f0_bias=0)): duration-branch
    LSTMs at default scale and an F0 head that stays below the 10 Hz voicing threshold, so nothing is integrated into
    a phase.  There the engine must meet max-abs <= 1e-3 against the fp32 oracle FREE-RUNNING -- no forced curves --
    which shows the teacher forcing elsewhere compensates the model's conditioning, not an engine defect."""
    import warnings
    warnings.filterwarnings("ignore")
    import torch
    from tts_amd.engine import Engine
    sds = spec.synthetic_state_dicts(0, lstm_gain=1.0, f0_gain=1.0, f0_bias=0.0)
    model = R.build(spec.DEFAULT_CONFIG, sds)
    eng = Engine(sds)
    try:
        rng = np.random.default_rng(77)
        id_lists = [rng.integers(1, 178, size=n).tolist() for n in (9, 40, 23)]
        pack = spec.synthetic_voice_pack(1)
        ref_s = np.stack([pack[len(x) - 1, 0] for x in id_lists]).astype(np.float32)
        ids, lens = eng.pack(id_lists)
        eng.debug_enable(True)
        with eng.session():
            pred, nfr = eng.durations(ids, lens, ref_s)
            noise = np.concatenate([R.counter_noise(300 + b, int(nfr[b]) * 600) for b in range(3)])
            audio, off = eng.decode(noise=noise)
            f0 = eng.debug_get("F0")
        assert float(np.abs(f0).max()) < 10.0, "the calm set must be unvoiced everywhere"
        for b, x in enumerate(id_lists):
            t = torch.tensor([[0] + x + [0]])
            a_ref, pd = model.forward_with_tokens(t, torch.from_numpy(ref_s[b:b + 1].copy()), 1.0,
                                                  noise=torch.from_numpy(noise.reshape(-1, 9)[off[b]:off[b + 1]][None].copy()))
            assert np.array_equal(pd.numpy(), pred[b, :lens[b]])
            err = float(np.abs(audio[off[b]:off[b + 1]] - a_ref.numpy()).max())
            print(f"calm weights, sentence {b}: free-running max-abs {err:.2e}")
            assert err <= AUDIO_TOL, (b, err)
    finally:
        eng.close()


STAGE_NOISE = ("bert_out", "d_en", "d", "dur_lstm", "t_en", "shared", "f0_blk0", "f0_blk1", "f0_blk2", "n_blk0", "n_blk1", "n_blk2")
# stages upstream of every recurrence: what a conv / attention / layer-norm kernel rounds shows here directly
STAGE_FEEDFORWARD = ("bert_out", "d_en")


def test_golden_stage_noise_is_the_oracles(engine):
    """Stage by stage through the token-rate and prosody layers (free-running), on the four committed vectors: the
    engine's rms distance from the committed float64 evaluation, divided by the fp32 oracle's own rms distance from it
    (tests/golden/synthetic_T*_stages.npz, 2048 strided samples per stage).  A kernel that rounds more than torch's CPU
    kernel would show as a ratio that jumps at its stage -- round 2's build measured 2.1-3.1 from `bert_out` / `t_en` on
    (the conv kernel's single sequential K chain); with blocked summation (conv_mfma.hip KSPLIT) the feed-forward stages
    sit at 0.8-1.6 on every vector.  Bound there: 1.6 per vector.

    From `d` on every stage sits behind the three-BiLSTM stack, which amplifies the ~1e-7 noise of `d_en` 5x (T = 8) to
    100x (T = 512) -- by a factor that is a random variable of the noise REALISATION, not of its size: five seeds at
    T = 256 gave d ratios of 1.96, 5.91, 1.25, 0.44 and 0.99 with bert_out / d_en at 1.19-1.27 every time
    (profiles/r03_stage_noise_T256_5seeds.txt), and the committed vectors give 0.87, 0.78, 1.96, 0.31.  One vector's
    ratio says nothing there; the bound is on the GEOMETRIC MEAN over the four vectors (<= 1.6), and every vector's
    engine error must stay below 5e-4 of the stage's range (the stage tolerance of the parity tests)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    from make_golden import stage_index
    logs = {k: [] for k in STAGE_NOISE}
    for T in (8, 64, 256, 512):
        g = _load(T)
        st = np.load(os.path.join(GOLD, f"synthetic_T{T}_stages.npz"))
        r = _run(engine, g, forced_curves=False)
        got_all = {**r["stages_T"], **r["stages_F"]}
        print(f"T{T}: stage, rms |engine-fp64| / max|fp64|, rms |oracle32-fp64| / max|fp64|, ratio")
        for k in STAGE_NOISE:
            C, L = (int(v) for v in st[f"{k}_shape"])
            flat = np.ascontiguousarray(got_all[k][:C, :L]).reshape(-1)
            e = flat[stage_index(flat.size)].astype(np.float64)
            r64, r32 = st[f"{k}_64"], st[f"{k}_32"].astype(np.float64)
            assert e.shape == r64.shape == r32.shape
            sc = float(st[f"{k}_absmax"])
            re_, ro = float(np.sqrt(((e - r64) ** 2).mean())) / sc, float(np.sqrt(((r32 - r64) ** 2).mean())) / sc
            ratio = (re_ + 2e-8) / (ro + 2e-8)
            logs[k].append(np.log(ratio))
            print(f"  {k:9s} {re_:.3e} {ro:.3e} {ratio:5.2f}")
            assert re_ <= 5e-4, (T, k, re_)
            if k in STAGE_FEEDFORWARD:
                assert ratio <= 1.6, (T, k, re_, ro)
    print("geometric mean of the ratio over the four vectors:")
    for k in STAGE_NOISE:
        gm = float(np.exp(np.mean(logs[k])))
        print(f"  {k:9s} {gm:5.2f}")
        assert gm <= 1.6, (k, gm, [float(np.exp(v)) for v in logs[k]])
