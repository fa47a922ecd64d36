"""Kokoro-82M architecture description: config, vocabulary, parameter table, synthetic weights.

The reference (fastrizwaan/tts) never holds the model arithmetic itself: it calls
``kokoro_onnx.Kokoro(model, voices).create(...)`` (speech7.py:82,127; basic.tts_kokoro.py:17,30)
which runs ``kokoro-v1.0.onnx`` -- an export of hexgrad/kokoro's ``KModel``.
Everything marked [EXT] below restates the public hexgrad/Kokoro-82M ``config.json`` and the
state-dict layout of ``kokoro-v1_0.pth`` (5 sub-state-dicts, old-style ``weight_g``/``weight_v``
weight-norm pairs, ``module.`` key prefix).  The loader (weights.py) asserts every shape in
this table, so a mismatch against a real checkpoint fails loudly instead of mis-computing.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

SAMPLE_RATE = 24000          # basic.tts_kokoro_2_play_pause_works.py:17,30
FRAME_SAMPLES = 600          # 1 predicted frame = 600 samples = 25 ms  [EXT]
STYLE_DIM_TOTAL = 256        # ref_s = [s_dec(128) | s_pros(128)]          [EXT]
MAX_PHONEMES = 510           # context 512 = 510 + 2 pad tokens            [EXT]

# [EXT] hexgrad/Kokoro-82M config.json
DEFAULT_CONFIG = {
    "istftnet": {
        "upsample_kernel_sizes": [20, 12],
        "upsample_rates": [10, 6],
        "gen_istft_hop_size": 5,
        "gen_istft_n_fft": 20,
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        "resblock_kernel_sizes": [3, 7, 11],
        "upsample_initial_channel": 512,
    },
    "dim_in": 64,
    "dropout": 0.2,
    "hidden_dim": 512,
    "max_conv_dim": 512,
    "max_dur": 50,
    "multispeaker": True,
    "n_layer": 3,
    "n_mels": 80,
    "n_token": 178,
    "style_dim": 128,
    "text_encoder_kernel_size": 5,
    "plbert": {
        "hidden_size": 768,
        "num_attention_heads": 12,
        "intermediate_size": 2048,
        "max_position_embeddings": 512,
        "num_hidden_layers": 12,
        "dropout": 0.1,
    },
}

# [EXT] phoneme -> id table shipped in Kokoro-82M config.json ("vocab").  A real config.json,
# when supplied, overrides this table (weights.load_config).
DEFAULT_VOCAB = {
    ";": 1, ":": 2, ",": 3, ".": 4, "!": 5, "?": 6, "—": 9, "…": 10, "\"": 11, "(": 12,
    ")": 13, "“": 14, "”": 15, " ": 16, "̃": 17, "ʣ": 18, "ʥ": 19,
    "ʦ": 20, "ʨ": 21, "ᵝ": 22, "ꭧ": 23, "A": 24, "I": 25, "O": 31, "Q": 33,
    "S": 35, "T": 36, "W": 39, "Y": 41, "ᵊ": 42, "a": 43, "b": 44, "c": 45, "d": 46, "e": 47,
    "f": 48, "h": 50, "i": 51, "j": 52, "k": 53, "l": 54, "m": 55, "n": 56, "o": 57, "p": 58,
    "q": 59, "r": 60, "s": 61, "t": 62, "u": 63, "v": 64, "w": 65, "x": 66, "y": 67, "z": 68,
    "ɑ": 69, "ɐ": 70, "ɒ": 71, "æ": 72, "β": 75, "ɔ": 76,
    "ɕ": 77, "ç": 78, "ɖ": 80, "ð": 81, "ʤ": 82, "ə": 83,
    "ɚ": 85, "ɛ": 86, "ɜ": 87, "ɟ": 90, "ɡ": 92, "ɥ": 99,
    "ɨ": 101, "ɪ": 102, "ʝ": 103, "ɯ": 110, "ɰ": 111, "ŋ": 112,
    "ɳ": 113, "ɲ": 114, "ɴ": 115, "ø": 116, "ɸ": 118, "θ": 119,
    "œ": 120, "ɹ": 123, "ɾ": 125, "ɻ": 126, "ʁ": 128, "ɽ": 129,
    "ʂ": 130, "ʃ": 131, "ʈ": 132, "ʧ": 133, "ʊ": 135, "ʋ": 136,
    "ʌ": 138, "ɣ": 139, "ɤ": 140, "χ": 142, "ʎ": 143, "ʒ": 147,
    "ʔ": 148, "ˈ": 156, "ˌ": 157, "ː": 158, "ʰ": 162, "ʲ": 164,
    "↓": 169, "→": 171, "↗": 172, "↘": 173, "ᵻ": 177,
}

ALBERT_EMBED = 128           # AlbertConfig default embedding_size [EXT]
SUBMODULES = ("bert", "bert_encoder", "predictor", "text_encoder", "decoder")


def _lstm(spec, prefix, n_in, hid):
    for sfx in ("", "_reverse"):
        spec[f"{prefix}.weight_ih_l0{sfx}"] = (4 * hid, n_in)
        spec[f"{prefix}.weight_hh_l0{sfx}"] = (4 * hid, hid)
        spec[f"{prefix}.bias_ih_l0{sfx}"] = (4 * hid,)
        spec[f"{prefix}.bias_hh_l0{sfx}"] = (4 * hid,)


def _wn_conv(spec, prefix, cout, cin, k, bias=True):
    """weight-normed Conv1d: weight_g [cout,1,1], weight_v [cout,cin,k]."""
    spec[f"{prefix}.weight_g"] = (cout, 1, 1)
    spec[f"{prefix}.weight_v"] = (cout, cin, k)
    if bias:
        spec[f"{prefix}.bias"] = (cout,)


def _wn_convT(spec, prefix, cin, cout_per_group, k, bias_n):
    """weight-normed ConvTranspose1d: weight [cin, cout/groups, k]; weight_norm dim=0."""
    spec[f"{prefix}.weight_g"] = (cin, 1, 1)
    spec[f"{prefix}.weight_v"] = (cin, cout_per_group, k)
    spec[f"{prefix}.bias"] = (bias_n,)


def _adain(spec, prefix, style, c):
    spec[f"{prefix}.fc.weight"] = (2 * c, style)
    spec[f"{prefix}.fc.bias"] = (2 * c,)


def _adain_resblk1d(spec, prefix, cin, cout, style, upsample=False):
    """istftnet.AdainResBlk1d [EXT]."""
    _wn_conv(spec, f"{prefix}.conv1", cout, cin, 3)
    _wn_conv(spec, f"{prefix}.conv2", cout, cout, 3)
    _adain(spec, f"{prefix}.norm1", style, cin)
    _adain(spec, f"{prefix}.norm2", style, cout)
    if cin != cout:
        _wn_conv(spec, f"{prefix}.conv1x1", cout, cin, 1, bias=False)
    if upsample:
        _wn_convT(spec, f"{prefix}.pool", cin, 1, 3, cin)


def _adain_resblock1(spec, prefix, c, k, style):
    """istftnet.AdaINResBlock1 [EXT]: 3 x (adain, snake, dilated conv, adain, snake, conv)."""
    for i in range(3):
        _wn_conv(spec, f"{prefix}.convs1.{i}", c, c, k)
        _wn_conv(spec, f"{prefix}.convs2.{i}", c, c, k)
        _adain(spec, f"{prefix}.adain1.{i}", style, c)
        _adain(spec, f"{prefix}.adain2.{i}", style, c)
        spec[f"{prefix}.alpha1.{i}"] = (1, c, 1)
        spec[f"{prefix}.alpha2.{i}"] = (1, c, 1)


def param_spec(config=None):
    """name -> shape for every tensor of the 5 sub-state-dicts (keys WITHOUT ``module.``)."""
    cfg = config or DEFAULT_CONFIG
    H = cfg["hidden_dim"]
    S = cfg["style_dim"]
    pl = cfg["plbert"]
    ist = cfg["istftnet"]
    out = OrderedDict()

    # ---- bert: transformers.AlbertModel (CustomAlbert) [EXT]
    b = OrderedDict()
    E, D, I = ALBERT_EMBED, pl["hidden_size"], pl["intermediate_size"]
    b["embeddings.word_embeddings.weight"] = (cfg["n_token"], E)
    b["embeddings.position_embeddings.weight"] = (pl["max_position_embeddings"], E)
    b["embeddings.token_type_embeddings.weight"] = (2, E)
    b["embeddings.LayerNorm.weight"] = (E,)
    b["embeddings.LayerNorm.bias"] = (E,)
    b["encoder.embedding_hidden_mapping_in.weight"] = (D, E)
    b["encoder.embedding_hidden_mapping_in.bias"] = (D,)
    L = "encoder.albert_layer_groups.0.albert_layers.0"
    b[f"{L}.full_layer_layer_norm.weight"] = (D,)
    b[f"{L}.full_layer_layer_norm.bias"] = (D,)
    for n in ("query", "key", "value", "dense"):
        b[f"{L}.attention.{n}.weight"] = (D, D)
        b[f"{L}.attention.{n}.bias"] = (D,)
    b[f"{L}.attention.LayerNorm.weight"] = (D,)
    b[f"{L}.attention.LayerNorm.bias"] = (D,)
    b[f"{L}.ffn.weight"] = (I, D)
    b[f"{L}.ffn.bias"] = (I,)
    b[f"{L}.ffn_output.weight"] = (D, I)
    b[f"{L}.ffn_output.bias"] = (D,)
    b["pooler.weight"] = (D, D)          # present in the checkpoint, unused by the forward
    b["pooler.bias"] = (D,)
    out["bert"] = b

    out["bert_encoder"] = OrderedDict(weight=(H, D), bias=(H,))

    # ---- predictor: ProsodyPredictor [EXT]
    p = OrderedDict()
    for i in range(cfg["n_layer"]):
        _lstm(p, f"text_encoder.lstms.{2 * i}", H + S, H // 2)
        p[f"text_encoder.lstms.{2 * i + 1}.fc.weight"] = (2 * H, S)
        p[f"text_encoder.lstms.{2 * i + 1}.fc.bias"] = (2 * H,)
    _lstm(p, "lstm", H + S, H // 2)
    p["duration_proj.linear_layer.weight"] = (cfg["max_dur"], H)
    p["duration_proj.linear_layer.bias"] = (cfg["max_dur"],)
    _lstm(p, "shared", H + S, H // 2)
    for br in ("F0", "N"):
        _adain_resblk1d(p, f"{br}.0", H, H, S)
        _adain_resblk1d(p, f"{br}.1", H, H // 2, S, upsample=True)
        _adain_resblk1d(p, f"{br}.2", H // 2, H // 2, S)
    for br in ("F0_proj", "N_proj"):
        p[f"{br}.weight"] = (1, H // 2, 1)
        p[f"{br}.bias"] = (1,)
    out["predictor"] = p

    # ---- text_encoder: TextEncoder [EXT]
    t = OrderedDict()
    t["embedding.weight"] = (cfg["n_token"], H)
    k = cfg["text_encoder_kernel_size"]
    for i in range(cfg["n_layer"]):
        _wn_conv(t, f"cnn.{i}.0", H, H, k)
        t[f"cnn.{i}.1.gamma"] = (H,)
        t[f"cnn.{i}.1.beta"] = (H,)
    _lstm(t, "lstm", H, H // 2)
    out["text_encoder"] = t

    # ---- decoder: istftnet.Decoder [EXT]
    d = OrderedDict()
    _adain_resblk1d(d, "encode", H + 2, 1024, S)
    for i in range(3):
        _adain_resblk1d(d, f"decode.{i}", 1024 + 2 + 64, 1024, S)
    _adain_resblk1d(d, "decode.3", 1024 + 2 + 64, 512, S, upsample=True)
    _wn_conv(d, "F0_conv", 1, 1, 3)
    _wn_conv(d, "N_conv", 1, 1, 3)
    _wn_conv(d, "asr_res.0", 64, 512, 1)
    g = "generator"
    d[f"{g}.m_source.l_linear.weight"] = (1, 9)
    d[f"{g}.m_source.l_linear.bias"] = (1,)
    c0 = ist["upsample_initial_channel"]
    rates, ks = ist["upsample_rates"], ist["upsample_kernel_sizes"]
    nfft = ist["gen_istft_n_fft"]
    for i, (u, kk) in enumerate(zip(rates, ks)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        _wn_convT(d, f"{g}.ups.{i}", cin, cout, kk, cout)
        if i + 1 < len(rates):
            sf0 = int(np.prod(rates[i + 1:]))
            d[f"{g}.noise_convs.{i}.weight"] = (cout, nfft + 2, sf0 * 2)
            d[f"{g}.noise_convs.{i}.bias"] = (cout,)
            _adain_resblock1(d, f"{g}.noise_res.{i}", cout, 7, S)
        else:
            d[f"{g}.noise_convs.{i}.weight"] = (cout, nfft + 2, 1)
            d[f"{g}.noise_convs.{i}.bias"] = (cout,)
            _adain_resblock1(d, f"{g}.noise_res.{i}", cout, 11, S)
        for j, rk in enumerate(ist["resblock_kernel_sizes"]):
            _adain_resblock1(d, f"{g}.resblocks.{i * 3 + j}", cout, rk, S)
    _wn_conv(d, f"{g}.conv_post", nfft + 2, c0 // (2 ** len(rates)), 7)
    out["decoder"] = d
    return out


def n_params(config=None):
    return sum(int(np.prod(s)) for sub in param_spec(config).values() for s in sub.values())


# --------------------------------------------------------------------------- synthetic weights
def synthetic_state_dicts(seed=0, config=None, dur_bias_frames=0.3, prefix="module.", lstm_gain=3.0, f0_gain=150.0,
                          f0_bias=120.0):
    """Seeded random weights with the exact checkpoint layout (dict of 5 state dicts, numpy f32).

    There is no network and no checkpoint in the container, so tests and the
    benchmark run on these.  Scales follow PyTorch's default initialisers so activations stay
    O(1) through the instance-norm stack.  Two deliberate departures keep the synthetic model
    in the regime real speech exercises:
      * duration_proj bias = logit(dur_bias_frames / max_dur) with the duration-branch LSTM
        weights x3 and duration_proj weights x16: durations of ~3-8 frames per token that vary
        from token to token (real Kokoro speech: ~3) instead of a constant sigmoid(0)*50 = 25;
      * F0_proj scaled so F0 ~ N(120, ~90) Hz: voiced (F0 > 10) and unvoiced frames both occur,
        so the harmonic source and its uv gate are exercised.
    ``lstm_gain=1, f0_gain=1, f0_bias=0`` gives the "calm" set of the free-running parity test: the duration-branch
    LSTMs at their default scale (no amplification of rounding noise) and F0 ~ N(0, 1) Hz, i.e. every frame unvoiced,
    so no F0 is integrated into a phase and two fp32 implementations stay sample-comparable end to end.
    """
    cfg = config or DEFAULT_CONFIG
    rng = np.random.default_rng(seed)
    spec = param_spec(cfg)
    out = OrderedDict()

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    for sub, table in spec.items():
        sd = OrderedDict()
        for name, shape in table.items():
            leaf = name.rsplit(".", 1)[-1]
            if sub == "bert":
                if "LayerNorm.weight" in name or "layer_norm.weight" in name:
                    v = np.ones(shape, np.float32) + uni(shape, 0.1)
                elif "LayerNorm.bias" in name or "layer_norm.bias" in name:
                    v = uni(shape, 0.1)
                elif leaf == "bias":
                    v = uni(shape, 0.05)
                elif "embeddings" in name:
                    v = (rng.standard_normal(shape) * 0.5).astype(np.float32)
                else:
                    # wider than ALBERT's N(0, 0.02) so attention is not near-uniform
                    v = (rng.standard_normal(shape) * (1.0 / math.sqrt(shape[1]))).astype(np.float32)
            elif leaf == "weight_g":
                v = None  # filled from weight_v below
            elif leaf == "weight_v":
                fan_in = int(np.prod(shape[1:]))
                v = uni(shape, 1.0 / math.sqrt(fan_in))
            elif leaf.startswith("alpha"):
                v = (np.ones(shape, np.float32) + uni(shape, 0.5))
            elif leaf in ("gamma",):
                v = np.ones(shape, np.float32) + uni(shape, 0.1)
            elif leaf in ("beta",):
                v = uni(shape, 0.1)
            elif "weight_ih" in leaf or "weight_hh" in leaf or "bias_ih" in leaf or "bias_hh" in leaf:
                hid = table[name.rsplit(".", 1)[0] + ".weight_hh_l0"][1]
                v = uni(shape, 1.0 / math.sqrt(hid))
            elif name == "embedding.weight":
                v = rng.standard_normal(shape).astype(np.float32)
            elif leaf == "bias":
                # fan_in of the matching weight
                wname = name[:-4] + "weight"
                wshape = table.get(wname) or table.get(name[:-4] + "weight_v")
                fan_in = int(np.prod(wshape[1:]))
                v = uni(shape, 1.0 / math.sqrt(fan_in))
            else:  # plain weight (Linear / Conv1d)
                fan_in = int(np.prod(shape[1:]))
                v = uni(shape, 1.0 / math.sqrt(fan_in))
            sd[name] = v
        # weight_g = ||weight_v|| over all dims but 0 (torch weight_norm init), jittered
        for name in list(sd):
            if name.endswith("weight_g"):
                wv = sd[name[:-1] + "v"]
                nrm = np.sqrt((wv.astype(np.float64) ** 2).sum(axis=tuple(range(1, wv.ndim)), keepdims=True))
                jit = 1.0 + rng.uniform(-0.2, 0.2, size=nrm.shape)
                sd[name] = (nrm * jit).astype(np.float32).reshape(table[name])
        out[sub] = sd

    p = out["predictor"]
    q = dur_bias_frames / cfg["max_dur"]
    p["duration_proj.linear_layer.bias"] = (
        np.full((cfg["max_dur"],), math.log(q / (1 - q)), np.float32)
        + rng.uniform(-0.5, 0.5, size=(cfg["max_dur"],)).astype(np.float32))
    p["duration_proj.linear_layer.weight"] = (p["duration_proj.linear_layer.weight"] * 16.0).astype(np.float32)
    for k in list(p):
        if (k.startswith("lstm.weight") or k.startswith("text_encoder.lstms")) and "weight_" in k:
            p[k] = (p[k] * lstm_gain).astype(np.float32)
    p["F0_proj.weight"] = (p["F0_proj.weight"] * f0_gain).astype(np.float32)
    p["F0_proj.bias"] = np.full((1,), f0_bias, np.float32)
    if prefix:
        out = OrderedDict((sub, OrderedDict((prefix + k, v) for k, v in sd.items())) for sub, sd in out.items())
    return out


def synthetic_voice_pack(seed=1, n_rows=MAX_PHONEMES, scale=0.15):
    """A voice pack with the shape of ``voices/<name>.pt``: f32[510,1,256] (kd.py:44-46)."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((1, 1, STYLE_DIM_TOTAL)) * scale
    drift = rng.standard_normal((n_rows, 1, STYLE_DIM_TOTAL)) * (scale * 0.1)
    return (base + drift).astype(np.float32)
