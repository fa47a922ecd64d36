"""Checkpoint / voice-pack loading for the hexgrad Kokoro-82M layouts.

* ``kokoro-v1_0.pth``: dict of 5 state dicts (bert, bert_encoder, predictor, text_encoder,
  decoder), keys prefixed ``module.``, weight-normed convs stored as ``weight_g``/``weight_v``
  (or the newer ``parametrizations.weight.original0/1``).  Loaded with
  ``torch.load(weights_only=True)`` only.
* ``config.json``: architecture + 178-entry phoneme vocab.
* voices: ``voices/<name>.pt`` f32[510,1,256] or ``voices-v1.0.bin`` (npz of name -> f32[510,1,256],
  the file the reference passes to ``Kokoro(model, voices)``: speech7.py:73-82, kd.py:44-46).

The reference's own model file ``kokoro-v1.0.onnx`` cannot be parsed here (no onnx library in
the image); ``find_model_files`` therefore looks for the ``.pth`` beside it.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import numpy as np

from . import spec

_IGNORED_SUFFIXES = ("position_ids",)


def _strip(k: str) -> str:
    return k[7:] if k.startswith("module.") else k


def fold_state_dicts(sds, config=None) -> "OrderedDict[str, np.ndarray]":
    """5 raw state dicts -> flat {"<sub>.<key>": f32 array} with weight-norm folded.

    w = g * v / ||v||, the norm taken over every dim but 0 (torch ``weight_norm(dim=0)``, also
    for ConvTranspose1d whose dim 0 is in_channels).  Shapes are asserted against
    ``spec.param_spec`` so an unexpected checkpoint fails here, loudly.
    """
    table = spec.param_spec(config)
    out = OrderedDict()
    for sub in spec.SUBMODULES:
        if sub not in sds:
            raise KeyError(f"checkpoint has no '{sub}' state dict")
        raw = OrderedDict()
        for k, v in sds[sub].items():
            k = _strip(k)
            if k.endswith(_IGNORED_SUFFIXES):
                continue
            k = k.replace("parametrizations.weight.original0", "weight_g").replace(
                "parametrizations.weight.original1", "weight_v")
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            raw[k] = np.ascontiguousarray(a, dtype=np.float32)
        want = table[sub]
        for k, shape in want.items():
            if k not in raw:
                raise KeyError(f"checkpoint is missing {sub}.{k}")
            if tuple(raw[k].shape) != tuple(shape):
                raise ValueError(f"{sub}.{k}: shape {tuple(raw[k].shape)} != expected {tuple(shape)}")
        extra = [k for k in raw if k not in want]
        if extra:
            raise ValueError(f"unexpected tensors in '{sub}': {extra[:5]}")
        for k in want:
            if k.endswith("weight_g"):
                base = k[: -len("weight_g")]
                g = raw[k].astype(np.float64)
                v = raw[base + "weight_v"].astype(np.float64)
                nrm = np.sqrt((v * v).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
                out[f"{sub}.{base}weight"] = (v * (g / nrm)).astype(np.float32)
            elif k.endswith("weight_v"):
                continue
            else:
                out[f"{sub}.{k}"] = raw[k]
    return out


def load_pth(path):
    import torch
    sds = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sds, dict) or not all(s in sds for s in spec.SUBMODULES):
        raise ValueError(f"{path}: not a Kokoro checkpoint (expected a dict with {spec.SUBMODULES})")
    return sds


def load_config(path):
    with open(path, "r", encoding="utf-8") as f:
        cfg = json.load(f)
    for key in ("n_token", "hidden_dim", "style_dim", "n_layer", "max_dur", "plbert", "istftnet"):
        if key not in cfg:
            raise ValueError(f"{path}: config.json lacks '{key}'")
    return cfg


def load_voices(path) -> "dict[str, np.ndarray]":
    """``voices-v1.0.bin`` (npz) / a ``voices`` directory of ``*.pt`` / a single ``.pt`` / ``.npy``."""
    voices = {}
    if os.path.isdir(path):
        import torch
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".pt"):
                voices[fn[:-3]] = np.asarray(torch.load(os.path.join(path, fn), map_location="cpu",
                                                        weights_only=True), dtype=np.float32)
    elif path.endswith(".pt"):
        import torch
        name = os.path.basename(path)[:-3]
        voices[name] = np.asarray(torch.load(path, map_location="cpu", weights_only=True), dtype=np.float32)
    else:
        with np.load(path, allow_pickle=False) as z:     # voices-v1.0.bin is an npz archive
            for name in z.files:
                voices[name] = np.asarray(z[name], dtype=np.float32)
    for name, v in voices.items():
        v = v.reshape(v.shape[0], -1)
        if v.shape[1] != spec.STYLE_DIM_TOTAL:
            raise ValueError(f"voice '{name}' has style width {v.shape[1]}, expected {spec.STYLE_DIM_TOTAL}")
        voices[name] = v.reshape(v.shape[0], 1, spec.STYLE_DIM_TOTAL)
    if not voices:
        raise ValueError(f"no voices found in {path}")
    return voices


def find_model_files(model_path=None, voices_path=None):
    """Search order of SURVEY.md 8b: explicit arg -> KOKORO_ONNX_PATH / KOKORO_VOICES_PATH
    (epubviewer_06_latest_1_nov_2025_theme.py:478-479) -> /app/share/kokoro-models ->
    ~/.local/share/kokoro-models (speech7.py:73-79).  Returns (pth, config_json, voices)."""
    cand_dirs = []
    if model_path:
        cand_dirs.append(model_path if os.path.isdir(model_path) else os.path.dirname(os.path.abspath(model_path)))
    for env in ("KOKORO_ONNX_PATH", "KOKORO_VOICES_PATH"):
        if os.environ.get(env):
            p = os.environ[env]
            cand_dirs.append(p if os.path.isdir(p) else os.path.dirname(os.path.abspath(p)))
    cand_dirs += ["/app/share/kokoro-models", os.path.expanduser("~/.local/share/kokoro-models")]
    pth = cfg = None
    if model_path and model_path.endswith(".pth") and os.path.exists(model_path):
        pth = model_path
    for d in cand_dirs:
        if pth is None:
            for fn in ("kokoro-v1_0.pth", "kokoro-v1.0.pth", "model.pth"):
                if os.path.exists(os.path.join(d, fn)):
                    pth = os.path.join(d, fn)
                    break
        if cfg is None and os.path.exists(os.path.join(d, "config.json")):
            cfg = os.path.join(d, "config.json")
    vo = voices_path or os.environ.get("KOKORO_VOICES_PATH")
    if vo is None or not os.path.exists(vo):
        for d in cand_dirs:
            for fn in ("voices-v1.0.bin", "voices"):
                if os.path.exists(os.path.join(d, fn)):
                    vo = os.path.join(d, fn)
                    break
            if vo and os.path.exists(vo):
                break
    return pth, cfg, vo
