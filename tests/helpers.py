"""Shared parity helpers: run one batch through the HIP engine and the CPU oracle."""
import numpy as np
import torch

from tts_amd import spec
from oracle import kokoro_ref as R

# oracle tap name -> (engine tap name, valid-length rule as (mul, add) of frames 'F' or tokens 'T')
STAGES = [
    ("bert_out", "T", 1, 0), ("d_en", "T", 1, 0), ("d", "T", 1, 0), ("dur_lstm", "T", 1, 0),
    ("en", "F", 1, 0), ("shared", "F", 1, 0), ("F0", "F", 2, 0), ("N", "F", 2, 0), ("t_en", "T", 1, 0),
    ("asr", "F", 1, 0), ("dec_encode", "F", 1, 0), ("dec_out", "F", 2, 0), ("har_source", "F", 600, 0),
    ("har", "F", 120, 1), ("gen0_noise_conv", "F", 20, 0), ("gen0_xsource", "F", 20, 0), ("gen0_out", "F", 20, 0),
    ("gen1_xsource", "F", 120, 1), ("gen1_out", "F", 120, 1), ("conv_post", "F", 120, 1),
]


def make_batch(seed, lengths, voice_seed=1):
    rng = np.random.default_rng(seed)
    id_lists = [rng.integers(1, 178, size=n).tolist() for n in lengths]
    pack = spec.synthetic_voice_pack(voice_seed)
    ref_s = np.stack([pack[len(x) - 1, 0] for x in id_lists]).astype(np.float32)   # KPipeline: pack[len(ps)-1]
    return id_lists, ref_s


def oracle_run(model, ids_list, ref_s_row, speed, noise, forced=None, taps=None, f0=None, n=None, har=None):
    ids = torch.tensor([[0] + list(ids_list) + [0]], dtype=torch.long)
    rs = torch.from_numpy(ref_s_row[None].copy())
    nz = None if noise is None else torch.from_numpy(noise[None].copy())
    audio, pd = model.forward_with_tokens(ids, rs, speed, noise=nz, pred_dur=forced, taps=taps,
                                          f0_override=f0, n_override=n, har_override=har)
    return audio.numpy(), pd.numpy()


def to_bct(name, t):
    """oracle tap -> [C, L] numpy (channel-first)."""
    a = t.detach().numpy()
    if name in ("bert_out", "d", "dur_lstm"):
        return a[0].T                      # [1, T, C]
    if name in ("F0", "N", "har_source"):
        return a.reshape(1, -1)
    return a[0]


def compare_har(got, ref, strong_thresh=1e-2):
    """[22, L] magnitude|phase tensors -> (max mag err, max strong-bin phase err mod 2pi, flips).
    A flip = a phase element differing by ~2*pi (branch cut of angle() crossed by rounding noise)."""
    mag_err = float(np.abs(got[:11] - ref[:11]).max())
    d = np.abs(got[11:] - ref[11:])
    dw = np.minimum(d, 2 * np.pi - d)
    strong = ref[:11] > strong_thresh
    flips = int((d > np.pi).sum())
    # weak bins: phase error scales like (noise / magnitude); bound the error of the complex value instead
    cerr = float(np.abs(got[:11] * np.exp(1j * got[11:]) - ref[:11] * np.exp(1j * ref[11:])).max())
    return mag_err, float(dw[strong].max()) if strong.any() else 0.0, flips, cerr
