"""Voice / language tables of Kokoro v1.0 as the reference's pickers present them (SURVEY.md 8f row 4;
kq3_1_f.py:288-298 language -> prefix, :303-376 voices per prefix).  A voice name is
``<language letter><f|m>_<name>``; the tables are data about the published voice pack, not code."""
from __future__ import annotations

from .text import LANG_CODES, LANG_TO_CODE

_NAMES = {
    "a": "af_heart af_alloy af_aoede af_bella af_jessica af_kore af_nicole af_nova af_river af_sarah af_sky "
         "am_adam am_echo am_eric am_fenrir am_liam am_michael am_onyx am_puck am_santa",
    "b": "bf_alice bf_emma bf_isabella bf_lily bm_daniel bm_fable bm_george bm_lewis",
    "j": "jf_alpha jf_gongitsune jf_nezumi jf_tebukuro jm_kumo",
    "z": "zf_xiaobei zf_xiaoni zf_xiaoxiao zf_xiaoyi zm_yunjian zm_yunxi zm_yunxia zm_yunyang",
    "e": "ef_dora em_alex em_santa",
    "f": "ff_siwis",
    "h": "hf_alpha hf_beta hm_omega hm_psi",
    "i": "if_sara im_nicola",
    "p": "pf_dora pm_alex pm_santa",
}
VOICES_BY_PREFIX = {k: tuple(v.split()) for k, v in _NAMES.items()}
DEFAULT_VOICE = "af_sarah"                                  # speech7.py:184


def voices_for_language(lang: str):
    """``update_voices_for_language`` (kq3_1_f.py:283-379): unknown languages fall back to American English."""
    prefix = LANG_TO_CODE.get(lang, lang if lang in LANG_CODES else "a")
    return VOICES_BY_PREFIX.get(prefix, VOICES_BY_PREFIX["a"])


def language_of_voice(voice: str) -> str:
    """'bf_emma' -> 'en-gb' (the pipeline language a voice belongs to)."""
    return LANG_CODES.get(voice[:1], "en-us")


def gender_of_voice(voice: str) -> str:
    return {"f": "female", "m": "male"}.get(voice[1:2], "unknown")


def all_voices():
    return tuple(v for p in VOICES_BY_PREFIX.values() for v in p)
