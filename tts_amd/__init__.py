"""tts_amd -- MI355X-native Kokoro-82M synthesis path of fastrizwaan/tts.

Front doors: ``Kokoro`` (the reference's real boundary, speech7.py:82,127), ``KPipeline`` /
``KModel`` (north_star API).  Everything computes in ``libk82.so`` (tts_amd/csrc, C-ABI in
include/k82.h); nothing is initialised at import time.
"""
__all__ = ["Kokoro", "KPipeline", "KModel", "Engine", "TTSEngine", "EpubBook", "SAMPLE_RATE"]
SAMPLE_RATE = 24000


def __getattr__(name):
    if name in ("Kokoro", "KPipeline", "KModel"):
        from . import kokoro
        return getattr(kokoro, name)
    if name == "TTSEngine":
        from .readaloud import TTSEngine
        return TTSEngine
    if name == "EpubBook":
        from .epub import EpubBook
        return EpubBook
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)
