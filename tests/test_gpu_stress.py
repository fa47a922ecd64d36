"""BASELINE.json configs[3] at FULL size under pytest: 128 sentences x 510 phonemes, durations forced to 4 frames per
token (F = 2048 frames = 51.2 s per sentence, 262 144 frames and 157 M samples per pass) -- the time-tiled long
generator / iSTFT path, the arena arithmetic and the grids at the largest batch the configs name.  The oracle cannot run
this size, so the checks are the size-independent properties of the domain:
offsets = 600 * cumulative frames; finite audio; a second pass gives identical bits; sentence k synthesized ALONE gives
the bits it has inside the batch (padding / batching invisible); the workspace stays within k82_reserve(128, 512, 2048).
Run once (fp32, about 3 s of GPU time per pass)."""
import numpy as np
import pytest
import torch

from tts_amd import spec

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_max_context_batch_128_properties(synthetic_sds):
    from tts_amd.engine import Engine
    B, T, DUR = 128, 512, 4
    rng = np.random.default_rng(0)
    ids = np.zeros((B, T), np.int64)
    ids[:, 1:511] = rng.integers(1, 178, size=(B, 510))
    lens = np.full(B, T, np.int32)
    pack = spec.synthetic_voice_pack(1)
    ref_s = np.stack([pack[509 - (b % 7), 0] for b in range(B)]).astype(np.float32)     # seven different style rows
    forced = np.full((B, T), DUR, np.int32)
    seeds = np.arange(B, dtype=np.uint64) + 500
    F = T * DUR
    total = B * F * 600
    eng = Engine(synthetic_sds)
    try:
        eng.reserve(B, T, F)
        cap_t, cap_f = eng.prof_read("arena_token")[1], eng.prof_read("arena_frame")[1]
        bufs = [torch.empty(total, dtype=torch.float32, device="cuda") for _ in range(2)]
        for buf in bufs:
            with eng.session():
                pred, nfr = eng.durations(ids, lens, ref_s, forced_dur=forced)
                _, off = eng.decode(seeds=seeds, out=(buf.data_ptr(), total))
            assert (pred == DUR).all() and (nfr == F).all()
            assert np.array_equal(off, 600 * F * np.arange(B + 1, dtype=np.int64))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(bufs[0]).all())
        assert float(bufs[0].abs().max()) > 1e-4                       # not silence
        assert torch.equal(bufs[0], bufs[1]), "two passes over the same batch differ"
        # no pass outgrew the reservation (the arenas only ever grow)
        assert eng.prof_read("arena_token")[1] == cap_t and eng.prof_read("arena_frame")[1] == cap_f
        used_t, used_f = eng.prof_read("arena_token")[2], eng.prof_read("arena_frame")[2]
        print(f"cfg4 workspace: token arena {used_t / 1e9:.2f} of {cap_t / 1e9:.2f} GB, frame arena {used_f / 1e9:.1f} of {cap_f / 1e9:.1f} GB")
        assert used_t <= cap_t and used_f <= cap_f
        # row k alone == row k inside the batch (first, a middle one with another style row, last)
        for k in (0, 61, B - 1):
            with eng.session():
                eng.durations(ids[k:k + 1], lens[k:k + 1], ref_s[k:k + 1], forced_dur=forced[k:k + 1])
                a, o = eng.decode(seeds=seeds[k:k + 1])
            assert o[1] == 600 * F
            row = bufs[0][k * 600 * F:(k + 1) * 600 * F].cpu().numpy()
            assert np.array_equal(a, row), (k, float(np.abs(a - row).max()))
    finally:
        eng.close()
