"""Python face of the C-ABI engine (include/k82.h): one handle = weights on one GPU.

Host arrays are numpy; device buffers may be passed as raw integer pointers (e.g.
``torch_tensor.data_ptr()``).  No arithmetic happens here -- everything runs in libk82.so, and
every failure of the library is raised as RuntimeError/ValueError (the reference's callers
wrap ``create`` in ``try/except Exception`` and fall back: speech7.py:126-133).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
import time

import numpy as np

from . import _lib, spec, weights as W


def _cfg_struct(config):
    pl = config["plbert"]
    return _lib.K82Config(
        n_token=config["n_token"], hidden_dim=config["hidden_dim"], style_dim=config["style_dim"],
        n_layer=config["n_layer"], max_dur=config["max_dur"], text_kernel=config["text_encoder_kernel_size"],
        bert_hidden=pl["hidden_size"], bert_heads=pl["num_attention_heads"], bert_inter=pl["intermediate_size"],
        bert_layers=pl["num_hidden_layers"], bert_max_pos=pl["max_position_embeddings"],
        bert_embed=spec.ALBERT_EMBED)


def _ptr(a):
    """numpy array / int device pointer / None -> c_void_p."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return a.ctypes.data_as(C.c_void_p)


def _host_buffer(n):
    """float32[n] host buffer for the D2H copy of a batch's audio: page-locked when large (a pageable target makes
    the runtime stage the copy through its own bounce buffer at a fraction of the PCIe rate); the returned ndarray
    keeps the pinned allocation alive."""
    if n >= (1 << 20):
        try:
            import torch
            return torch.empty((n,), dtype=torch.float32, pin_memory=True).numpy()
        except Exception:
            pass
    return np.empty((n,), dtype=np.float32)


class Engine:
    """Kokoro-82M on one MI355X.  ``state_dicts``: the 5 raw checkpoint state dicts."""

    def __init__(self, state_dicts, config=None, device=0, precision="fp32"):
        self.lib = _lib.load()
        # k82_durations leaves the token-rate state of ITS batch inside the handle for k82_decode, so the pair is one
        # critical section: every caller that may share this Engine with another thread (both front doors do, and the
        # reference runs two pool threads on one Kokoro object: epubviewer_06_latest_1_nov_2025_theme.py:407,692,715)
        # runs the pair inside ``with engine.session():``.  Re-entrant, so durations()/decode() take it themselves too.
        self._lock = threading.RLock()
        self.host_alloc_s = 0.0            # seconds spent allocating (page-locked) host buffers for audio
        self._pending = None               # (owner thread id, n_frames) of the durations call a decode may follow
        self.config = config or spec.DEFAULT_CONFIG
        self.device = device
        self.h = C.c_void_p()
        self._parent = None                # Engine.fork(): the engine whose weights this one reads
        self._forks = []
        cfg = _cfg_struct(self.config)
        rc = self.lib.k82_create(C.byref(cfg), device, C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"k82_create failed ({rc}): {self.lib.k82_last_error(None).decode()}")
        flat = W.fold_state_dicts(state_dicts, self.config)
        for name, arr in flat.items():
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * arr.ndim)(*arr.shape)
            self._check(self.lib.k82_set_tensor(self.h, name.encode(), _ptr(arr), shape, arr.ndim))
        self._check(self.lib.k82_finalize(self.h))
        self.set_precision(precision)

    def fork(self):
        """k82_fork: a second execution context on THIS engine's weights (own stream, workspace and phase-1 state; no
        second upload).  Calls on the two objects do not serialise against each other: a thread per object overlaps one
        batch's token-rate phase with another batch's frame-rate phase on the same GPU (BatchSynthesizer(contexts=2)).
        Same kernels and launch rules, so a batch gives the same bits on either.  Close forks before their parent."""
        if self._parent is not None:
            raise RuntimeError("Engine.fork: fork the engine that owns the weights, not one of its forks")
        child = object.__new__(Engine)
        child.lib = self.lib
        child._lock = threading.RLock()
        child.host_alloc_s = 0.0
        child._pending = None
        child.config, child.device = self.config, self.device
        child._parent = self
        child._forks = []
        child.h = C.c_void_p()
        with self._lock:
            rc = self.lib.k82_fork(self.h, C.byref(child.h))
            if rc != 0:
                raise RuntimeError(f"k82_fork failed ({rc}): {self.lib.k82_last_error(None).decode()}")
            child.precision = self.precision
            self._forks.append(child)
        return child

    def set_precision(self, precision):
        """"fp32" (default), "bf16" (bf16 MFMA for the decoder / generator convolutions) or "bf16x3" (the same
        convolutions with every operand split into two bf16 numbers and three MFMAs per product: fp32-grade
        results from the bf16 matrix pipe) -- k82_set_precision."""
        mode = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2}[precision]
        self._check(self.lib.k82_set_precision(self.h, mode))
        self.precision = ("fp32", "bf16", "bf16x3")[mode]
        for f in self._forks:              # the contexts of one engine compute in one mode
            f.set_precision(precision)

    def reserve(self, batch, tokens, frames):
        """k82_reserve: size the workspace once for batches up to ``batch`` sentences x ``tokens`` tokens x ``frames``
        predicted frames per sentence, so no later pass of that size allocates.  A request the handle already covers
        is a no-op (a second front-door object on a shared engine neither re-allocates nor disturbs the state a
        durations() call left); a request that does not fit leaves the workspace as it was and raises."""
        with self._lock:
            want = (int(batch) * ((int(tokens) + 31) // 32 * 32), int(batch) * int(frames))
            have = getattr(self, "_reserved", (0, 0))
            if want[0] <= have[0] and want[1] <= have[1]:
                return
            self._pending = None
            self._check(self.lib.k82_reserve(self.h, int(batch), int(tokens), int(frames)))
            self._reserved = (max(want[0], have[0]), max(want[1], have[1]))

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.k82_last_error(self.h).decode()
            if rc in (-22, -28):
                raise ValueError(f"k82 error {rc}: {msg}")
            raise RuntimeError(f"k82 error {rc}: {msg}")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            for f in list(getattr(self, "_forks", ())):      # forks read this engine's weights: they go first
                f.close()
            self.lib.k82_destroy(self.h)
            self.h = C.c_void_p()
            par = getattr(self, "_parent", None)
            if par is not None and self in par._forks:
                par._forks.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ the hot path
    @contextlib.contextmanager
    def session(self):
        """Exclusive use of the handle across a durations() -> decode() pair (or any sequence of calls whose
        results depend on each other: set_precision / debug taps).  Other threads block at their own session()."""
        with self._lock:
            yield self

    @staticmethod
    def pack(id_lists):
        """list of per-sentence id lists (WITHOUT pads) -> (ids i64[B,Tmax] with [0,*ids,0], lens i32[B])."""
        lens = np.array([len(x) + 2 for x in id_lists], dtype=np.int32)
        tmax = int(lens.max())
        ids = np.zeros((len(id_lists), tmax), dtype=np.int64)
        for b, x in enumerate(id_lists):
            ids[b, 1:1 + len(x)] = np.asarray(x, dtype=np.int64)
        return ids, lens

    def durations(self, ids, lens, ref_s, speed=None, forced_dur=None, stream=None):
        """Phase 1.  Returns (pred_dur i32[B,Tmax], n_frames i64[B])."""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        B = lens.shape[0]
        if isinstance(ids, np.ndarray):
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            tmax = ids.shape[1]
        else:
            ids, tmax = ids                      # (device pointer, Tmax)
        if isinstance(ref_s, np.ndarray):
            ref_s = np.ascontiguousarray(ref_s, dtype=np.float32).reshape(B, spec.STYLE_DIM_TOTAL)
        sp = None if speed is None else np.ascontiguousarray(np.broadcast_to(np.asarray(speed, np.float32), (B,)))
        fd = None if forced_dur is None else np.ascontiguousarray(forced_dur, dtype=np.int32).reshape(B, tmax)
        pred = np.zeros((B, tmax), dtype=np.int32)
        nfr = np.zeros((B,), dtype=np.int64)
        with self._lock:
            self._pending = None
            self._check(self.lib.k82_durations(self.h, _ptr(ids), _ptr(lens), B, tmax, _ptr(ref_s), _ptr(sp), _ptr(fd),
                                               _ptr(pred), _ptr(nfr), _ptr(stream)))
            self._pending = (threading.get_ident(), nfr)
        return pred, nfr

    def decode(self, noise=None, seeds=None, out=None, stream=None):
        """Phase 2 for the batch of the preceding ``durations`` call OF THIS THREAD.

        ``out``: None -> host numpy buffer; or (device_ptr, capacity_floats).  Returns
        (audio, offsets i64[B+1]); audio is the numpy buffer or None when ``out`` was given.
        Raises RuntimeError when another thread's durations() came in between (the pair was not run inside
        ``with engine.session():``) instead of decoding that other batch.
        """
        with self._lock:
            pend = self._pending
            if pend is None or pend[0] != threading.get_ident():
                raise RuntimeError("Engine.decode: the handle does not hold this thread's durations() state -- run the "
                                   "durations()/decode() pair inside `with engine.session():` when threads share an Engine")
            nfr = pend[1]
            B = nfr.shape[0]
            total = int(nfr.sum()) * spec.FRAME_SAMPLES
            nz = K = None
            if noise is not None:
                nz = np.ascontiguousarray(noise, dtype=np.float32)
                if nz.size != total * 9:
                    raise ValueError(f"noise holds {nz.size} values, the batch needs {total * 9}")
            if seeds is not None:
                K = np.ascontiguousarray(seeds, dtype=np.uint64)
                if K.shape != (B,):
                    raise ValueError(f"seeds has shape {K.shape}, the batch has {B} sentences")
            ns = _lib.K82Noise(noise=_ptr(nz), seeds=_ptr(K))
            off = np.zeros((B + 1,), dtype=np.int64)
            if out is None:
                t0 = time.perf_counter()
                audio = _host_buffer(total)
                self.host_alloc_s += time.perf_counter() - t0
                self._check(self.lib.k82_decode(self.h, C.byref(ns), _ptr(audio), total, _ptr(off), _ptr(stream)))
                return audio, off
            ptr, cap = out
            self._check(self.lib.k82_decode(self.h, C.byref(ns), _ptr(ptr), int(cap), _ptr(off), _ptr(stream)))
            return None, off

    def synthesize(self, id_lists, ref_s, speed=None, noise=None, seeds=None):
        """list of id lists -> list of f32 audio arrays (+ pred_dur).  Atomic with respect to other threads."""
        ids, lens = self.pack(id_lists)
        with self.session():
            pred, _ = self.durations(ids, lens, ref_s, speed)
            audio, off = self.decode(noise=noise, seeds=seeds)
        return [audio[off[b]:off[b + 1]] for b in range(len(id_lists))], pred

    def pcm16(self, audio):
        a = np.ascontiguousarray(audio, dtype=np.float32)
        out = np.empty(a.shape, dtype=np.int16)
        self._check(self.lib.k82_pcm16(self.h, _ptr(a), _ptr(out), a.size, None))
        return out

    def pcm16_device(self, src_ptr, dst_ptr, n, stream=None):
        """k82_pcm16 on buffers that already live on the GPU (raw device pointers: float32 in, int16 out, ``n`` samples):
        the conversion the bulk N-GPU gather does BEFORE its one copy off the device (scheduler.ordered_gather_pcm16)."""
        self._check(self.lib.k82_pcm16(self.h, _ptr(int(src_ptr)), _ptr(int(dst_ptr)), int(n), _ptr(stream)))

    def pcm16_static(self, audio, out=None):
        """k82_pcm16_static: the same conversion without any allocation on the device side (static staging array,
        at most ``K82_PCM16_STATIC_MAX`` samples per call).  ``out``: optional preallocated int16 array."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        if out is None:
            out = np.empty(a.shape, dtype=np.int16)
        if out.dtype != np.int16 or out.size < a.size or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous int16 array of at least audio.size elements")
        self._check(self.lib.k82_pcm16_static(self.h, _ptr(a), _ptr(out), a.size, None))
        return out[:a.size] if out.ndim == 1 else out

    # ------------------------------------------------------------------ hooks
    def prof_enable(self, on=True):
        self._check(self.lib.k82_prof_enable(self.h, 1 if on else 0))

    def prof_read(self, name):
        ms, n, w = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.k82_prof_read(self.h, name.encode(), C.byref(ms), C.byref(n), C.byref(w)))
        return ms.value, n.value, w.value

    def debug_enable(self, on=True):
        self._check(self.lib.k82_debug_enable(self.h, 1 if on else 0))

    def debug_get(self, name):
        """-> f32[B, C, pitch] (padded; slice the valid region yourself)."""
        dims = (C.c_int64 * 3)()
        self._check(self.lib.k82_debug_info(self.h, name.encode(), dims))
        out = np.empty(tuple(int(d) for d in dims), dtype=np.float32)
        self._check(self.lib.k82_debug_copy(self.h, name.encode(), _ptr(out)))
        return out

    def debug_force(self, name, data=None):
        """k82_debug_force: pin stage ``name`` ("F0", "N": f32[2F]; "har": f32[22, 120F+1]) of the next decode()
        calls to ``data`` (None removes the override).  Single-sentence batches; parity tests only."""
        if data is None:
            self._check(self.lib.k82_debug_force(self.h, name.encode(), None, 0, 0))
            return
        a = np.ascontiguousarray(data, dtype=np.float32)
        a = a.reshape(1, -1) if a.ndim == 1 else a
        self._check(self.lib.k82_debug_force(self.h, name.encode(), _ptr(a), a.shape[0], a.shape[1]))

    def op_conv1d(self, x, w, bias=None, dil=1, act=0, slope=1.0, alpha=None, iters=1):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        B, cin, L = x.shape
        cout, _, taps = w.shape
        y = np.empty((B, cout, L), np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        a = None if alpha is None else np.ascontiguousarray(alpha, np.float32)
        ms = C.c_float()
        self._check(self.lib.k82_op_conv1d(self.h, _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, cin, cout, L, taps, dil,
                                           act, slope, _ptr(a), iters, C.byref(ms)))
        return y, ms.value
