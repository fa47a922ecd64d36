#!/usr/bin/env python3
"""Headline benchmark: Kokoro-82M audio-seconds synthesized per wall-second on N MI355X.

Workload at every N (weak scaling): BASELINE.json configs[1] per GPU -- fp32, a synthetic batch
of 64 phoneme sequences x 256 tokens (254 random ids + the two pad tokens), seeded synthetic
weights of the exact Kokoro-82M architecture (no checkpoint exists offline), style rows of a
synthetic voice pack, speed 1.0.  A step = the whole hot path for one batch: k82_durations
(ALBERT .. duration head) + k82_decode (length regulation .. iSTFT), durations PREDICTED by the
model (no forced durations), inputs already resident in HBM, audio left in HBM.

One process per GPU (torch.distributed over RCCL only for the barrier and the gather of two
scalars per rank -- sentences are independent, there is no data-path collective).  Rank 0 prints
ONE JSON line.  Launched under torchrun (RANK / WORLD_SIZE in the environment) it is one rank of
the job; launched plainly as `python bench.py --gpus N` with N > 1 it starts N fresh rank
processes itself (before this process has touched torch.cuda) and forwards rank 0's line.  It
refuses to run when --gpus disagrees with the world size it finds: an N-GPU request never
silently measures one GPU.  `roofline` is for the dominant kernel conv_mfma_kernel<2,2> (fp32 MFMA), timed with
HIP events on the engine's stream during the timed steps; `cpu_baseline` is the CPU oracle
(a stand-in "port": the reference's onnxruntime path is not installable offline) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 (same guide); only used with --dtype bf16
PEAK_FP32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the B = 1 latency loop and the other-precision legs")
    ap.add_argument("--latency-sentences", type=int, default=45)
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic on this box")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # the pass rocprofv3 runs: one step, no JSON extras
    ap.add_argument("--dtype", choices=("fp32", "bf16", "bf16x3"), default="fp32",
                    help="fp32 (BASELINE configs[1], the default); bf16 MFMA in the decoder/generator (cfg3's dtype); "
                         "bf16x3: the same convolutions with split operands, three bf16 MFMAs per product (fp32-grade)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks share GPU 0 and talk over gloo: exercises the launcher and the rank gather on a 1-GPU "
                         "box; the JSON line is marked rehearsal and is not a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                 f"(torchrun --nproc-per-node {args.gpus}) or run plain `python bench.py --gpus {args.gpus}`")

    import torch
    ngpu = torch.cuda.device_count()
    if args.rehearse_on_one_gpu:
        local = 0
    if ngpu < max(1, min(world, local + 1)):
        sys.exit(f"bench.py: rank {rank} needs GPU {local} but this machine exposes {ngpu} GPU(s) "
                 f"(--gpus {args.gpus} needs {args.gpus})")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from tts_amd import spec
    from tts_amd.engine import Engine

    sds = spec.synthetic_state_dicts(0)
    eng = Engine(sds, device=local, precision=args.dtype)
    bf16 = args.dtype != "fp32"
    # peak for ALGORITHMIC flops of the dominant kernel: bf16x3 issues three bf16 MFMAs per algorithmic product
    peak_tf = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}[args.dtype]
    B, T = args.batch, args.tokens
    rng = np.random.default_rng(1000 + rank)                 # each rank synthesizes different sentences
    ids = np.zeros((B, T), dtype=np.int64)
    ids[:, 1:T - 1] = rng.integers(1, 178, size=(B, T - 2))
    lens = np.full((B,), T, dtype=np.int32)
    pack = spec.synthetic_voice_pack(1)
    ref_s = np.repeat(pack[T - 3], B, axis=0).astype(np.float32)   # KPipeline row: pack[len(phonemes) - 1]
    seeds = np.arange(B, dtype=np.uint64) + np.uint64(rank * 100000)
    ids_d = torch.from_numpy(ids).cuda()
    ref_d = torch.from_numpy(ref_s).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    # sizing pass (also the first warm-up): durations are deterministic for fixed inputs
    _, nfr = eng.durations((ids_d.data_ptr(), T), lens, ref_d.data_ptr(), stream=stream)
    total = int(nfr.sum()) * spec.FRAME_SAMPLES
    audio_d = torch.empty(total, dtype=torch.float32, device="cuda")

    def step():
        eng.durations((ids_d.data_ptr(), T), lens, ref_d.data_ptr(), stream=stream)
        eng.decode(seeds=seeds, out=(audio_d.data_ptr(), total), stream=stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    conv_ms, conv_n, conv_flops = eng.prof_read("conv_mfma_2x2")
    fam = {}
    for name in ("conv_mfma_1x4", "lstm", "instnorm", "attention", "istft", "gather", "source", "other"):
        ms, n, w = eng.prof_read(name)
        fam[name] = {"ms_per_step": ms / args.steps, "launches_per_step": n / args.steps}
        if ms > 0 and w > 0:
            if name in ("instnorm", "istft", "gather"):          # HBM-bound: algorithmic bytes / HIP-event time
                gbs = w / (ms * 1e-3) / 1e9
                fam[name].update({"bound": "hbm", "achieved_GBps": gbs, "frac_of_8TBps": gbs / PEAK_HBM_GBPS})
            elif name == "conv_mfma_1x4":
                fam[name].update({"bound": "mfma", "achieved_TFLOPs": w / (ms * 1e-3) / 1e12})
            elif name == "lstm":                                  # latency-bound: sequential steps per second
                fam[name].update({"bound": "latency", "sentence_steps_per_s": w / (ms * 1e-3)})
    eng.prof_enable(False)

    # ---- the rest of BASELINE.json's metric, outside the timed region (rank 0 of an N = 1 run only): the same step in the
    # other two precision modes, and "p50 first-audio ms" -- one sentence at a time, B = 1, call -> float32 PCM on the host
    extra = {}
    if world == 1 and not args.no_extras:
        extra = extras(eng, args, step, audio_s=total / spec.SAMPLE_RATE)

    audio_s = total / spec.SAMPLE_RATE
    per_rank = [[dt, audio_s]]
    if world > 1:                      # every rank's (seconds, audio-seconds per step): value uses the SLOWEST rank's time
        mine = torch.tensor([dt, audio_s], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v[0].item()), float(v[1].item())] for v in allr]
    dt = max(p[0] for p in per_rank)
    audio_all = sum(p[1] for p in per_rank)
    finite = bool(torch.isfinite(audio_d).all().item())

    if rank == 0:
        # HBM bytes per launch of the dominant kernel family come from the committed rocprofv3 PMC passes of this
        # same command (counters cannot be read from inside the process); see profiles/*_traffic.json.
        traffic, traffic_src = None, None
        under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
        if world == 1 and not args.no_traffic and not args.traffic_child and not under_profiler:
            traffic, traffic_src = measure_traffic(args)
        elif under_profiler:
            traffic_src = "this run is itself under a profiler"
        tname = {"fp32": "r03_traffic.json", "bf16": "r03_traffic_bf16.json"}.get(args.dtype)
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tname or "")
        if traffic is None and tname and os.path.exists(tp) and B == 64 and T == 256:
            with open(tp) as fh:
                traffic = json.load(fh)["conv_mfma_2x2_family"]["traffic_bytes_per_launch_corrected"]
            traffic_src = (f"profiles/{tname} (a committed builder-run figure, NOT measured in this run: " + (traffic_src or "PMC passes skipped") +
                           "): (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes (tools/make_traffic.py)")
        out = {
            "metric": "audio-sec synthesized/sec (node) Kokoro-82M",
            "value": audio_all * args.steps / dt,
            "unit": "audio-s/s",
            "n_gpus": world,
            **({"rehearsal": "all ranks on GPU 0 over gloo: NOT a measurement"} if args.rehearse_on_one_gpu else {}),
            "per_rank_audio_s_per_s": [p[1] * args.steps / p[0] for p in per_rank],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 (decoder/generator MFMA operands; fp32 accumulate, activations, durations, F0)",
                      "bf16x3": "f32 emulated on the bf16 matrix pipe in the decoder/generator (operands split hi+lo, 3 MFMAs per "
                                "product, fp32 accumulate; 2^-16 relative per product); f32 MFMA elsewhere"}[args.dtype],
            "data": "synthetic phoneme ids, seeded random weights of the Kokoro-82M architecture, synthetic voice pack",
            "config": {"workload": (f"configs[1] shape in {args.dtype} mode" if bf16 else "configs[1]") + ": Kokoro-82M, 64 phoneme sequences x 256 tokens per GPU, "
                                   "durations predicted by the model", "batch_per_gpu": B, "tokens": T,
                       "frames_per_gpu": int(nfr.sum()), "audio_s_per_step_per_gpu": audio_s,
                       "parallelism": f"dp{world} (sentence sharding, no data-path collective)",
                       "audio_finite": finite},
            "roofline": {
                "kernel": "conv_mfma_kernel, 128-row family (implicit-GEMM Conv1d; " + ("4x1 / 4x2 / 2x2 waves, v_mfma_f32_32x32x16_bf16)" if bf16 else "2x2 waves, v_mfma_f32_32x32x2_f32)"),
                "bound": "mfma",
                "achieved": conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None,
                "peak": peak_tf,
                "unit": "TFLOP/s",
                "frac": (conv_flops / (conv_ms * 1e-3) / 1e12 / peak_tf) if conv_ms > 0 else None,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches_per_step": conv_n / args.steps,
                "avg_launch_ms": conv_ms / conv_n if conv_n else None,
                "ms_per_step": conv_ms / args.steps,
                "algorithmic_flops_per_step": conv_flops / args.steps,
            },
            "other_kernels": fam,
            **extra,
        }
        assert len(per_rank) == args.gpus == world
        if not args.no_cpu_baseline and world == 1:          # reported baseline: rank 0, N = 1 only
            out["cpu_baseline"] = cpu_baseline(sds, ids, ref_s, T)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_traffic(args, timeout_s=120):
    """roofline.traffic of THIS box: HBM bytes per launch of the dominant kernel family from the PMC counters, collected
    as MI355X_MICROARCH.md prescribes -- two separate child runs, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    (nothing else on the command line), each over one step of this same workload (`--traffic-child`: one sizing pass
    + one step, no extras); both counters are KB per dispatch and FETCH_SIZE reads half of wide coalesced reads on
    gfx950, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The children are fresh processes started with
    subprocess (this one keeps running and is not replaced); the program after `--` is python3 itself.
    Returns (bytes per launch, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="k82_pmc_")
    sums, launches = {}, 0
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--dtype", args.dtype,
                   "--batch", str(args.batch), "--tokens", str(args.tokens)]
            env = dict(os.environ, TMPDIR=tmp)
            try:
                r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} pass exceeded {timeout_s} s"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass exited with {r.returncode}"
            per = {}
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = row["Kernel_Name"]
                        # the 128-row family the roofline object is about (every wave shape of it), not the 32 / 64-row kernels
                        if row["Counter_Name"] == counter and "conv_mfma_kernel" in k and "<1, 4," not in k:
                            key = row["Dispatch_Id"]
                            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
            if not per:
                return None, f"the --pmc {counter} pass recorded no conv_mfma_kernel dispatch"
            sums[counter] = sum(per.values())
            launches = len(per)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return ((2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0 / launches,
            f"measured in this run: two child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over one step of this workload "
            f"({launches} launches of the family incl. the sizing pass); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md)")


def extras(eng, args, step, audio_s):
    """Extra keys of the JSON line (BASELINE.json metric: "...; p50 first-audio ms"; configs[2] is quoted in bf16).

    first_audio_ms_*: configs[4] -- sentences one at a time (B = 1) through Engine.synthesize, host ids in, float32 PCM on
    the host out; lengths follow the first sentences of Sound-of-Sirens.epub (tests/golden/epub_sound_of_sirens.json.gz
    holds the character lengths only; the phoneme ids are synthetic, one per character, as everywhere in this bench); the
    first 5 calls (workspace growth) are dropped.  Play/next path: speech7.py:124-133,273-350.
    other_dtypes: the timed step (same batch, same inputs) in the other precision modes, 1 warm-up + 2 timed steps each.
    """
    import gzip
    import torch
    from tts_amd import spec, text as T
    out = {}
    peak = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}
    with gzip.open(os.path.join(ROOT, "tests", "golden", "epub_sound_of_sirens.json.gz"), "rt", encoding="utf-8") as fh:
        lens = [min(max(int(n), 1), T.MAX_PHONEME_LENGTH) for n in json.load(fh)["char_lengths"][:args.latency_sentences]]
    rng = np.random.default_rng(4242)
    sent = [rng.integers(1, 178, size=n).tolist() for n in lens]
    pack = spec.synthetic_voice_pack(1)

    def latency():
        lat, aud = [], []
        for k, x in enumerate(sent):
            r = T.style_row(pack, len(x), "kokoro_onnx")[None]
            t0 = time.perf_counter()
            audios, _ = eng.synthesize([x], r, speed=1.0, seeds=[k])
            lat.append((time.perf_counter() - t0) * 1e3)
            aud.append(len(audios[0]) / spec.SAMPLE_RATE)
        lat, aud = np.array(lat[5:]), np.array(aud[5:])
        return {"p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)), "mean_audio_s": float(aud.mean()),
                "sentences": int(lat.size), "mean_tokens": float(np.mean([len(x) + 2 for x in sent[5:]]))}

    first = {}
    others = {}
    for mode in (args.dtype,) + tuple(m for m in ("fp32", "bf16x3", "bf16") if m != args.dtype):
        eng.set_precision(mode)
        if mode != args.dtype:
            step()
            torch.cuda.synchronize()
            eng.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            ms, n, fl = eng.prof_read("conv_mfma_2x2")
            eng.prof_enable(False)
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else None
            others[mode] = {"value": audio_s / dt, "unit": "audio-s/s", "ms_per_step": dt * 1e3, "steps": 2,
                            "roofline": {"bound": "mfma", "achieved": tf, "peak": peak[mode], "unit": "TFLOP/s",
                                         "frac": tf / peak[mode] if tf else None, "ms_per_step": ms / 2}}
        first[mode] = latency()
    eng.set_precision(args.dtype)
    out["first_audio_ms_p50"] = first[args.dtype]["p50"]
    out["first_audio_ms_p95"] = first[args.dtype]["p95"]
    out["first_audio"] = {"workload": "configs[4]: B = 1, one sentence per call, host ids in -> float32 PCM on the host; sentence lengths "
                                      "of Sound-of-Sirens.epub (first sentences), synthetic phoneme ids", "unit": "ms", "by_dtype": first}
    out["other_dtypes"] = others
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not imported
    torch or touched the GPU, and does not replace itself), wait for all of them, forward rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs} (--gpus {n})\n")
        return next(rc for rc in rcs if rc) if all(isinstance(rc, int) for rc in rcs) else 1
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _physical_cores():
    """Physical cores this process may run on: distinct (package, core id) pairs of the CPUs in its affinity mask."""
    cpus = sorted(os.sched_getaffinity(0))
    seen = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id") as fh:
                core = fh.read().strip()
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as fh:
                pkg = fh.read().strip()
            seen.add((pkg, core))
        except OSError:
            seen.add(("?", str(c)))
    return max(1, len(seen)), len(cpus)


def cpu_baseline(sds, ids, ref_s, T, budget_s=24.0):
    """the CPU oracle (kind "port": a stand-in for the reference's onnxruntime-CPU path, which is not
    installable offline) on the first sentences of the same batch, B = 1 sequentially as the reference runs, in two
    legs -- one thread, and all the physical cores this process is allowed (capped by OMP_NUM_THREADS, which the GPU
    box sets to its per-GPU CPU share) -- for about budget_s seconds of CPU work in total.  `value` is the faster
    leg (what a user of the reference would get from this host), both legs are reported."""
    import warnings
    warnings.filterwarnings("ignore")
    import torch
    from tts_amd import spec
    from oracle import kokoro_ref as R
    model = R.build(spec.DEFAULT_CONFIG, sds)
    phys, logical = _physical_cores()
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    many = max(1, min(phys, cap) if cap > 0 else phys)
    legs = []
    for threads, share in ((many, 0.6), (1, 0.4)):
        torch.set_num_threads(threads)
        done_audio, n = 0.0, 0
        t0 = time.perf_counter()
        while n < ids.shape[0] and (n == 0 or time.perf_counter() - t0 < budget_s * share):
            x = torch.from_numpy(ids[n:n + 1, :T].copy())
            a, _ = model.forward_with_tokens(x, torch.from_numpy(ref_s[n:n + 1].copy()), 1.0)
            done_audio += a.numel() / spec.SAMPLE_RATE
            n += 1
        dt = time.perf_counter() - t0
        legs.append({"threads": threads, "audio_s_per_s": done_audio / dt, "sentences": n, "cpu_seconds": dt})
    best = max(legs, key=lambda l: l["audio_s_per_s"])
    return {"value": best["audio_s_per_s"], "unit": "audio-s/s", "cores": best["threads"], "kind": "port",
            "cpu_model": _cpu_model(), "physical_cores_available": phys, "logical_cpus_available": logical,
            "legs": legs,
            "sample": f"first {legs[0]['sentences']} (all-core leg, {legs[0]['cpu_seconds']:.1f} s) / first {legs[1]['sentences']} "
                      f"(1-thread leg, {legs[1]['cpu_seconds']:.1f} s) of the 64 sentences of the step (256 tokens each), batch=1 "
                      "sequential as the reference runs, fp32 torch-CPU restatement (oracle/kokoro_ref.py)"}


if __name__ == "__main__":
    main()
