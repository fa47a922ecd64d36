# GPU box: the driver's own command (`python bench.py`, no flags) -- wall time and the keys of its one JSON line.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03_bench_default}; mkdir -p $O
t0=$SECONDS
timeout -k 10 700 python -u $GRAFT_REPO_ROOT/bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench.py failed"; tail -5 $O/bench_default.err; exit 1; }
echo "wall seconds: $((SECONDS - t0))"
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline()); r = d["roofline"]
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "frac", round(r["frac"], 4), "traffic", r["traffic"])
print("traffic_source:", r["traffic_source"])
print("cpu_baseline", d["cpu_baseline"]["value"], "first_audio_ms_p50", d["first_audio_ms_p50"], "others", {k: round(v["value"], 1) for k, v in d["other_dtypes"].items()})
PY
