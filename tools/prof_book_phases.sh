# GPU box: kernel trace of a book pass (tools/bench_book.py), split into phase 1 (albert_embed .. durations_kernel) and
# phase 2 (the rest) per batch: GPU-busy time, gaps between kernels, and the kernels phase 1 spends its time in.
# usage: bash tools/prof_book_phases.sh [dtype] [max_sentences] [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${3:-r03_book_phases}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 $R/tools/bench_book.py --dtype ${1:-bf16} --skip-latency --max-sentences ${2:-0} > $O/run.log 2>$O/run.err || exit 1
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
starts = [i for i, r in enumerate(rows) if "albert_embed" in r["Kernel_Name"]]
starts = starts[2:]                                  # the warm-up pass
tot = {"p1_wall": 0, "p1_busy": 0, "p2_wall": 0, "p2_busy": 0, "between": 0}
per1, per2 = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
def name(r):
    n = r["Kernel_Name"]
    return n.split("(")[0][:78]
for a, b in zip(starts, starts[1:] + [len(rows)]):
    j = next(i for i in range(a, b) if "durations_kernel" in rows[i]["Kernel_Name"])
    p1, p2 = rows[a:j + 1], rows[j + 1:b]
    tot["p1_wall"] += E(p1[-1]) - S(p1[0]); tot["p1_busy"] += sum(E(r) - S(r) for r in p1)
    if p2:
        tot["p2_wall"] += E(p2[-1]) - S(p2[0]); tot["p2_busy"] += sum(E(r) - S(r) for r in p2)
        tot["between"] += S(p2[0]) - E(p1[-1])
    for r in p1: per1[name(r)][0] += E(r) - S(r); per1[name(r)][1] += 1
    for r in p2: per2[name(r)][0] += E(r) - S(r); per2[name(r)][1] += 1
nb = len(starts)
print(f"{nb} batches; seconds: " + ", ".join(f"{k} {v * 1e-9:.3f}" for k, v in tot.items()))
print(f"whole trace {1e-9 * (E(rows[-1]) - S(rows[starts[0]])):.3f} s")
for title, per in (("phase 1", per1), ("phase 2", per2)):
    print(title)
    for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
        print(f"  {n:78s} {1e-9 * t:7.3f} s  {c:6d} launches  avg {1e-3 * t / c:8.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
