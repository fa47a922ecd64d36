# GPU box: kernel trace of the first-audio loop (tools/latency_ab.py): GPU-busy time vs wall time per sentence, top kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${2:-r03_kt_first_audio}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 $R/tools/latency_ab.py ${1:-fp32} 40 > $O/run.log 2>&1 || exit 1
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
starts = [i for i, r in enumerate(rows) if "albert_embed" in r["Kernel_Name"]]
starts = starts[len(starts) // 2:]                      # the second repetition (warm)
n = len(starts) - 1
seg = rows[starts[0]:starts[-1]]
busy = sum(E(r) - S(r) for r in seg)
wall = 0; p1w = 0; p1b = 0; gaps_in = 0
for a, b in zip(starts, starts[1:]):
    s = rows[a:b]
    wall += E(s[-1]) - S(s[0])
    j = next(i for i in range(len(s)) if "durations_kernel" in s[i]["Kernel_Name"])
    p1w += E(s[j]) - S(s[0]); p1b += sum(E(r) - S(r) for r in s[:j + 1])
print(f"{n} sentences: {len(seg) / n:.0f} kernels each; first kernel -> last kernel {1e-6 * wall / n:.2f} ms, GPU busy {1e-6 * busy / n:.2f} ms; phase 1 {1e-6 * p1w / n:.2f} ms (busy {1e-6 * p1b / n:.2f})")
per = defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0][:70]
    per[k][0] += E(r) - S(r); per[k][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"  {k:70s} {1e-6 * t / n:7.3f} ms  {c / n:6.1f} launches  avg {1e-3 * t / c:7.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
