# GPU box: kernel trace of the single-sentence path (tools/prof_latency.py): GPU-busy time vs wall time per sentence
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_kt_latency; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 $R/tools/prof_latency.py ${1:-fp32} 20 > $O/run.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 10 sentences: split at albert_embed_kernel launches (one per durations call)
starts = [i for i, r in enumerate(rows) if "albert_embed" in r["Kernel_Name"]]
seg = rows[starts[-6]:starts[-1]]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"5 sentences: {len(seg)} kernels, wall {1e-6 * (t1 - t0) / 5:.2f} ms/sentence, GPU busy {1e-6 * busy / 5:.2f} ms/sentence, gaps {1e-6 * (t1 - t0 - busy) / 5:.2f} ms")
from collections import defaultdict
per = defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:60]
    per[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[n][1] += 1
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {n:60s} {1e-6 * t / 5:7.3f} ms  {c / 5:6.1f} launches  avg {1e-3 * t / c:7.1f} us")
# phase 1 only: from albert_embed to the durations kernel
k = starts[-2]
j = next(i for i in range(k, len(rows)) if "durations_kernel" in rows[i]["Kernel_Name"])
p1 = rows[k:j + 1]
b1 = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in p1)
print(f"phase 1: {len(p1)} kernels, wall {1e-6 * (int(p1[-1]['End_Timestamp']) - int(p1[0]['Start_Timestamp'])):.2f} ms, busy {1e-6 * b1:.2f} ms")
PY
find $O -name "*kernel_trace.csv" -delete
