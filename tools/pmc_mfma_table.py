"""MFMA-pipe utilisation per conv instantiation from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES) over
`bench.py --steps 1 --warmup 0 --no-cpu-baseline`:  python tools/pmc_mfma_table.py <dir> [label]
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "conv_mfma_kernel" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
print("==", sys.argv[2] if len(sys.argv) > 2 else root)
rows = []
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(cs["GRBM_GUI_ACTIVE"])
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    rows.append((cyc * n, f"{k[:92]:92s} launches={n:4d}  cycles/launch={cyc:12.0f}  "
                 f"mfma_busy={100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:4.1f}%  "
                 f"wait_inst={100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:4.1f}%"))
for _, line in sorted(rows, reverse=True):
    print(line)
