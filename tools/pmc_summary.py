"""Summarise rocprofv3 --pmc output: mean counter value per kernel (dispatch-level), printed as a table.
usage: python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [substring-of-kernel-name]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if filt and filt not in k:
                continue
            per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    print(k[:100])
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
