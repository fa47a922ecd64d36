"""Build profiles/*_traffic.json from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`bench.py --steps 1 --warmup 0 --no-cpu-baseline`:  python tools/make_traffic.py <fetch_dir> <write_dir> <out.json> [dtype]

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are KB per
dispatch; FETCH_SIZE reads 1/2 of wide coalesced reads on gfx950, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(float)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "conv_mfma_kernel" in r["Kernel_Name"]:
                    per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
        for (_, k), v in per.items():
            acc[k].append(v)
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
out = {"command": f"rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --dtype {dtype} ; "
                  "same with --pmc WRITE_SIZE (separate passes)",
       "unit_note": "FETCH_SIZE/WRITE_SIZE are KB per dispatch; on gfx950 FETCH_SIZE reads 1/2 of wide coalesced reads "
                    "(MI355X_MICROARCH.md, HBM), so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; Infinity-Cache hits are counted. "
                    "The counts include the sizing pass and the token-rate launches of the same kernels.",
       "kernels": {}}
n = f_sum = w_sum = 0.0
for k in sorted(fetch):
    fam = "conv_mfma_kernel" in k and "<1, 4," not in k      # the 128-row family: 2x2, 4x1 and 4x2 wave shapes (not the 64-row 1x4 kernel)
    fv, wv = fetch[k], write.get(k, [])
    out["kernels"][k] = {"launches": len(fv), "FETCH_SIZE_KB_mean": sum(fv) / len(fv),
                         "WRITE_SIZE_KB_mean": (sum(wv) / len(wv)) if wv else None}
    if fam:
        n += len(fv); f_sum += sum(fv); w_sum += sum(wv)
out["conv_mfma_2x2_family"] = {"launches": int(n), "fetch_KB_per_launch_raw": f_sum / n, "write_KB_per_launch": w_sum / n,
                               "traffic_bytes_per_launch_corrected": (2 * f_sum + w_sum) / n * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["conv_mfma_2x2_family"]))
