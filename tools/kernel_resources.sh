#!/bin/bash
# Resource usage (VGPRs, scratch, spilled SGPRs, occupancy) of every conv_mfma_kernel instantiation, one line each.
# usage: tools/kernel_resources.sh [extra hipcc flags]  > table
cd "$(dirname "$0")/../tts_amd/csrc" || exit 1
for f in conv_mfma conv_ring; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c $f.hip -o /tmp/k82_res_$f.o 2>&1; done |
python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in sorted(rows.items()):
    t = re.search(r"conv_mfma_kernelI(.*)EEv5ConvP", name)
    print((t.group(1) if t else name).replace("Li", "").replace("Lb", "b").replace("E", ","), "VGPR", r.get("VGPRs"), "AGPR", r.get("AGPRs"),
          "scratch", r.get("ScratchSize"), "sgpr_spill", r.get("SGPRs Spill"), "occ", r.get("Occupancy"), "lds", r.get("LDS Size"))
'
