# GPU box: tools/pmc_conv.sh <tag> <precision> <cin> <cout> <L> <B> <taps> <dil> <act>  -- three counter passes, one summary
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
P3="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P -d $O/p$i -o p --output-format csv -- python3 $R/tools/pmc_conv.py "$@" > $O/run$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/run$i.log; exit 1; }
done
python3 - "$O" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "conv_mfma_kernel" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
m = {c: sorted(v)[len(v) // 2] for c, v in acc.items()}
for c in sorted(m):
    print(f"{c:40s} {m[c]:16.0f}")
cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
if cyc:
    print(f"cycles/launch {cyc:.0f}")
    for c, units in (("TA_TA_BUSY_sum", 256), ("TD_TD_BUSY_sum", 256)):
        if c in m: print(f"{c} per unit / cycles = {m[c] / units / cyc:.3f}")
if "SQ_BUSY_CU_CYCLES" in m:
    b = m["SQ_BUSY_CU_CYCLES"]
    print(f"mfma busy / (busy cu cycles) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / b / 4:.3f} (per SIMD: counter / 4)")
    print(f"wait_inst_any / wave_cycles = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}; wait_inst_lds = {m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES']:.3f}")
    print(f"active valu {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f} lds {m['SQ_ACTIVE_INST_LDS'] / m['SQ_WAVE_CYCLES']:.3f} vmem {m['SQ_ACTIVE_INST_VMEM'] / m['SQ_WAVE_CYCLES']:.3f} of wave cycles")
PY
find $O -name "*counter_collection.csv" -delete
cat $O/run1.log | tail -1
