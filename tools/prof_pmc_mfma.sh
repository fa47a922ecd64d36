set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03}_pmc_mfma; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in ${PMC_MODES:-fp32 bf16}; do
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $O/$m -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic --no-extras --dtype $m > /dev/null 2>&1
python3 $R/tools/pmc_mfma_table.py $O/$m $m >> $O/summary.txt
done
find $O -name "*counter_collection.csv" -delete
cat $O/summary.txt
