# GPU box: rocprofv3 kernel-trace summary of the configs[1] step in a given precision mode (default bf16)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${2:-r03_kt_bf16}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --dtype ${1:-bf16} > $O/bench.json 2> $O/bench.err || exit 1
find $O -name "*kernel_trace.csv" -delete
head -40 $O/kt/kt_kernel_stats.csv | cut -c1-200
