# kernel-trace stats of one bench.py run: tools/prof_kt.sh <tag> [bench args...]   (GPU box)
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -25 $O/kernel_stats.csv | cut -c1-260
