# GPU box: the round's profile set.  TAG=r03_final bash tools/prof_round.sh
#   bench.json            the driver's own command with K = 3, W = 1 (all extras, live PMC traffic passes, CPU baseline)
#   kt/, bench_kt.json    rocprofv3 --kernel-trace --stats of the SAME timed region and nothing else (--no-extras: sizing pass +
#                         1 warm-up + 3 timed steps = 4 steps in the summary; the conv rows / 4 agree with roofline.ms_per_step)
#   bench_<mode>.json     the other precision modes
#   traffic_<mode>.json   FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, tools/make_traffic.py)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras > $O/bench_kt.json 2> $O/bench_kt.err
echo kt done
timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --dtype bf16x3 > $O/bench_bf16x3.json 2>/dev/null
timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --dtype bf16 > $O/bench_bf16.json 2>/dev/null
echo modes done
for m in ${PMC_MODES:-fp32 bf16}; do for c in FETCH_SIZE WRITE_SIZE; do
timeout -s KILL 240 rocprofv3 --pmc $c -d $O/pmc_${m}_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic --no-extras --dtype $m > /dev/null 2>&1
echo pmc $m $c done
done
python3 $R/tools/make_traffic.py $O/pmc_${m}_FETCH_SIZE $O/pmc_${m}_WRITE_SIZE $O/traffic_$m.json $m
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
ls -R $O | head -40
