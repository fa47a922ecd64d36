set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-traffic > $O/bench.json 2> $O/bench.err
echo kt done
timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --dtype bf16x3 > $O/bench_bf16x3.json 2>/dev/null
timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --dtype bf16 > $O/bench_bf16.json 2>/dev/null
echo modes done
for m in ${PMC_MODES:-fp32 bf16}; do for c in FETCH_SIZE WRITE_SIZE; do
timeout -s KILL 240 rocprofv3 --pmc $c -d $O/pmc_${m}_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic --no-extras --dtype $m > /dev/null 2>&1
echo pmc $m $c done
done
python3 $R/tools/make_traffic.py $O/pmc_${m}_FETCH_SIZE $O/pmc_${m}_WRITE_SIZE $O/traffic_$m.json $m
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
ls -R $O | head -40
