#!/bin/bash
# A/B of whole-bench rates between the libraries built by tools/ab_conv.sh build (GPU box): tools/ab_bench.sh <nvariants> [dtype]
cd "$(dirname "$0")/.."
n=$1; dt=${2:-fp32}
for r in 1 2; do for ((v=0; v<n; v++)); do
  echo "== variant $v (round $r)"
  K82_LIB_PATH=$PWD/tts_amd/libk82_v$v.so timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --dtype $dt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f  ms/step %.1f  conv frac %.4f  avg_launch_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" || exit 1
done; done
