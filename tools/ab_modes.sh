#!/bin/bash
# GPU box: tools/ab_modes.sh <nvariants> [nshapes] -- every variant library in fp32 and bf16x3 (and bf16) mode, one box
cd "$(dirname "$0")/.."
n=$1
for prec in ${K82_MODES:-fp32 bf16x3 bf16}; do for ((v=0; v<n; v++)); do
  echo "== variant $v $prec"
  K82_PRECISION=$prec K82_LIB_PATH=$PWD/tts_amd/libk82_v$v.so timeout -k 10 200 python -u tools/bench_conv.py ${2:-8} 2>/dev/null | grep -v "leaky\|amdgpu.ids\|precision" || exit 1
done; done
