#!/bin/bash
# A/B builds of the conv kernel on ONE box (devices differ by several %, so variants are only comparable
# inside one gpurun call).  CPU side: tools/ab_conv.sh build "<flags 0>" "<flags 1>" ...   (use "" for none)
# GPU box: tools/ab_conv.sh run <nvariants> [nshapes]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift; i=0
  for F in "$@"; do
    d=/tmp/ab_$i; rm -rf $d; mkdir -p $d
    for f in conv_mfma conv_ring ops engine; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $F -c tts_amd/csrc/$f.hip -o $d/$f.o & done
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $F -c tts_amd/csrc/source.hip -o $d/source.o
    wait
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tts_amd/libk82_v$i.so $d/*.o
    i=$((i+1))
  done
  ls -la tts_amd/libk82_v*.so
else
  n=$2
  for r in 1 2; do for ((v=0; v<n; v++)); do
    echo "== variant $v (round $r)"; K82_LIB_PATH=$PWD/tts_amd/libk82_v$v.so timeout -k 10 200 python -u tools/bench_conv.py ${3:-8} 2>/dev/null | grep -v "leaky\|amdgpu.ids"
  done; done
fi
