#!/bin/bash
# Timing-only ablation builds of csrc/conv.hip (results are wrong by construction): where does a stage spend its time?
# usage (on the GPU box): bash tools/conv_ablate.sh "1 2 4 8 16 6" "vae 512@128"
set -e
cd "$(dirname "$0")/../xmask3d_amd/csrc"
OBJS=$(ls *.o | grep -v '^conv\.o$' | tr '\n' ' ')
for n in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -mllvm -pragma-unroll-threshold=1000000 -DCV_ABL=$n -c conv.hip -o /tmp/conv_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/conv_abl$n.o -o /tmp/libxm3d_abl$n.so
  echo "== CV_ABL=$n"
  XM3D_LIB=/tmp/libxm3d_abl$n.so python ../../tools/conv_bench.py 20 5 "$2" 2>&1 | grep -v amdgpu.ids | sed 's/library conv.*//'
done
