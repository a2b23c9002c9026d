#!/bin/bash
# A/B of the convolution / GEMM kernels of a rebuilt library against the previous build on one box.  usage: bash run/ab_conv.sh [nobench]
set -e -o pipefail
mkdir -p gpurun_out
BASE=xmask3d_amd/ab/libxm3d_hip_base.so
python tools/conv_bench.py 20 10 > gpurun_out/ab_conv_new.log 2>&1
XM3D_LIB=$BASE python tools/conv_bench.py 20 10 > gpurun_out/ab_conv_base.log 2>&1
python tools/gemm_bench.py > gpurun_out/ab_gemm_new.log 2>&1
XM3D_LIB=$BASE python tools/gemm_bench.py > gpurun_out/ab_gemm_base.log 2>&1
[ "$1" = nobench ] && exit 0
python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_new.log 2>&1
XM3D_LIB=$BASE python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_base.log 2>&1
python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_new2.log 2>&1
XM3D_LIB=$BASE python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_base2.log 2>&1
for f in new base new2 base2; do grep -h '"value"' gpurun_out/ab_bench_$f.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], d.get('fp32', {}).get('value'))"; done
