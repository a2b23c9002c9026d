#!/bin/bash
# A/B of the GEMM kernels of a rebuilt library against the previous build on one box.  usage: bash run/ab_gemm.sh
set -e -o pipefail
mkdir -p gpurun_out
BASE=xmask3d_amd/ab/libxm3d_hip_base.so
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_f32acc.py tests/test_gpu_conv_gemm.py -q -x -m gpu > gpurun_out/ab_gemm_tests.log 2>&1 || { tail -30 gpurun_out/ab_gemm_tests.log; exit 1; }
echo "tests: $(tail -n 1 gpurun_out/ab_gemm_tests.log)"
python tools/gemm_bench.py > gpurun_out/ab_gemm_new.log 2>&1
XM3D_LIB=$BASE python tools/gemm_bench.py > gpurun_out/ab_gemm_base.log 2>&1
for f in new base new2 base2; do
  if [ ${f#base} != $f ]; then export XM3D_LIB=$BASE; else unset XM3D_LIB; fi
  python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_$f.log 2>&1
  grep -h '"value"' gpurun_out/ab_bench_$f.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], d.get('fp32', {}).get('value'))"; done
