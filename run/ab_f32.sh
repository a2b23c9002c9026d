#!/bin/bash
# fp32 configuration A/B on one box: in-kernel operand split (default) against the split pass (XM3D_GEMM_F32_SPLIT=pass).  usage: bash run/ab_f32.sh
set -e -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_f32acc.py tests/test_gpu_conv_gemm.py tests/test_gpu_gemm.py -q -x -m gpu > gpurun_out/ab_f32_tests.log 2>&1 || { tail -n 30 gpurun_out/ab_f32_tests.log; exit 1; }
echo "tests: $(tail -n 1 gpurun_out/ab_f32_tests.log)"
for f in kernel pass kernel pass; do
  XM3D_GEMM_F32_SPLIT=$f python bench.py --no-cpu-baseline --train-steps 0 --dtype fp32 --fp32-steps 0 > gpurun_out/ab_f32_$f.log 2>&1
  grep -h '"value"' gpurun_out/ab_f32_$f.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'])"; done
