#!/bin/bash
# A/B of a rebuilt library against the previous build on one box (XM3D_LIB selects the baseline .so):
# attention micro-bench, attention parity tests, then the bench line of both.  usage: bash run/ab_attn.sh
set -e -o pipefail
mkdir -p gpurun_out
BASE=xmask3d_amd/ab/libxm3d_hip_base.so
python tools/attn_bench.py 20 > gpurun_out/ab_attn_new.log 2>&1
XM3D_LIB=$BASE python tools/attn_bench.py 20 > gpurun_out/ab_attn_base.log 2>&1
python -m pytest tests/test_gpu_attention.py -q -x -m gpu > gpurun_out/ab_attn_tests.log 2>&1
echo "tests done: $(tail -1 gpurun_out/ab_attn_tests.log)"
python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_new.log 2>&1
echo "bench new done"
XM3D_LIB=$BASE python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_base.log 2>&1
echo "bench base done"

grep -h '"value"' gpurun_out/ab_bench_new.log gpurun_out/ab_bench_base.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d.get('fp32', {}).get('value'))"
