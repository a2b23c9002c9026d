#!/bin/bash
# A/B of a rebuilt library against the previous build on one box (XM3D_LIB selects the baseline .so):
# attention parity tests, attention micro-bench, then the bench line of both.  usage: bash run/ab_attn.sh [nobench]
set -e -o pipefail
mkdir -p gpurun_out
BASE=xmask3d_amd/ab/libxm3d_hip_base.so
python -m pytest tests/test_gpu_attention.py -q -x -m gpu > gpurun_out/ab_attn_tests.log 2>&1 || { tail -30 gpurun_out/ab_attn_tests.log; exit 1; }
echo "tests done: $(tail -1 gpurun_out/ab_attn_tests.log)"
python tools/attn_bench.py 20 > gpurun_out/ab_attn_new.log 2>&1
XM3D_LIB=$BASE python tools/attn_bench.py 20 > gpurun_out/ab_attn_base.log 2>&1
tail -n 4 gpurun_out/ab_attn_base.log; tail -n 4 gpurun_out/ab_attn_new.log
[ "$1" = nobench ] && exit 0
python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_new.log 2>&1
XM3D_LIB=$BASE python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/ab_bench_base.log 2>&1
grep -h '"value"' gpurun_out/ab_bench_new.log gpurun_out/ab_bench_base.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d.get('fp32', {}).get('value'))"
