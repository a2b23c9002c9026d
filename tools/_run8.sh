set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_attention.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r2_attn2.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r2_attn2.log; tail -5 gpurun_out/r2_attn2.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/attn_bench.py 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_attn_bench1.log
