set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_attention.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r2_attn1.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r2_attn1.log; tail -25 gpurun_out/r2_attn1.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_gpu_bench_parity.py -q -m gpu -s -p no:cacheprovider > gpurun_out/r2_parity3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_parity3.log
grep -E "parity bf16|votes|passed|failed|Error|assert|rc=" gpurun_out/r2_parity3.log | tail -20
