set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_bench_parity.py -q -m gpu -s -p no:cacheprovider > gpurun_out/r2_parity2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_parity2.log
grep -E "parity|votes|passed|failed|Error|assert|rc=" gpurun_out/r2_parity2.log | tail -40
timeout -k 10 500 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider > gpurun_out/r2_train2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_train2.log
tail -30 gpurun_out/r2_train2.log
