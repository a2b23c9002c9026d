set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_assign.py tests/test_gpu_attention.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r2_assign1.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r2_assign1.log; tail -15 gpurun_out/r2_assign1.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r2_train3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_train3.log; tail -8 gpurun_out/r2_train3.log
timeout -k 10 300 python tools/train_bench.py 1 8 fp32 graph 2>&1 | grep -v amdgpu.ids | tail -5
