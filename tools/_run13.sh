set -o pipefail
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r2_gpu_tests2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_gpu_tests2.log
tail -15 gpurun_out/r2_gpu_tests2.log
