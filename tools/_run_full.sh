set -e
S=$(date +%s)
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_default.log 2> gpurun_out/r02_bench_default.err
echo "bench wall $(( $(date +%s) - S )) s"
tail -1 gpurun_out/r02_bench_default.log
S=$(date +%s)
timeout -k 10 650 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests.log 2>&1
echo "tests wall $(( $(date +%s) - S )) s"
tail -3 gpurun_out/r02_gpu_tests.log
