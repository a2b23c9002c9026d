set -e
S=$(date +%s)
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_default.log 2> gpurun_out/r02_bench_default.err
echo "bench wall $(( $(date +%s) - S )) s"
tail -1 gpurun_out/r02_bench_default.log | cut -c1-220
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_driver_flags.log 2>/dev/null
tail -1 gpurun_out/r02_bench_driver_flags.log | cut -c1-220
S=$(date +%s)
timeout -k 10 650 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r02_gpu_tests.log 2>&1
echo "tests wall $(( $(date +%s) - S )) s"
tail -20 gpurun_out/r02_gpu_tests.log
