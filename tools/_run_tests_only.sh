S=$(date +%s)
timeout -k 10 650 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02_gpu_tests.log 2>&1
echo "rc=$? tests wall $(( $(date +%s) - S )) s"
tail -14 gpurun_out/r02_gpu_tests.log | cut -c1-200
