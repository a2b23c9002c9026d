timeout -k 10 700 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r02_gpu_tests_dur.log 2>&1
tail -60 gpurun_out/r02_gpu_tests_dur.log
