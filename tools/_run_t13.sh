set -e
timeout -k 10 200 python -m pytest tests/test_gpu_attention.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 400 python tools/prof_lines.py 20 > gpurun_out/r02_prof_lines.txt 2>&1
head -75 gpurun_out/r02_prof_lines.txt
