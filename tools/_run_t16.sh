set -e
timeout -k 10 200 python -m pytest tests/test_gpu_msda_fuse.py -m gpu -x -q -k group_norm 2>&1 | tail -2
B="--steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --fp32-steps 0"
timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], [ (k['kernel'][:30], round(k['achieved'],1), round(k['avg_launch_us'],1)) for k in d['roofline_kernels']])"
