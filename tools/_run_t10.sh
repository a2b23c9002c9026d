set -e
S=$(date +%s)
timeout -k 10 400 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q --durations=5 2>&1 | tail -9
echo "parity wall $(( $(date +%s) - S )) s"
S=$(date +%s)
timeout -k 10 300 python bench.py --no-cpu-baseline --train-steps 0 2> gpurun_out/t10.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['fp32'])"
echo "bench wall $(( $(date +%s) - S )) s"
grep fp32 gpurun_out/t10.err | tail -2
