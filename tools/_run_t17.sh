set -e
timeout -k 10 300 python -m pytest tests/test_gpu_attention.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q -k "bf16" 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --fp32-steps 0"
timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('latency_ms_single_scene'))"
