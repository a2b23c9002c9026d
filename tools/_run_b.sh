B="--steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --fp32-steps 0"
timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('latency_ms_single_scene'), d['roofline_dense_stage']['ms_per_view'])"
