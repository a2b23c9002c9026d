set -e
B="--steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --fp32-steps 0"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('latency_ms_single_scene'))"; }
timeout -k 10 200 python bench.py $B 2>/dev/null | pick default
XM3D_ATTENTION=library timeout -k 10 200 python bench.py $B 2>/dev/null | pick attn_library
XM3D_SPCONV_ALGO=tiles timeout -k 10 200 python bench.py $B 2>/dev/null | pick spconv_tiles
timeout -k 10 200 python bench.py $B --scene-pool 1 2>/dev/null | pick pool1
