B="--no-cpu-baseline --train-steps 0 --fp32-steps 0"
for cfg in "4 20 5" "8 64 16" "8 32 8" "5 40 10"; do
set -- $cfg
timeout -k 10 300 python bench.py $B --scene-pool $1 --steps $2 --warmup $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pool $1 steps $2 warmup $3:', round(d['value'],2), round(d['ms_per_step'],2))"
done
