#!/bin/bash
# scenes per forward A/B on one box.  usage: bash run/ab_spf.sh
set -e -o pipefail
mkdir -p gpurun_out
for n in 4 6 4 6 5; do
python bench.py --no-cpu-baseline --train-steps 0 --fp32-steps 0 --scenes-per-forward $n --steps 24 --warmup 6 > gpurun_out/ab_spf_$n.log 2>&1
grep -h '"value"' gpurun_out/ab_spf_$n.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('spf $n', d['value'])"
done
