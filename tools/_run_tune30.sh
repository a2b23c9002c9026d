set -e
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/xmask3d_amd/miopen_db
XM3D_CL=1 timeout -k 10 700 python tools/tune_miopen.py 30 > gpurun_out/tune_bf16_cl30.log 2>&1 || true
grep "B=" gpurun_out/tune_bf16_cl30.log
wc -l xmask3d_amd/miopen_db/*.txt
mkdir -p gpurun_out/miopen_r2b && cp xmask3d_amd/miopen_db/*.txt gpurun_out/miopen_r2b/
B="--steps 24 --warmup 6 --no-cpu-baseline --train-steps 0 --fp32-steps 0"
for spf in 4 6; do
timeout -k 10 200 python bench.py $B --scenes-per-forward $spf --scene-pool 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spf $spf', d['value'], d['ms_per_step'], d['roofline_dense_stage']['ms_per_view'])"
done
