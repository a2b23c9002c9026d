set -e
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/xmask3d_amd/miopen_db
wc -l xmask3d_amd/miopen_db/*.txt
XM3D_CL=1 timeout -k 10 1000 python tools/tune_miopen.py 20 10 5 1 > gpurun_out/tune_bf16_cl.log 2>&1 || true
grep "B=" gpurun_out/tune_bf16_cl.log
wc -l xmask3d_amd/miopen_db/*.txt
mkdir -p gpurun_out/miopen_r2 && cp xmask3d_amd/miopen_db/*.txt gpurun_out/miopen_r2/
