set -e
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/xmask3d_amd/miopen_db
wc -l xmask3d_amd/miopen_db/*
XM3D_TUNE_DTYPE=fp32 XM3D_CL=1 timeout -k 10 900 python tools/tune_miopen.py 5 > gpurun_out/tune_fp32_cl.log 2>&1 || true
tail -4 gpurun_out/tune_fp32_cl.log
wc -l xmask3d_amd/miopen_db/*
mkdir -p gpurun_out/miopen_fp32 && cp xmask3d_amd/miopen_db/* gpurun_out/miopen_fp32/
