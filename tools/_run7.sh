set -o pipefail
for cfg in "tiles 768" "split 128" "split 256" "split 512" "split 768" "split 1536"; do set -- $cfg; echo "== algo $1 wg_target $2"; XM3D_SPCONV_ALGO=$1 XM3D_SPLIT_WG_TARGET=$2 timeout -k 10 120 python tools/prof_3d.py 10 batch 2>&1 | grep -v amdgpu.ids; done
XM3D_SPCONV_ALGO=tiles timeout -k 10 120 python tools/prof_3d.py 10 full 2>&1 | grep -v amdgpu.ids
XM3D_SPCONV_ALGO=split timeout -k 10 120 python tools/prof_3d.py 10 full 2>&1 | grep -v amdgpu.ids
