set -o pipefail
for v in 0 2; do echo "== variant $v"; XM3D_SPLIT_VARIANT=$v timeout -k 10 200 python tools/spconv_bench.py 20 2>&1 | grep -E "96-> 96|128-> 96"; done | tee gpurun_out/r2_spconv_bench6.log
timeout -k 10 120 python tools/prof_3d.py 10 batch 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/prof_3d.py 10 full 2>&1 | grep -v amdgpu.ids
