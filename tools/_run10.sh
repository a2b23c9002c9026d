set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_spconv.py tests/test_seams.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r2_t5.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r2_t5.log; tail -8 gpurun_out/r2_t5.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python tools/spconv_bench.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_spconv_bench5.log
timeout -k 10 120 python tools/prof_3d.py 10 batch 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/prof_3d.py 10 full 2>&1 | grep -v amdgpu.ids
