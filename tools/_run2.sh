set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_spconv.py -x -q -m gpu > gpurun_out/r2_t4.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r2_t4.log; tail -3 gpurun_out/r2_t4.log
if [ $rc -ne 0 ]; then exit 1; fi
rm -f gpurun_out/r2_spconv_bench4.log
for v in 0 2 3 4; do echo "== variant $v" >> gpurun_out/r2_spconv_bench4.log; XM3D_SPLIT_VARIANT=$v timeout -k 10 200 python tools/spconv_bench.py 20 >> gpurun_out/r2_spconv_bench4.log 2>&1 || exit 1; done
grep -E "variant|96-> 96|128-> 96|64-> 64|128->128|192->128|32-> 32|384|256" gpurun_out/r2_spconv_bench4.log
