set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r2a
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2a -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --train-steps 0 --fp32-steps 0 > gpurun_out/r2_prof_a.log 2>&1; echo "rc=$?" >> gpurun_out/r2_prof_a.log
tail -3 gpurun_out/r2_prof_a.log
python3 tools/ktrace_window.py gpurun_out/prof_r2a 45 > gpurun_out/r02_bench_v1_timed_window.txt 2>&1; head -50 gpurun_out/r02_bench_v1_timed_window.txt
find gpurun_out/prof_r2a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_v1_kernel_stats.csv
find gpurun_out/prof_r2a -name "*kernel_trace.csv" -delete
