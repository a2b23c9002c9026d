set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r2b
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2b -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --train-steps 0 --fp32-steps 0 > gpurun_out/r2_prof_b.log 2>&1; echo "rc=$?" >> gpurun_out/r2_prof_b.log
tail -2 gpurun_out/r2_prof_b.log | cut -c1-300
python3 tools/ktrace_window.py gpurun_out/prof_r2b 60 > gpurun_out/r02_bench_timed_window.txt 2>&1; head -5 gpurun_out/r02_bench_timed_window.txt
python3 tools/ktrace_post.py gpurun_out/prof_r2b 45 > gpurun_out/r02_bench_postproc_section.txt 2>&1; cat gpurun_out/r02_bench_postproc_section.txt
find gpurun_out/prof_r2b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_kernel_stats.csv
find gpurun_out/prof_r2b -name "*kernel_trace.csv" -delete
