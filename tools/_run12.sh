set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/roof2_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/roof2_stats -- python3 bench.py --roofline-only > gpurun_out/r2_roof_stats.log 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/roof2_fetch -- python3 bench.py --roofline-only > gpurun_out/r2_roof_fetch.log 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/roof2_write -- python3 bench.py --roofline-only > gpurun_out/r2_roof_write.log 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/roof2_sq -- python3 bench.py --roofline-only > gpurun_out/r2_roof_sq.log 2>&1; echo "rc=$?"
python3 tools/kstats.py gpurun_out/roof2_stats 8 | tee gpurun_out/r02_roofline_spconv_kernel_stats.txt
find gpurun_out/roof2_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_roofline_spconv_kernel_stats.csv
for d in fetch write sq; do python3 tools/pmc_summary.py gpurun_out/roof2_$d k_spconv_split; done | tee gpurun_out/r02_roofline_spconv_pmc_raw.txt
find gpurun_out/roof2_stats -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/r2_roof_stats.log
