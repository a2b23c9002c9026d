set -e
rm -rf gpurun_out/roof3_stats gpurun_out/roof3_fetch gpurun_out/roof3_write gpurun_out/roof3_sq gpurun_out/roofline_pmc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/roof3_stats -- python3 bench.py --roofline-only > gpurun_out/roof3_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/roof3_fetch -- python3 bench.py --roofline-only > gpurun_out/roof3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/roof3_write -- python3 bench.py --roofline-only > gpurun_out/roof3_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/roof3_sq -- python3 bench.py --roofline-only > gpurun_out/roof3_sq.log 2>&1
find gpurun_out/roof3_stats -name "*kernel_trace.csv" -delete
for k in "k_conv3x3<256, 2, false, 8>" "k_conv3x3<256, 0, false, 8>" "k_conv3x3<256, 2, false, 4>" "k_conv3x3<256, 0, false, 4>" "k_spconv_splitILi6ELi1ELi96ELi4ELi3ELi2ELb1E" "k_spconv_splitILi4ELi1ELi64ELi8ELi3ELi2ELb1E"; do
  echo "## $k"; for d in roof3_fetch roof3_write roof3_sq; do python tools/pmc_summary.py gpurun_out/$d "$k" || true; done
done > gpurun_out/r03_roofline_pmc.txt
for w in 4 8; do for m in 0 2; do
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_conv3x3<256,$m,false,$w>" "k_conv3x3<256, $m, false, $w>" gpurun_out/roof3_fetch gpurun_out/roof3_write || true
done; done
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_spconv_split<4,1,64,8,3,2,true>" "k_spconv_splitILi4ELi1ELi64ELi8ELi3ELi2ELb1E" gpurun_out/roof3_fetch gpurun_out/roof3_write
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_spconv_split<6,1,96,4,3,2,true>" "k_spconv_splitILi6ELi1ELi96ELi4ELi3ELi2ELb1E" gpurun_out/roof3_fetch gpurun_out/roof3_write
cat gpurun_out/r03_roofline_pmc.txt; tail -2 gpurun_out/roof3_stats.log | cut -c1-1500
grep -h "k_conv3x3\|k_spconv_split\|Name" gpurun_out/roof3_stats/*/*kernel_stats.csv | head
