set -e
rm -rf gpurun_out/roof4_stats gpurun_out/roof4_fetch gpurun_out/roof4_write gpurun_out/roof4_sq gpurun_out/roofline_pmc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/roof4_stats -- python3 bench.py --roofline-only > gpurun_out/roof4_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/roof4_fetch -- python3 bench.py --roofline-only > gpurun_out/roof4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/roof4_write -- python3 bench.py --roofline-only > gpurun_out/roof4_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/roof4_sq -- python3 bench.py --roofline-only > gpurun_out/roof4_sq.log 2>&1
# sparse-conv instantiations: <NT,NTW,CC,IPC,NG,NPW,PRE,BF,G>; BF = plain-bf16 form, the other the f32-accurate split form
S96B="k_spconv_splitILi6ELi1ELi96ELi4ELi3ELi2ELb1ELb1ELi1E|k_spconv_split<6, 1, 96, 4, 3, 2, true, true, 1>"
S96F="k_spconv_splitILi6ELi1ELi96ELi4ELi3ELi2ELb1ELb0ELi1E|k_spconv_split<6, 1, 96, 4, 3, 2, true, false, 1>"
S64B="k_spconv_splitILi4ELi1ELi64ELi8ELi3ELi2ELb1ELb1ELi1E|k_spconv_split<4, 1, 64, 8, 3, 2, true, true, 1>"
S64F="k_spconv_splitILi4ELi1ELi64ELi8ELi3ELi2ELb1ELb0ELi1E|k_spconv_split<4, 1, 64, 8, 3, 2, true, false, 1>"
find gpurun_out/roof4_stats -name "*kernel_trace.csv" -delete
for k in "k_conv3x3<256, 2, false, 8>" "k_conv3x3<256, 0, false, 8>" "k_conv3x3<256, 2, false, 4>" "k_conv3x3<256, 0, false, 4>" "$S96B" "$S96F" "$S64B" "$S64F"; do
  echo "## $k"; for d in roof4_fetch roof4_write roof4_sq; do python tools/pmc_summary.py gpurun_out/$d "$k" || true; done
done > gpurun_out/r04_roofline_pmc.txt
for w in 4 8; do for m in 0 2; do
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_conv3x3<256,$m,false,$w>" "k_conv3x3<256, $m, false, $w>" gpurun_out/roof4_fetch gpurun_out/roof4_write || true
done; done
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_spconv_split<4,1,64,8,3,2,true,true,1>" "$S64B" gpurun_out/roof4_fetch gpurun_out/roof4_write
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_spconv_split<4,1,64,8,3,2,true,false,1>" "$S64F" gpurun_out/roof4_fetch gpurun_out/roof4_write
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_spconv_split<6,1,96,4,3,2,true,true,1>" "$S96B" gpurun_out/roof4_fetch gpurun_out/roof4_write
python tools/pmc_summary.py --json gpurun_out/roofline_pmc.json "k_spconv_split<6,1,96,4,3,2,true,false,1>" "$S96F" gpurun_out/roof4_fetch gpurun_out/roof4_write
cat gpurun_out/r04_roofline_pmc.txt; tail -2 gpurun_out/roof4_stats.log | cut -c1-1500
grep -h "k_conv3x3\|k_spconv_split\|Name" gpurun_out/roof4_stats/*/*kernel_stats.csv | head
