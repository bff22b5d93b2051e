set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2u/pmc
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r2u/pmc/$tag -- python3 $R/tools/bench_conv.py 3 "batched L0 luma" > $R/gpurun_out/r2u/pmc/$tag.log 2>&1
  f=$(find $R/gpurun_out/r2u/pmc/$tag -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep "conv3x3s1_wave_kernel<7, 2>" $f) > $R/gpurun_out/r2u/pmc/pmc_$tag.csv
  k=$(find $R/gpurun_out/r2u/pmc/$tag -name "*kernel_trace.csv" | head -1)
  (head -1 $k; grep "conv3x3s1_wave_kernel<7, 2>" $k) > $R/gpurun_out/r2u/pmc/trace_$tag.csv
  rm -rf $R/gpurun_out/r2u/pmc/$tag
  wc -l $R/gpurun_out/r2u/pmc/pmc_$tag.csv
done
