# SQ counters of the prefill kernels (separate --pmc passes, kernel trace only): LDS activity / conflicts, matrix-core busy cycles, wave cycles
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf $O/prof_sq$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_sq$i -o sq -- python $GRAFT_REPO_ROOT/tools/g4k_exp.py 0 > $O/prof_sq$i.log 2>&1; tail -1 $O/prof_sq$i.log | cut -c1-160
  f=$(find $O/prof_sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_generic.py $f gemm4k_kernel,attn_scores_mfma,attn_pv_mfma,attn_softmax_probs,quantize
  rm -rf $O/prof_sq$i
done 2>&1 | tee $O/r03_pmc_sq_prefill.txt
