# instruction-cache counters of the decode kernels (separate --pmc passes, kernel trace only): is the start-up of the short mat-vec
# launches (1 - 1.3 us per stage of straight-line code, profiles/r05_gemv_timeline.txt) instruction fetch?
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH InstrFetchLatency SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1)); rm -rf $O/prof_ic$i
  PS_HIP_MODE_OR=1 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_ic$i -o ic -- python $GRAFT_REPO_ROOT/bench.py --eager --prompt-len 128 --steps 12 --warmup 2 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_ic$i.log 2>&1; tail -1 $O/prof_ic$i.log | cut -c1-160
  f=$(find $O/prof_ic$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_generic.py $f gemv4_kernel,attn_decode2,argmax,get_rows
  rm -rf $O/prof_ic$i
done 2>&1 | tee $O/r05_pmc_icache_decode.txt
