# PMC pass alone (FETCH_SIZE), side legs off so that it fits: HBM bytes per launch of the decode / prefill kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; rm -rf $O/prof_pmc
PS_HIP_MODE_OR=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_pmc.log 2>&1; tail -2 $O/prof_pmc.log | cut -c1-300
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(find gpurun_out/prof_pmc -name "*counter_collection.csv" | head -1) --json gpurun_out/r04_pmc_traffic.json > gpurun_out/r04_pmc_fetch_size_8b_q4k.txt 2>&1; head -12 gpurun_out/r04_pmc_fetch_size_8b_q4k.txt
timeout 300 python tools/gpu_attn_timeline.py > gpurun_out/r04_attention_timeline_raw.txt 2>&1; head -16 gpurun_out/r04_attention_timeline_raw.txt
