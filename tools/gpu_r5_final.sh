# the round's last pass after a host-side change: the whole GPU suite, the PMC pass (FETCH_SIZE, its own run; the record carries the source hash) and the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/gpu_r5_tests.sh
cd /tmp; rm -rf $O/prof_pmc
PS_HIP_MODE_OR=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_pmc.log 2>&1; tail -1 $O/prof_pmc.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(find gpurun_out/prof_pmc -name "*counter_collection.csv" | head -1) --json gpurun_out/r05_pmc_traffic.json > gpurun_out/r05_pmc_fetch_size_8b_q4k.txt 2>&1; head -8 gpurun_out/r05_pmc_fetch_size_8b_q4k.txt
cp gpurun_out/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
rm -rf gpurun_out/prof_pmc
timeout 900 python bench.py > $O/r05_bench_8b_full.json 2> $O/r05_bench_8b_full.err; tail -2 $O/r05_bench_8b_full.err; cut -c1-300 $O/r05_bench_8b_full.json
