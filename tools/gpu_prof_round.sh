# round profile set: kernel trace (eager, decode window) + separate PMC pass (FETCH_SIZE) + default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err; cut -c1-300 gpurun_out/bench_default.json
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_kt $GRAFT_REPO_ROOT/gpurun_out/prof_pmc
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log | cut -c1-200
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_pmc.log | cut -c1-200
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_pmc | head; find $GRAFT_REPO_ROOT/gpurun_out/prof_pmc -name "*.csv" | head
