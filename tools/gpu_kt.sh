cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_summary.py $(ls gpurun_out/prof_kt/*.db | head -1) --decode 2>&1 | head -20
