cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_pf
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pf -o pf -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_pf.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_pf.log | cut -c1-400
