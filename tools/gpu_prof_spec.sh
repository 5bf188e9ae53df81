cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_spec
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_spec -o sp -- python $GRAFT_REPO_ROOT/tools/bench_speculative.py --steps 32 > $GRAFT_REPO_ROOT/gpurun_out/prof_spec.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_spec.log | cut -c1-600
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_spec | head; python tools/prof_summary.py $(ls gpurun_out/prof_spec/*.db | head -1) 2>&1 | head -40
