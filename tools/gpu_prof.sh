cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps 4 --warmup 1 --prompt-len 8 --no-cpu-baseline > /dev/null 2>&1   # create the model file once
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v1 -o v1 -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --prompt-len 2048 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_v1.log 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_v1.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_v1 -type f | head; 
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_v1 -name "*kernel_stats.csv" | head -1); head -30 "$f"
