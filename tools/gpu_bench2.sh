cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench_8b_v2.json 2> gpurun_out/bench_8b_v2.err; tail -3 gpurun_out/bench_8b_v2.err; cat gpurun_out/bench_8b_v2.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v2 -o v2 -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --prompt-len 2048 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_v2.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_v2.log
