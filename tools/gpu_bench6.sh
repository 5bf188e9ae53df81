cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench_8b_v3.json 2> gpurun_out/bench_8b_v3.err; tail -3 gpurun_out/bench_8b_v3.err; cat gpurun_out/bench_8b_v3.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v3 -o v3 -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --prompt-len 2048 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_v3.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_v3.log
cd $GRAFT_REPO_ROOT; timeout 300 python tools/gpu_timeline.py 5 1 2 > gpurun_out/timeline.txt 2>&1; tail -2 gpurun_out/timeline.txt
