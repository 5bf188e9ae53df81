cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
df -h /tmp | tail -1; nproc; free -g | head -2 | tail -1
python bench.py --preset small-llama --wtype Q4_K --prompt-len 64 --n-ctx 256 --steps 32 --warmup 4 --cpu-steps 4 2>&1 | tail -3
time python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench_8b_v1.json 2> gpurun_out/bench_8b_v1.err; tail -3 gpurun_out/bench_8b_v1.err; cat gpurun_out/bench_8b_v1.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v1 -o v1 -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --prompt-len 256 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_v1.log 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_v1.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_v1 | head -20
