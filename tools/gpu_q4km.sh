cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python bench.py --wtype Q4_K_M --prompt-len 512 --steps 128 --warmup 8 --n-ctx 1024 --no-cpu-baseline > gpurun_out/bench_8b_q4km.json 2> gpurun_out/bench_q4km.err; tail -2 gpurun_out/bench_q4km.err; cut -c1-700 gpurun_out/bench_8b_q4km.json
cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_q4km
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_q4km -o q -- python $GRAFT_REPO_ROOT/bench.py --wtype Q4_K_M --eager --prompt-len 512 --steps 16 --warmup 2 --n-ctx 1024 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_q4km.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_summary.py $(ls gpurun_out/prof_q4km/*.db | head -1) --decode 2>&1 | head -24
