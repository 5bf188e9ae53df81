# (PS_HIP_MODE_OR=1: every model of the profiled process launches eagerly -- rocprofv3 crashes on captured-graph replays)
# round profile set: default bench line and kernel trace (eager, decode window); the PMC pass (FETCH_SIZE) is tools/gpu_pmc_r4.sh (side legs off: under counters the full bench does not fit its time limit)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err; cut -c1-400 gpurun_out/bench_default.json
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_kt
PS_HIP_MODE_OR=1 timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls gpurun_out/prof_kt/*.db | head -1) --decode > gpurun_out/r04_decode_kernel_stats_8b_q4k.txt 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof_kt/*.db | head -1) > gpurun_out/r04_all_kernel_stats_8b_q4k.txt 2>&1
head -12 gpurun_out/r04_decode_kernel_stats_8b_q4k.txt
