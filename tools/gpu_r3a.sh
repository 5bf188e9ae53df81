# round 3, session A: parity of the new single-launch attention, its timeline, decode rate with / without it, kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $O/r3a_pytest.txt 2>&1; tail -5 $O/r3a_pytest.txt
timeout 300 python tools/gpu_attn_timeline.py 2048 > $O/r3a_attn_timeline.txt 2>&1; cat $O/r3a_attn_timeline.txt | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3a_bench_new.json 2> $O/r3a_bench_new.err; tail -2 $O/r3a_bench_new.err; cut -c1-300 $O/r3a_bench_new.json
PS_HIP_MODE_OR=16 timeout 600 python bench.py --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3a_bench_two_launch.json 2> $O/r3a_bench_two_launch.err; cut -c1-300 $O/r3a_bench_two_launch.json
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt.log 2>&1; tail -1 $O/prof_kt.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) --decode > $O/r3a_decode_kernel_stats.txt 2>&1; head -16 $O/r3a_decode_kernel_stats.txt
