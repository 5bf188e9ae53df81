# round 3: attention iteration loop (model tests, timeline, decode rate, kernel trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_speculative.py tests/test_gpu_host.py -m gpu -x -q > $O/r3c_pytest.txt 2>&1; tail -3 $O/r3c_pytest.txt
timeout 300 python tools/gpu_attn_timeline.py 2048 2>&1 | head -14 > $O/r3c_attn_timeline.txt; cat $O/r3c_attn_timeline.txt | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3c_bench_new.json 2> $O/r3c_bench_new.err; tail -2 $O/r3c_bench_new.err | cut -c1-300; cut -c1-200 $O/r3c_bench_new.json
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) --decode > $O/r3c_decode_kernel_stats.txt 2>&1; head -10 $O/r3c_decode_kernel_stats.txt
