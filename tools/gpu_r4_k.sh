cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_speculative.py tests/test_gpu_model.py -m gpu -q -x > $O/r04k_pytest.txt 2>&1; tail -2 $O/r04k_pytest.txt
timeout 600 python tools/par_exp.py 12 0:0 1:0 > $O/r04k_par_exp.txt 2>&1; cat $O/r04k_par_exp.txt
timeout 600 python tools/par_timeline.py 50 49 48 > $O/r04_par_timeline.txt 2>&1; cat $O/r04_par_timeline.txt
timeout 600 python tools/bench_verify.py Q4_K 1,2,4,8,12,16,24,32 > $O/r04_tree_forward_latency_8b.json 2> $O/r04k_tree.err; cut -c1-400 $O/r04_tree_forward_latency_8b.json; tail -2 $O/r04k_tree.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) 2>&1 | head -24 | cut -c1-175 | tee $O/r04_tree12_kernel_stats.txt
