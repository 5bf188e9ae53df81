# round 4: narrow-batch mat-mul with the K walk off the waves (gemm4k_par_kernel): parity, tree forward A/B, kernel trace of 12-wide forwards
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_speculative.py tests/test_gpu_model.py -m gpu -q -x > $O/r04j_pytest.txt 2>&1; tail -3 $O/r04j_pytest.txt
timeout 600 python tools/bench_verify.py Q4_K 2,8,12,16 0,1,4,8 > $O/r04j_tree_par.txt 2> $O/r04j_tree.err; cut -c1-300 $O/r04j_tree_par.txt; tail -2 $O/r04j_tree.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) 2>&1 | head -24 | cut -c1-175 | tee $O/r04_tree12_kernel_stats.txt
