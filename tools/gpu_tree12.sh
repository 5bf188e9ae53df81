# 12-wide tree forward of the 8B shape: parity tests that cover the narrow mat-mul, latency, kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_speculative.py -m gpu -q --maxfail=10 -k "mul_mat or real_layer or tree or long_context or speculative" 2>&1 | tail -3
python tools/bench_verify.py Q4_K 2,12,16 2>&1 | tail -1
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) 2>&1 | head -14 | cut -c1-175 | tee $O/r03_tree12_kernel_stats_wav.txt
