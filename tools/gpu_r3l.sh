# wave-autonomous narrow mat-mul: parity tests, tree-forward latency by width for the ring configurations, kernel trace at 12 wide
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_speculative.py tests/test_gpu_golden.py -m gpu -q --maxfail=10 2>&1 | tail -12
for c in 0 2; do echo "== cfg $c"; PS_GEMM4K_WAV_CFG=$c python tools/bench_verify.py Q4_K 2,4,8,12,16 2>&1 | tail -1; done
echo "== old"; PS_NO_GEMM4K_WAV=1 python tools/bench_verify.py Q4_K 2,12,16 2>&1 | tail -1
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) 2>&1 | head -16 | cut -c1-175 | tee $O/r3l_tree12_kernel_stats.txt
