# where a 12-wide tree forward spends its time (kernel trace of tools/bench_verify.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
python tools/bench_verify.py Q4_K 1,2,4,8,12,16 2>&1 | tail -1 | tee $O/r3k_verify.json
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) 2>&1 | head -40 | cut -c1-175 | tee $O/r3k_tree12_kernel_stats.txt
