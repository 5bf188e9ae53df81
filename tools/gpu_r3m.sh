# after the wave-autonomous narrow mat-mul: full GPU suite, tree-forward latency by width, speculative iteration cost, 12-wide kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -8 | tee $O/r3m_pytest.txt
python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,128 2>&1 | tail -1 | tee $O/r03_tree_forward_latency_8b.json
python tools/bench_verify.py Q4_K_M 1,12 2>&1 | tail -1
python tools/bench_speculative.py --steps 48 2>&1 | tail -1 | tee $O/r03_speculative_8b_1b_draft.json; python tools/bench_speculative.py --steps 48 --self-draft 2>&1 | tail -1 | tee $O/r03_speculative_8b_self_draft.json
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) 2>&1 | head -20 | cut -c1-175 | tee $O/r03_tree12_kernel_stats_wav.txt
