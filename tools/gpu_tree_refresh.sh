cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,128 2>&1 | tail -1 > $O/r03_tree_forward_latency_8b.json; cut -c1-400 $O/r03_tree_forward_latency_8b.json
python tools/bench_speculative.py --steps 48 2>&1 | tail -1 > $O/r03_speculative_8b_1b_draft.json; cut -c1-300 $O/r03_speculative_8b_1b_draft.json
