# round 4, last GPU minutes: the default bench line and the 8B Q5_K_M line under the final build, the CPU baseline as a child process under a time limit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 170 python bench.py > $O/r04_bench_8b_full.json 2> $O/r04o_full.err; echo "rc $?"; cut -c1-260 $O/r04_bench_8b_full.json
timeout 80 python bench.py --wtype Q5_K_M --no-kv-f16 --no-graph-path > $O/r04_bench_8b_q5_k_m.json 2> $O/r04o_q5km.err; echo "rc $?"; cut -c1-200 $O/r04_bench_8b_q5_k_m.json
