# round 3 evidence, core part (after a change that does not touch the other configurations): full GPU suite, default bench + kernel trace + PMC pass
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r03_pytest_gpu.txt 2>&1; tail -4 $O/r03_pytest_gpu.txt
bash tools/gpu_prof_round.sh
cp $O/bench_default.json $O/r03_bench_8b_full.json
