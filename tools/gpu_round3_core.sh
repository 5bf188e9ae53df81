# round 3 evidence, core part (after a change that does not touch the other configurations): full GPU suite, default bench + kernel trace + PMC pass
# (MIXED=1: also the two mixed-type benches, whose batched Q5_K / Q6_K mat-muls share k_gemm4k.hip)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r03_pytest_gpu.txt 2>&1; tail -4 $O/r03_pytest_gpu.txt
bash tools/gpu_prof_round.sh
cp $O/bench_default.json $O/r03_bench_8b_full.json
if [ "${MIXED:-0}" = "1" ]; then
for wt in Q4_K_M Q5_K_M; do
  n=$(echo $wt | tr 'A-Z' 'a-z')
  timeout 900 python bench.py --wtype $wt --no-kv-f16 --no-graph-path > $O/r03_bench_8b_$n.json 2> $O/r03_bench_8b_$n.err; cut -c1-160 $O/r03_bench_8b_$n.json
done
fi
