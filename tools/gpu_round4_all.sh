# round 4 evidence: full GPU suite, default bench + kernel trace + PMC pass, the other configurations, tree / speculative latency
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_pytest_gpu.txt 2>&1; tail -4 $O/r04_pytest_gpu.txt
bash tools/gpu_prof_round4.sh
cp $O/bench_default.json $O/r04_bench_8b_full.json
if [ "${MORE:-1}" = "1" ]; then
timeout 600 python bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 128 --no-kv-f16 --no-graph-path > $O/r04_bench_llama32_1b_q4_0.json 2> $O/r04_bench_1b.err; cut -c1-200 $O/r04_bench_llama32_1b_q4_0.json
timeout 600 python bench.py --preset qwen2-0.5b --wtype Q8_0 --prompt-len 512 --steps 128 --no-kv-f16 --no-graph-path > $O/r04_bench_qwen2_05b_q8_0.json 2> $O/r04_bench_05b.err; cut -c1-200 $O/r04_bench_qwen2_05b_q8_0.json
for wt in Q4_K_M Q5_K_M; do
  n=$(echo $wt | tr 'A-Z' 'a-z')
  timeout 900 python bench.py --wtype $wt --no-kv-f16 --no-graph-path > $O/r04_bench_8b_$n.json 2> $O/r04_bench_8b_$n.err; cut -c1-200 $O/r04_bench_8b_$n.json
done
timeout 600 python tools/bench_verify.py > $O/r04_tree_forward_latency_8b.json 2> $O/r04_tree.err; cut -c1-300 $O/r04_tree_forward_latency_8b.json
timeout 900 python tools/bench_speculative.py > $O/r04_speculative_8b_1b_draft.json 2> $O/r04_spec.err; cut -c1-300 $O/r04_speculative_8b_1b_draft.json
fi
