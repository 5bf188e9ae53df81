cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 60 python bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 128 --no-kv-f16 --no-graph-path > $O/r04_bench_llama32_1b_q4_0.json 2> $O/r04_bench_1b.err; cut -c1-160 $O/r04_bench_llama32_1b_q4_0.json
timeout 60 python bench.py --preset qwen2-0.5b --wtype Q8_0 --prompt-len 512 --steps 128 --no-kv-f16 --no-graph-path > $O/r04_bench_qwen2_05b_q8_0.json 2> $O/r04_bench_05b.err; cut -c1-160 $O/r04_bench_qwen2_05b_q8_0.json
timeout 60 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "quant and not kquant" 2>&1 | tail -1
