cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "f16 or fp16" > $O/r04l_pytest.txt 2>&1; tail -2 $O/r04l_pytest.txt
timeout 300 python tools/f16_gemm_bench.py 1,2,3,0 2048,512 > $O/r04l_f16_gemm_final.txt 2>&1; cat $O/r04l_f16_gemm_final.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-graph-path --wide-chunk 0 > $O/r04l_bench.json 2> $O/r04l_bench.err; tail -2 $O/r04l_bench.err | cut -c1-300
python - <<'PY'
import json
b=json.loads(open('/root/repo/gpurun_out/r04l_bench.json').read().strip().splitlines()[-1])
print("decode", b["value"], "prefill", b["prefill_tokens_per_s"], b["prefill_tokens_per_s_warm"])
print("fp16_prefill_mode", json.dumps(b.get("fp16_prefill_mode"))[:900])
PY
