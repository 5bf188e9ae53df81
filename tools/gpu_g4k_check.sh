# batched Q4_K / Q5_K mat-mul: the parity tests that cover it, then the prefill timer (tools/g4k_exp.py) and the bench's prefill
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q --maxfail=5 -k "mul_mat or prefill or real_layer or golden or generate or wide" 2>&1 | tail -3
python tools/g4k_exp.py 0 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 --steps 32 > gpurun_out/g4k_bench.json 2> gpurun_out/g4k_bench.err || tail -3 gpurun_out/g4k_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/g4k_bench.json").read().strip().splitlines()[-1])
print(d["value"], d.get("prefill_tokens_per_s"), d.get("prefill_tokens_per_s_warm"), d["prefill_roofline"]["dominant_launch"])
PY
