# round 3: full GPU suite + the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/r3e_pytest.txt 2>&1; tail -16 $O/r3e_pytest.txt | cut -c1-200
timeout 900 python bench.py > $O/r3e_bench_default.json 2> $O/r3e_bench_default.err; tail -3 $O/r3e_bench_default.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3e_bench_default.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","prefill_tokens_per_s","prefill_tokens_per_s_warm","roofline","prefill_roofline","cpu_baseline","parity","fp16_kv_mode","graph_path"):
    print(k, json.dumps(d.get(k))[:600])
PY
