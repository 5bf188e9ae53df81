# fp16 prefill perf mode: tolerance tests, then the bench with its side leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --maxfail=10 -k "perf_mode" -s 2>&1 | grep -v "^E  " | tail -25
timeout 900 python bench.py --no-cpu-baseline > $O/perf16_bench.json 2> $O/perf16_bench.err
tail -3 $O/perf16_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/perf16_bench.json").read().strip().splitlines()[-1])
print(d["value"], d.get("prefill_tokens_per_s"), d.get("prefill_tokens_per_s_warm"))
print(json.dumps(d.get("fp16_prefill_mode")))
PY
