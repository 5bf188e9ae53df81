cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_boundary.py -m gpu -q --maxfail=5 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-kv-f16 --wide-chunk 0 --steps 64 > gpurun_out/r05_bench_e.json 2> gpurun_out/r05_bench_e.err || tail -3 gpurun_out/r05_bench_e.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_e.json").read().strip().splitlines()[-1])
print("decode", d["value"], "prefill", d.get("prefill_tokens_per_s"), d.get("prefill_tokens_per_s_warm"))
g = d.get("graph_path"); g.pop("what", None); print("graph_path", g)
PY
