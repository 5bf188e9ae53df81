cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_host.py -m gpu -q --maxfail=10 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-graph-path --steps 128 > $O/r3r.json 2> $O/r3r.err
python - $O/r3r.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(round(d["value"],1), d["fp16_kv_mode"])
PY
timeout 600 python bench.py --wtype Q5_K_M --no-cpu-baseline --no-graph-path --no-kv-f16 --steps 128 2>/dev/null | cut -c1-140
