cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for m in 0 1; do
PS_HIP_MODE_OR=$m timeout 600 python bench.py --no-cpu-baseline --no-kv-f16 --steps 64 > $O/r3o_bench.json 2> $O/r3o_bench.err
python - $O/r3o_bench.json $m <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); g=d["graph_path"]; print("mode_or", sys.argv[2], round(d["value"],1), {k:g[k] for k in ("prefill_tokens_per_s","decode_tokens_per_s","decode_tokens_per_s_after_capture","steps","ids_equal_direct")})
PY
done
