# gemvk: split RoPE fusion + ring shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -25 > $O/r3j_pytest.txt; tail -8 $O/r3j_pytest.txt
quick() { # wtype env
env $2 timeout 400 python bench.py --wtype $1 --steps 128 --warmup 8 --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3j_tmp.json 2> $O/r3j_tmp.err
python - $O/r3j_tmp.json "$1 $2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print("==", sys.argv[2], round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms; prefill", round(d.get("prefill_tokens_per_s"),0), d["roofline"]["kernel"][:44], round(d["roofline"]["frac"],3))
PY
tail -2 $O/r3j_tmp.err | grep -v synthetic
}
for c in 0 1 2 3 4; do quick Q5_K_M PS_GEMVK_CFG=$c; done
for c in 0 2 3 4; do quick Q4_K_M PS_GEMVK_CFG=$c; done
quick Q4_K PS_X=0
