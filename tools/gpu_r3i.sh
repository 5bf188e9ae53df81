# gemvk bring-up (Q6_K / Q5_K decode kernel): GPU suite, Q4_K_M / Q5_K_M decode with and without it, kernel traces
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -25 > $O/r3i_pytest.txt; tail -8 $O/r3i_pytest.txt
quick() { # wtype env
env $2 timeout 400 python bench.py --wtype $1 --steps 128 --warmup 8 --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3i_tmp.json 2> $O/r3i_tmp.err
python - $O/r3i_tmp.json "$1 $2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print("==", sys.argv[2], round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms; prefill", round(d.get("prefill_tokens_per_s"),0), d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],3))
PY
tail -3 $O/r3i_tmp.err
}
trace() { # wtype tag env
cd /tmp; rm -rf $O/prof_kt
env $3 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --wtype $1 --prompt-len 512 --steps 32 --warmup 4 --eager --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
echo "== trace $1 $3"; python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) --decode 2>&1 | head -22 | cut -c1-170 | tee $O/r3i_decode_kernel_stats_$2.txt
}
quick Q4_K_M PS_X=0
quick Q4_K_M PS_NO_GEMVK=1
quick Q5_K_M PS_X=0
quick Q5_K_M PS_GEMVK_CFG=1
quick Q4_K_M PS_GEMVK_CFG=1
trace Q4_K_M q4km PS_X=0
trace Q5_K_M q5km PS_X=0
