# round 3, gemvb bring-up: GPU suite, then the 1B Q4_0 / 0.5B Q8_0 decode with and without the new kernel (+ kernel traces)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -25 > $O/r3g_pytest.txt; tail -3 $O/r3g_pytest.txt
for cfg in "llama-3.2-1b Q4_0" "qwen2-0.5b Q8_0"; do
set -- $cfg
for mode in "PS_X=0" "PS_NO_GEMVB=1"; do
env $mode timeout 300 python bench.py --preset $1 --wtype $2 --prompt-len 512 --steps 128 --warmup 8 --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3g_$1_$mode.json 2> $O/r3g_$1_$mode.err
echo "== $1 $2 $mode"; python - $O/r3g_$1_$mode.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["parity"]["logits_bit_equal"], d["roofline"]["kernel"][:40], d["roofline"]["frac"], d.get("prefill_tokens_per_s"))
PY
done
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --preset $1 --wtype $2 --prompt-len 512 --steps 32 --warmup 4 --n-ctx 1024 --eager --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) --decode 2>&1 | head -14 | cut -c1-170 | tee $O/r3g_decode_kernel_stats_$1.txt
done
