# gemvb, second pass: op + model tests, 1B / 0.5B decode, kernel traces (0.5B also with the old kernels for a per-launch comparison)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_speculative.py -m gpu -q --maxfail=10 2>&1 | tail -5
trace() { # preset wtype tag [env]
cd /tmp; rm -rf $O/prof_kt
env $4 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --preset $1 --wtype $2 --prompt-len 512 --steps 32 --warmup 4 --n-ctx 1024 --eager --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
echo "== trace $1 $2 $4"; python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) --decode 2>&1 | head -14 | cut -c1-170 | tee $O/r3h_decode_kernel_stats_$3.txt
}
quick() { # preset wtype env
env $3 timeout 300 python bench.py --preset $1 --wtype $2 --prompt-len 512 --steps 128 --warmup 8 --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/r3h_tmp.json 2> $O/r3h_tmp.err
python - $O/r3h_tmp.json "$1 $2 $3" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print("==", sys.argv[2], round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms; prefill", round(d.get("prefill_tokens_per_s"),0), d["roofline"]["kernel"][:30], round(d["roofline"]["frac"],3))
PY
}
quick llama-3.2-1b Q4_0 PS_X=0
quick qwen2-0.5b Q8_0 PS_X=0
quick qwen2-0.5b Q8_0 PS_NO_GEMVB=1
trace llama-3.2-1b Q4_0 1b PS_X=0
trace qwen2-0.5b Q8_0 05b PS_X=0
trace qwen2-0.5b Q8_0 05b_old PS_NO_GEMVB=1
