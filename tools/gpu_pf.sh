cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 128 > gpurun_out/bench_pf.json 2> gpurun_out/bench_pf.err; tail -2 gpurun_out/bench_pf.err; python -c "
import json; d=json.load(open('gpurun_out/bench_pf.json')); print('decode', d['value'], 'prefill', d['prefill_tokens_per_s'], d['roofline']['avg_launch_us'], d['roofline']['all_matvec']['ms_per_token'])"
bash tools/gpu_cfgs.sh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l[:l.rfind('}')+1]) if l.rstrip().endswith('}') else None
        if d: print(d['metric'], d['value'], d['prefill_tokens_per_s'])
"
