cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/bench_pf.json 2> gpurun_out/bench_pf.err; tail -2 gpurun_out/bench_pf.err; python -c "
import json; d=json.load(open('gpurun_out/bench_pf.json')); print('decode', d['value'], 'prefill', d['prefill_tokens_per_s'])"
timeout 500 python tools/bench_verify.py Q4_K 8,12,32 2>/dev/null
