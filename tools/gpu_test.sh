cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
timeout 400 python bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 128 --warmup 8 --n-ctx 1024 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_1b_q4_0.json; python -c "
import json; d=json.load(open('gpurun_out/bench_1b_q4_0.json')); print('1B Q4_0 decode', d['value'], 'prefill', d['prefill_tokens_per_s'])"
