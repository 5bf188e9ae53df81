cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null > gpurun_out/b.json; python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('8B decode', d['value'], 'prefill', d['prefill_tokens_per_s'])"
