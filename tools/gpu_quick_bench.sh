# quick check after a kernel change: parity tests, then short default-workload bench lines (PS_GEMM8_C8=1: 8-column variant)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_speculative.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --steps 32 2>/dev/null > gpurun_out/exp.json; python -c "
import sys, json; d=json.load(open('gpurun_out/exp.json')); print('decode', round(d['value'],1), 'prefill', round(d['prefill_tokens_per_s'],1))"
PS_GEMM8_C8=1 timeout 300 python bench.py --no-cpu-baseline --steps 8 2>/dev/null > gpurun_out/exp2.json; python -c "
import sys, json; d=json.load(open('gpurun_out/exp2.json')); print('C8: decode', round(d['value'],1), 'prefill', round(d['prefill_tokens_per_s'],1))"
