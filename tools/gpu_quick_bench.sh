cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --steps 128 2>/dev/null > gpurun_out/exp.json; python -c "
import sys, json; d=json.load(open('gpurun_out/exp.json')); print('decode', round(d['value'],1), 'gate/up us', round(d['roofline']['avg_launch_us'],2), 'all', d['roofline']['all_matvec']['ms_per_token'])"
TL_KEYS='5 1 2' bash tools/gpu_tl.sh > /dev/null 2>&1; grep -E "^key" gpurun_out/timeline.txt | cut -c1-140
