cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench_8b_v3.json 2> gpurun_out/bench_8b_v3.err; tail -2 gpurun_out/bench_8b_v3.err; python -c "
import json; d=json.load(open('gpurun_out/bench_8b_v3.json')); print(d['first_ids'], d['value'])"
TL_KEYS='21 22' timeout 300 python tools/gpu_timeline.py 21 22 > gpurun_out/timeline.txt 2>&1
