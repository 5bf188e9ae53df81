cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.txt
timeout 600 python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,96,112,120,128 > gpurun_out/verify.json 2> gpurun_out/verify.err; tail -2 gpurun_out/verify.err; cat gpurun_out/verify.json
timeout 500 python tools/bench_speculative.py --steps 48 > gpurun_out/spec_8b_1b.json 2> gpurun_out/spec.err; tail -2 gpurun_out/spec.err; cat gpurun_out/spec_8b_1b.json
timeout 500 python tools/bench_speculative.py --steps 48 --self-draft > gpurun_out/spec_8b_self.json 2>> gpurun_out/spec.err; tail -2 gpurun_out/spec.err; cat gpurun_out/spec_8b_self.json
