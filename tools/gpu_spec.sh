cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python tools/bench_speculative.py --steps 48 > gpurun_out/spec_8b_1b.json 2> gpurun_out/spec.err; tail -2 gpurun_out/spec.err; cat gpurun_out/spec_8b_1b.json
timeout 500 python tools/bench_speculative.py --steps 48 --self-draft > gpurun_out/spec_8b_self.json 2>> gpurun_out/spec.err; tail -2 gpurun_out/spec.err; cat gpurun_out/spec_8b_self.json
