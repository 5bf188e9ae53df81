cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_speculative.py -x -q -m gpu 2>&1 | tail -3
timeout 500 python tools/bench_verify.py > gpurun_out/verify.json 2> gpurun_out/verify.err; tail -2 gpurun_out/verify.err; cat gpurun_out/verify.json
