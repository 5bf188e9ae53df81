# round 4, the last GPU seconds: the op / speculative / model tests under the final build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 125 python -m pytest tests/test_gpu_ops.py tests/test_gpu_speculative.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r04_pytest_gpu_final_build.txt 2>&1; tail -3 gpurun_out/r04_pytest_gpu_final_build.txt
