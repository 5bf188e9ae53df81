cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_host.py tests/test_gpu_speculative.py -x -q -m gpu 2>&1 | tail -15
