cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_speculative.py -x -q -m gpu 2>&1 | tail -5
bash tools/gpu_spec.sh
