set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import sys; sys.path.insert(0,'.'); from powerserve_amd import hip; c=hip.Ctx(0); print(c.name())" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -25
