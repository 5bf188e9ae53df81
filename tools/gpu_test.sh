cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | cut -c1-200
