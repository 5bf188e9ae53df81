cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "q4k or chunk or wide or batched or golden" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/prefill_ab.py 2>&1 | tail -1 | tee gpurun_out/r06_g4k_ring.txt
