# round 6, call B: the fused QKV + attention launch -- parity (model / fullsize / boundary / host tests), then decode with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_host.py tests/test_gpu_golden.py -m gpu -q --maxfail=3 2>&1 | tail -8
echo "== fused"; timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee gpurun_out/r06_fused_ab.txt
echo "== two launches"; PS_NO_QKV_ATTN=1 timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee -a gpurun_out/r06_fused_ab.txt
