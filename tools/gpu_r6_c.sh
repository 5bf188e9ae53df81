# round 6, call C: fused QKV + attention v2 (cache waves): quick parity, timeline, decode A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "generate_matches or one_launch or long or hidden or kv" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -5
TL_KEYS=43 timeout 300 python tools/gpu_attn_timeline.py 2>&1 | tail -22 | tee gpurun_out/r06_fused_timeline.txt
echo "== fused"; timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee gpurun_out/r06_fused_ab.txt
echo "== two launches"; PS_NO_QKV_ATTN=1 timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee -a gpurun_out/r06_fused_ab.txt
