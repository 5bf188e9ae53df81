cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for f in 0 1 2; do echo "== kv_after $f"; G4_FLAGS=$f TL_KEYS=43 timeout 300 python tools/gpu_attn_timeline.py 2>&1 | tail -20; done | tee gpurun_out/r06_fused_timeline_kvafter.txt
timeout 300 python tools/g4_variants.py 0 100 200 0 100 200 2>&1 | tail -6 | tee gpurun_out/r06_fused_ab.txt
echo "== two launches"; PS_NO_QKV_ATTN=1 timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee -a gpurun_out/r06_fused_ab.txt
