cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" nocons noprod; do
  if [ -z "$v" ]; then unset PS_HIP_LIB; else export PS_HIP_LIB=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_$v.so; fi
  timeout 300 python tools/prefill_ab.py 2>&1 | tail -1
done | tee gpurun_out/r06_g4k_whatif2.txt
