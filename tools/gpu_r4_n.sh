# round 4: A/B builds of the wide chunk mat-mul
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for v in "" andor; do
  if [ -z "$v" ]; then lib=""; else lib=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_$v.so; fi
  PS_HIP_LIB=$lib timeout 300 python tools/prefill_ab.py 2>&1 | tail -1
done > $O/r04n_andor.txt 2>&1; cat $O/r04n_andor.txt
