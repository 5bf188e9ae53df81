# GPU side of tools/flag_sweep_build.sh: the decode check per library (ids must stay the same), then the Q5_K_M line under the all-files build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for v in "" relocc nopost trackers nohighrp o2 noslpall; do
  if [ -z "$v" ]; then lib=""; else lib=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_$v.so; fi
  echo "== lib '$v'"; PS_HIP_LIB=$lib timeout 120 python tools/g4_variants.py 0 2>&1 | tail -1
done > $O/flag_sweep.txt 2>&1; cat $O/flag_sweep.txt
PS_HIP_LIB=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_noslpall.so timeout 120 python bench.py --wtype Q5_K_M --no-kv-f16 --no-graph-path > $O/flag_sweep_q5km_noslpall.json 2> $O/flag_sweep_q5km.err; echo "rc $?"; cut -c1-200 $O/flag_sweep_q5km_noslpall.json
