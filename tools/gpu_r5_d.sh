# round 5, call D: new parity tests (two reference builds, contract library, device arg-max on the op-API path), compiler-flag sweep on the decode kernels, default bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_host.py -m gpu -q --maxfail=5 -s -k "both_reference_builds or contract_build or device_argmax" 2>&1 | grep -v "^$" | tail -25
for v in "" relocc nopost trackers nohighrp o2 noslpall; do
  if [ -z "$v" ]; then lib=""; else lib=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_$v.so; fi
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  echo "== lib '$v'"; PS_HIP_LIB=$lib timeout 120 python tools/g4_variants.py 0 2>&1 | tail -1
done > $O/r05_flag_sweep.txt 2>&1; cat $O/r05_flag_sweep.txt
timeout 900 python bench.py > $O/r05_bench_d.json 2> $O/r05_bench_d.err || tail -3 $O/r05_bench_d.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_d.json").read().strip().splitlines()[-1])
print("decode", d["value"], "prefill", d.get("prefill_tokens_per_s"), d.get("prefill_tokens_per_s_warm"))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic_source"], "all", d["roofline"]["all_matvec"]["frac"])
print("graph_path", d.get("graph_path"))
print("parity", {k: v for k, v in d.get("parity", {}).items() if k != "reference_build"})
print("cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "kind")})
PY
