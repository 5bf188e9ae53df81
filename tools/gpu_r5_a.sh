# round 5, call A: the new boundary / time-out tests, the XCD-aware item order of the wide Q4_K mat-mul (parity under each setting, then timing), baseline bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_boundary.py -m gpu -q --maxfail=10 2>&1 | tail -15
for c in 1 2 4; do
  echo "== parity PS_G4K_CBX=$c"; PS_G4K_CBX=$c timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -k "q4k or q5k or wide or headline or prefill" 2>&1 | tail -2
done
for c in 0 1 2 4; do PS_G4K_CBX=$c timeout 300 python tools/prefill_ab.py 2>&1 | tail -1 | sed "s/^/cbx $c: /"; done | tee gpurun_out/r05_cbx.txt
timeout 600 python bench.py --steps 64 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err || tail -3 gpurun_out/r05_bench_a.err
cut -c1-600 gpurun_out/r05_bench_a.json
