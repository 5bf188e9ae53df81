# round 6, call A: where a decode layer's time goes between its launches (timeline build), the chain wave's L2 prefetch (what-if sweep), the round's new parity tests, a baseline bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== boundaries (hipGraph replay)"; timeout 300 python tools/gpu_boundaries.py 2>&1 | tail -8 | tee gpurun_out/r06_boundaries.txt
echo "== boundaries (eager)"; TL_MODE=1 timeout 300 python tools/gpu_boundaries.py 2>&1 | tail -8 | tee -a gpurun_out/r06_boundaries.txt
echo "== L2 prefetch sweep"; timeout 600 python tools/g4_variants.py 0 100 200 300 0 100 200 2>&1 | tail -8 | tee gpurun_out/r06_g4_prefetch.txt
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_boundary.py tests/test_gpu_host.py -m gpu -q --maxfail=5 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --maxfail=5 -k "rope or generate_matches" 2>&1 | tail -5
echo "== bench"; timeout 600 python bench.py --steps 64 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err || tail -3 gpurun_out/r06_bench_a.err
cut -c1-700 gpurun_out/r06_bench_a.json
