# round 4, fifth GPU pass: activation requests ahead of the weight requests (one early barrier): parity subset, timelines, launch timings A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "mul_mat_quant or real_layer or test_generate" > $O/r04e_pytest.txt 2>&1; tail -1 $O/r04e_pytest.txt
PS_G4_CFG=20 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "mul_mat_quant or real_layer" > $O/r04e_pytest20.txt 2>&1; tail -1 $O/r04e_pytest20.txt
for cfg in 0 20; do
  G4_CFG=$cfg timeout 300 python tools/gpu_timeline.py 5 2 1 > $O/r04e_timeline_cfg$cfg.txt 2>&1
done
timeout 900 python tools/g4_variants.py 0 100 20 21 0 100 > $O/r04e_gemv_variants.txt 2>&1; cat $O/r04e_gemv_variants.txt
