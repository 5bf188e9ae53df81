# round 4, first GPU pass: LDS-DMA semantics, the new parity tests, gemv7 (k_gemv7.hip) parity under its variants, launch timings by variant
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 tools/micro/bin/ldsdma > $O/r04_micro_ldsdma.txt 2>&1; cat $O/r04_micro_ldsdma.txt
SUB="tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py"
for cfg in 20 21 22; do
  PS_G4_CFG=$cfg timeout 900 python -m pytest $SUB -m gpu -q -x -k "mul_mat or real_layer or generate or long_cache or golden or one_launch" > $O/r04_pytest_cfg$cfg.txt 2>&1; echo "cfg $cfg: $(tail -1 $O/r04_pytest_cfg$cfg.txt)"
done
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_pytest_gpu.txt 2>&1; tail -4 $O/r04_pytest_gpu.txt
timeout 900 python tools/g4_variants.py 0 20 21 22 23 0 > $O/r04_gemv7_variants.txt 2>&1; cat $O/r04_gemv7_variants.txt
