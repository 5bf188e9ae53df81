# round 4, seventh GPU pass: pair quantizer in gemvk (mixed-type models), three chunks in flight with register-resident operands (cfg 41)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q -x > $O/r04g_pytest.txt 2>&1; tail -1 $O/r04g_pytest.txt
timeout 900 python tools/g4_variants.py 0 41 40 0 > $O/r04g_gemv_variants.txt 2>&1; cat $O/r04g_gemv_variants.txt
