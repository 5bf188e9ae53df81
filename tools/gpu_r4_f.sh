# round 4, sixth GPU pass: paired-tile quantizer in the gemv4 prologue: parity, timelines, launch timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q -x > $O/r04f_pytest.txt 2>&1; tail -1 $O/r04f_pytest.txt
G4_CFG=0 timeout 300 python tools/gpu_timeline.py 5 2 1 > $O/r04f_timeline_cfg0.txt 2>&1
timeout 900 python tools/g4_variants.py 0 42 0 > $O/r04f_gemv_variants.txt 2>&1; cat $O/r04f_gemv_variants.txt
