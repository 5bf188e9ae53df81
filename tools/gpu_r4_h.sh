# round 4, eighth GPU pass: N-at-once ggml_v_expf in the soft-max phases: parity, decode + prefill timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "not real_dimensions and not headline" > $O/r04h_pytest.txt 2>&1; tail -1 $O/r04h_pytest.txt
timeout 900 python tools/g4_variants.py 0 0 > $O/r04h_gemv_variants.txt 2>&1; cat $O/r04h_gemv_variants.txt
timeout 600 python tools/prefill_ab.py > $O/r04h_prefill_ab.txt 2>&1; cat $O/r04h_prefill_ab.txt
