# round 4: nibble operands with v_and_or (high nibbles in place against s / 16) in the chunk mat-mul's producers and the wave-per-tile kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_speculative.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/r04n_pytest.txt 2>&1; tail -2 $O/r04n_pytest.txt
timeout 600 python tools/prefill_ab.py > $O/r04n_prefill_ab.txt 2>&1; cat $O/r04n_prefill_ab.txt
timeout 600 python tools/bench_verify.py Q4_K 2,8,12,16 > $O/r04n_tree.txt 2> $O/r04n_tree.err; cut -c1-300 $O/r04n_tree.txt
