# round 4: library without SLP-packed fp32 VALU; the chunk mat-mul's chains as scalar v_fma_f32
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/prefill_ab.py > $O/r04n_scalar_prefill.txt 2>&1; cat $O/r04n_scalar_prefill.txt
timeout 300 python tools/g4_variants.py 0 > $O/r04n_scalar_decode.txt 2>&1; tail -1 $O/r04n_scalar_decode.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "chunk or batched or narrow or prefill or wide or kquant or real_layer" > $O/r04n_pytest.txt 2>&1; tail -2 $O/r04n_pytest.txt
