# round 4, third GPU pass: batch-attention swizzle + G4K_PAD 64 (parity, prefill timing), fp16 GEMM epilogue fix, gemv4 YS default vs old default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "not real_dimensions and not headline" > $O/r04c_pytest_gpu.txt 2>&1; tail -3 $O/r04c_pytest_gpu.txt
timeout 600 python tools/prefill_ab.py > $O/r04c_prefill_ab.txt 2>&1
PS_MODE=32 timeout 600 python tools/prefill_ab.py >> $O/r04c_prefill_ab.txt 2>&1
cat $O/r04c_prefill_ab.txt
timeout 900 python tools/g4_variants.py 0 42 0 42 > $O/r04c_gemv_variants.txt 2>&1; cat $O/r04c_gemv_variants.txt
