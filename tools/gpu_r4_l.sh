# round 4: the fp16 perf mode's GEMMs: race screen of every kernel variant, then us / TFLOP/s / error by variant and shape (profiles/r04_f16_gemm.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/f16_gemm_screen.py 1,2,3 6 > $O/r04l_f16_screen.txt 2>&1; tail -3 $O/r04l_f16_screen.txt
timeout 300 python tools/f16_gemm_bench.py 1,2,3,0 2048,512 > $O/r04l_f16_gemm_final.txt 2>&1; cat $O/r04l_f16_gemm_final.txt
