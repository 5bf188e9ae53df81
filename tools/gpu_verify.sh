cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,128 > gpurun_out/verify_gemm4k.json 2> gpurun_out/verify.err; tail -2 gpurun_out/verify.err; cat gpurun_out/verify_gemm4k.json
PS_NO_GEMM4K=1 timeout 600 python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,128 > gpurun_out/verify_nogemm4k.json 2>> gpurun_out/verify.err; tail -2 gpurun_out/verify.err; cat gpurun_out/verify_nogemm4k.json
