cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -4
for v in 1 0; do echo "softmax in its own launch: $v"; if [ $v = 1 ]; then export PS_NO_PV_SOFTMAX=1; else unset PS_NO_PV_SOFTMAX; fi; timeout 300 python tools/prefill_ab.py 2>&1 | tail -1; timeout 300 python tools/bench_verify.py Q4_K 1,8,12,16,32,64 2>&1 | tail -1; done | tee gpurun_out/r06_pv_softmax_ab.txt
