# round 4, second GPU pass: gemv7 with register-resident activation operands + gemv4 hedge (parity, timelines, launch timings), own fp16 GEMM, G4K_PAD A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for cfg in 20 40; do
  PS_G4_CFG=$cfg timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "mul_mat_quant or real_layer or test_generate" > $O/r04b_pytest_cfg$cfg.txt 2>&1; echo "cfg $cfg: $(tail -1 $O/r04b_pytest_cfg$cfg.txt)"
done
for cfg in 0 20 21; do
  G4_CFG=$cfg timeout 300 python tools/gpu_timeline.py 5 2 > $O/r04b_timeline_cfg$cfg.txt 2>&1
done
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "fp16_prefill" > $O/r04b_pytest_f16.txt 2>&1; echo "f16: $(tail -1 $O/r04b_pytest_f16.txt)"
timeout 900 python tools/g4_variants.py 0 20 21 22 40 0 > $O/r04b_gemv_variants.txt 2>&1; cat $O/r04b_gemv_variants.txt
timeout 600 python tools/prefill_ab.py > $O/r04b_prefill_ab.txt 2>&1
PS_HIP_LIB=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_pad64.so timeout 600 python tools/prefill_ab.py >> $O/r04b_prefill_ab.txt 2>&1
PS_MODE=32 timeout 600 python tools/prefill_ab.py >> $O/r04b_prefill_ab.txt 2>&1
cat $O/r04b_prefill_ab.txt
