# prefill attention kernels: parity tests that cover them, then the per-chunk kernel profile (tools/gpu_pf_prof.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_speculative.py -m gpu -q --maxfail=5 -k "prefill or long_context or tree or real_layer or attn or generate or speculative or gqa" 2>&1 | tail -3
bash tools/gpu_pf_prof.sh 0 attn_scores_mfma,attn_softmax_probs,attn_pv_mfma 2>&1 | tee gpurun_out/attn_pf.txt
