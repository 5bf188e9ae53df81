# round 4: attention with 512-thread work-groups (A/B), wave soft-max back on per-value exp (prefill)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
PS_ATTN_NT=512 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "long_cache or one_launch or real_layer" > $O/r04i_pytest_nt512.txt 2>&1; tail -1 $O/r04i_pytest_nt512.txt
timeout 600 python tools/g4_variants.py 0 > $O/r04i_variants_nt1024.txt 2>&1; cat $O/r04i_variants_nt1024.txt
PS_ATTN_NT=512 timeout 600 python tools/g4_variants.py 0 > $O/r04i_variants_nt512.txt 2>&1; cat $O/r04i_variants_nt512.txt
timeout 600 python tools/prefill_ab.py > $O/r04i_prefill_ab.txt 2>&1; cat $O/r04i_prefill_ab.txt
