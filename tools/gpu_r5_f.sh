cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --maxfail=3 -x -k "generate_matches_oracle or one_launch" 2>&1 | tail -3
timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee gpurun_out/r05_g4_ss.txt
