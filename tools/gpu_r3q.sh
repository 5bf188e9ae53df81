cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TL_BS=12 python tools/gpu_g4k_timeline.py 50 49 2>&1 | cut -c1-400
