cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/gpu_attn_timeline.py > $O/r04n_attn_timeline.txt 2>&1; head -16 $O/r04n_attn_timeline.txt
