cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/f16_flags.py > $O/r04l_f16_flags.txt 2>&1; cat $O/r04l_f16_flags.txt
