cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
PS_HIP_LIB=$GRAFT_REPO_ROOT/powerserve_amd/lib/libps_hip_mark2.so timeout 300 python tools/g4k_marks.py 52 > $O/r04n_g4k_marks.txt 2>&1; cat $O/r04n_g4k_marks.txt
