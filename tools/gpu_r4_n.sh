# round 4: wave-configuration sweep of the decode mat-vec under the build without SLP-packed fp32
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/g4_variants.py 0 2 6 4 40 41 42 12 13 0 > $O/r04n_variants_noslp.txt 2>&1; cat $O/r04n_variants_noslp.txt
