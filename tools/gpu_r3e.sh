# round 3: full GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/r3e_pytest.txt 2>&1; tail -30 $O/r3e_pytest.txt | cut -c1-250
