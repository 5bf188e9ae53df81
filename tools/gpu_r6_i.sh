cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "4 0.1" "4 0.03" "8 0.1" "8 0.03" "4 0.01"; do set -- $cfg; timeout 600 python tools/bench_speculative.py --steps 96 --truncated-draft $1 --late-scale $2 2>&1 | tail -1 | cut -c1-700; done | tee gpurun_out/r06_speculative_truncated2.txt
