cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
sed -i 's#^timeout 900 python bench.py > gpurun_out/bench_default.json.*#echo skip default bench#' tools/gpu_prof_round.sh
bash tools/gpu_prof_round.sh
