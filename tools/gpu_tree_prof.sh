# per-kernel time of a 12-wide tree forward of the 8B shape (rocprofv3 kernel trace): tools/gpu_tree_prof.sh [tag]; env is passed through (A/B switches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-tree12}
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o t -- python tools/bench_verify.py Q4_K ${WIDTHS:-12} > gpurun_out/${TAG}_bench.txt 2>&1
tail -1 gpurun_out/${TAG}_bench.txt
python tools/prof_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -24 gpurun_out/${TAG}_kernel_stats.txt
