cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/gpu_timeline.py ${TL_KEYS:-5} > gpurun_out/timeline.txt 2>&1; tail -2 gpurun_out/timeline.txt
