cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | cut -c1-160
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --eager --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | cut -c1-160
done
HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/gpu_timeline.py 5 > gpurun_out/timeline_devkarg.txt 2>&1
