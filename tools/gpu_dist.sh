cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --steps 32 --warmup 4 --no-cpu-baseline 2>gpurun_out/dist.err | cut -c1-400; tail -3 gpurun_out/dist.err
