cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4
python bench.py --preset small-llama --wtype Q4_K --prompt-len 64 --n-ctx 256 --steps 32 --warmup 4 --no-cpu-baseline --force-dist 2>&1 | tail -2 | cut -c1-300
( time python bench.py > gpurun_out/bench_8b_full.json 2> gpurun_out/bench_8b_full.err ) 2>&1 | tail -3; tail -2 gpurun_out/bench_8b_full.err; cut -c1-3000 gpurun_out/bench_8b_full.json
