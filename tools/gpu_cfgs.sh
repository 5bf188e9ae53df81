cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 128 --warmup 8 --n-ctx 1024 --no-cpu-baseline > gpurun_out/bench_1b_q4_0.json 2> gpurun_out/bench_1b.err; tail -2 gpurun_out/bench_1b.err; cut -c1-900 gpurun_out/bench_1b_q4_0.json
timeout 400 python bench.py --preset qwen2-0.5b --wtype Q8_0 --prompt-len 32 --steps 32 --warmup 4 --n-ctx 256 --no-cpu-baseline > gpurun_out/bench_05b_q8_0.json 2> gpurun_out/bench_05b.err; tail -2 gpurun_out/bench_05b.err; cut -c1-900 gpurun_out/bench_05b_q8_0.json
