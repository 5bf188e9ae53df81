# end-of-round refresh: smoke, default bench line (with CPU baseline), kernel trace, PMC pass, side configurations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_prof_round.sh 2>&1 | tail -8
cd $GRAFT_REPO_ROOT
bash tools/gpu_cfgs.sh > /dev/null 2>&1
timeout 500 python bench.py --wtype Q4_K_M --prompt-len 512 --steps 128 --warmup 8 --n-ctx 1024 --no-cpu-baseline > gpurun_out/bench_8b_q4km.json 2> gpurun_out/bench_q4km.err
bash tools/gpu_spec.sh > /dev/null 2>&1
timeout 500 python tools/bench_verify.py > gpurun_out/verify.json 2>/dev/null
ls gpurun_out | head -40
