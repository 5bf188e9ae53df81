cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
PS_NO_G3=1 timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -x -q -k "generate_matches or e2e or real_layer" 2>&1 | tail -2
for mode in "PS_NO_G3=1" "PS_G3_SMALL=0"; do
cd /tmp; rm -rf $O/prof_kt
env $mode timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 32 --warmup 4 --n-ctx 1024 --eager --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
echo "== $mode"; python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) --decode 2>&1 | head -11 | cut -c1-170
done
