# the fused QKV + attention launch after a change: quick parity, in-kernel timeline (needs the timeline build), rocprofv3 decode kernel stats, decode with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "generate_matches or one_launch or long or fused" 2>&1 | tail -3
TL_KEYS=43 timeout 300 python tools/gpu_attn_timeline.py 2>&1 | tail -20 | tee gpurun_out/r06_fused_timeline.txt
cd /tmp; rm -rf $O/prof_kt
PS_HIP_MODE_OR=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_kt.log 2>&1; tail -1 $O/prof_kt.log | cut -c1-200
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/prof_summary.py $DB --decode > gpurun_out/r06_decode_kernel_stats_fused.txt 2>&1
head -8 gpurun_out/r06_decode_kernel_stats_fused.txt
rm -rf gpurun_out/prof_kt
timeout 300 python tools/g4_variants.py 0 0 2>&1 | tail -2 | tee gpurun_out/r06_fused_ab.txt
echo "== two launches"; PS_NO_QKV_ATTN=1 timeout 300 python tools/g4_variants.py 0 2>&1 | tail -1 | tee -a gpurun_out/r06_fused_ab.txt
