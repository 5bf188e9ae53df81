cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for v in fused two; do
cd /tmp; rm -rf $O/prof_kt
if [ $v = two ]; then export PS_NO_QKV_ATTN=1; fi
PS_HIP_MODE_OR=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_kt.log 2>&1; tail -1 $O/prof_kt.log | cut -c1-200
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/prof_summary.py $DB --decode > gpurun_out/r06_decode_kernel_stats_$v.txt 2>&1
head -12 gpurun_out/r06_decode_kernel_stats_$v.txt
rm -rf gpurun_out/prof_kt
done
