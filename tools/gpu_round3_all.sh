# round 3 evidence set: full GPU suite, default bench + kernel trace + PMC pass, the other configurations, attention timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r03_pytest_gpu.txt 2>&1; tail -4 $O/r03_pytest_gpu.txt
bash tools/gpu_prof_round.sh
cp $O/bench_default.json $O/r03_bench_8b_full.json
for cfg in "llama-3.2-1b Q4_0 r03_bench_llama32_1b_q4_0" "qwen2-0.5b Q8_0 r03_bench_qwen2_05b_q8_0"; do
  set -- $cfg
  timeout 600 python bench.py --preset $1 --wtype $2 --prompt-len 512 --steps 128 --n-ctx 1024 --no-kv-f16 --no-graph-path > $O/$3.json 2> $O/$3.err; cut -c1-160 $O/$3.json
done
for wt in Q4_K_M Q5_K_M; do
  n=$(echo $wt | tr 'A-Z' 'a-z')
  timeout 900 python bench.py --wtype $wt --no-kv-f16 --no-graph-path > $O/r03_bench_8b_$n.json 2> $O/r03_bench_8b_$n.err; cut -c1-160 $O/r03_bench_8b_$n.json
done
timeout 300 python tools/gpu_attn_timeline.py 2048 > $O/r03_attention_timeline_raw.txt 2>&1; head -14 $O/r03_attention_timeline_raw.txt | cut -c1-200
cd /tmp; rm -rf $O/prof_kt1b
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt1b -o kt -- python $GRAFT_REPO_ROOT/bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 32 --warmup 4 --n-ctx 1024 --eager --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_kt1b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt1b/*.db | head -1) --decode > $O/r03_decode_kernel_stats_1b_q4_0.txt 2>&1; head -10 $O/r03_decode_kernel_stats_1b_q4_0.txt | cut -c1-170

for wt in Q4_K_M Q5_K_M; do
  n=$(echo $wt | tr 'A-Z' 'a-z')
  cd /tmp; rm -rf $O/prof_ktm
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ktm -o kt -- python $GRAFT_REPO_ROOT/bench.py --wtype $wt --prompt-len 512 --steps 32 --warmup 4 --eager --no-cpu-baseline --no-kv-f16 --no-graph-path > $O/prof_ktm.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/prof_summary.py $(ls $O/prof_ktm/*.db | head -1) --decode > $O/r03_decode_kernel_stats_8b_$n.txt 2>&1
done
python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,128 2>&1 | tail -1 > $O/r03_tree_forward_latency_8b.json; cut -c1-300 $O/r03_tree_forward_latency_8b.json
python tools/bench_speculative.py --steps 48 2>&1 | tail -1 > $O/r03_speculative_8b_1b_draft.json
python tools/bench_speculative.py --steps 48 --self-draft 2>&1 | tail -1 > $O/r03_speculative_8b_self_draft.json
cd /tmp; rm -rf $O/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 12 > $O/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls $O/prof_kt/*.db | head -1) > $O/r03_tree12_kernel_stats_wav.txt 2>&1
