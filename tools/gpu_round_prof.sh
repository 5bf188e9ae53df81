# a round's profile set (R=r06 ...): default bench line, kernel trace (eager, decode window + prefill breakdown), PMC pass (FETCH_SIZE, its own run), tree / speculative latency, the other configurations
# (PS_HIP_MODE_OR=1: every model of the profiled process launches eagerly -- rocprofv3 crashes on captured-graph replays)
R=${R:-r06}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; rm -rf $O/prof_pmc
PS_HIP_MODE_OR=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_pmc.log 2>&1; tail -1 $O/prof_pmc.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(find gpurun_out/prof_pmc -name "*counter_collection.csv" | head -1) --json gpurun_out/${R}_pmc_traffic.json > gpurun_out/${R}_pmc_fetch_size_8b_q4k.txt 2>&1; head -8 gpurun_out/${R}_pmc_fetch_size_8b_q4k.txt
cp gpurun_out/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json   # (the bench below reads it: roofline.traffic, with the source hash it was taken on)
rm -rf gpurun_out/prof_pmc
timeout 900 python bench.py > $O/${R}_bench_8b_full.json 2> $O/${R}_bench_8b_full.err; tail -2 $O/${R}_bench_8b_full.err; cut -c1-300 $O/${R}_bench_8b_full.json
cd /tmp; rm -rf $O/prof_kt
PS_HIP_MODE_OR=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 32 --warmup 4 --no-cpu-baseline --no-kv-f16 --no-graph-path --wide-chunk 0 > $O/prof_kt.log 2>&1; tail -1 $O/prof_kt.log | cut -c1-200
cd $GRAFT_REPO_ROOT
DB=$(ls gpurun_out/prof_kt/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/prof_summary.py $DB --decode > gpurun_out/${R}_decode_kernel_stats_8b_q4k.txt 2>&1
python tools/prof_summary.py $DB > gpurun_out/${R}_all_kernel_stats_8b_q4k.txt 2>&1
python tools/prefill_breakdown.py $DB > gpurun_out/${R}_prefill_kernel_breakdown_8b_q4k.txt 2>&1
head -12 gpurun_out/${R}_decode_kernel_stats_8b_q4k.txt; head -14 gpurun_out/${R}_prefill_kernel_breakdown_8b_q4k.txt
rm -rf gpurun_out/prof_kt
python tools/bench_verify.py Q4_K 1,2,4,8,12,16,32,64,128 2>&1 | tail -1 > $O/${R}_tree_forward_latency_8b.json; cut -c1-300 $O/${R}_tree_forward_latency_8b.json
python tools/bench_speculative.py --steps 48 2>&1 | tail -1 > $O/${R}_speculative_8b_1b_draft.json; cut -c1-300 $O/${R}_speculative_8b_1b_draft.json
timeout 600 python bench.py --preset llama-3.2-1b --wtype Q4_0 --prompt-len 512 --steps 128 --n-ctx 1024 > $O/${R}_bench_llama32_1b_q4_0.json 2>/dev/null; cut -c1-200 $O/${R}_bench_llama32_1b_q4_0.json
timeout 600 python bench.py --preset qwen2-0.5b --wtype Q8_0 --prompt-len 512 --steps 128 --n-ctx 1024 > $O/${R}_bench_qwen2_05b_q8_0.json 2>/dev/null; cut -c1-200 $O/${R}_bench_qwen2_05b_q8_0.json
timeout 900 python bench.py --wtype Q4_K_M --no-kv-f16 > $O/${R}_bench_8b_q4_k_m.json 2>/dev/null; cut -c1-200 $O/${R}_bench_8b_q4_k_m.json
timeout 900 python bench.py --wtype Q5_K_M --no-kv-f16 > $O/${R}_bench_8b_q5_k_m.json 2>/dev/null; cut -c1-200 $O/${R}_bench_8b_q5_k_m.json
python tools/bench_speculative.py --steps 96 --truncated-draft 4 --late-scale 0.03 2>&1 | tail -1 > $O/${R}_speculative_8b_late_scaled_target.json; cut -c1-400 $O/${R}_speculative_8b_late_scaled_target.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${R}_bench_8b_driver_line.json 2>/dev/null; cut -c1-200 $O/${R}_bench_8b_driver_line.json
