# prefill profile by chunk: tools/g4k_exp.py under rocprofv3 --kernel-trace, per-kernel averages per 128-token chunk
# usage: gpu_pf_prof.sh [flags] [kernel-substrings]
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_pf
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pf -o pf -- python $GRAFT_REPO_ROOT/tools/g4k_exp.py ${1:-0} > $GRAFT_REPO_ROOT/gpurun_out/pf.log 2>&1
cd $GRAFT_REPO_ROOT
grep "flags" gpurun_out/pf.log | head -1
python tools/prof_by_chunk.py $(find gpurun_out/prof_pf -name "*.db" | head -1) ${2:-attn_scores_mfma,attn_softmax_probs,attn_pv_mfma,rope_append,quantize_tiles,quantize_norm,"gemm4k_kernel<1","gemm4k_kernel<0"} 32 ${3:-16}
rm -rf gpurun_out/prof_pf
