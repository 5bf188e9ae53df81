# Next experiment (no GPU time was left for it in round 4): compiler scheduling flags on the single-token kernels, one A/B library each
# (tools/ab_build.py), to be timed by tools/gpu_flag_sweep.sh.  -fno-slp-vectorize was found this way (+1.6 % decode, same bits).  Run on the dev box.
cd "$(dirname "$0")/.."
python tools/ab_build.py relocc   k_gemv4.hip,k_attn.hip -mllvm -amdgpu-schedule-relaxed-occupancy=true
python tools/ab_build.py nopost   k_gemv4.hip,k_attn.hip -mllvm -enable-post-misched=0
python tools/ab_build.py trackers k_gemv4.hip,k_attn.hip -mllvm -amdgpu-use-amdgpu-trackers=1
python tools/ab_build.py nohighrp k_gemv4.hip,k_attn.hip -mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1
python tools/ab_build.py o2       k_gemv4.hip,k_attn.hip -O2
# the flag on every file (round 4 kept it to three files: powerserve_amd/build.py says why) -- with the 8B Q5_K_M bench line as the first thing to run
python tools/ab_build.py noslpall k_quant.hip,k_gemv.hip,k_gemvk.hip,k_gemm4k.hip,k_gemv6.hip,k_ops.hip,perf16.hip -fno-slp-vectorize
