# (build the timeline library first, in the dev container: python -m powerserve_amd.build --timeline -- it travels with the snapshot)
# round 5: in-kernel timelines of the decode mat-vec families (gate/up, QKV, O / down) and of the single-token attention on the shipping build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/gpu_timeline.py 5 1 2 > gpurun_out/r05_gemv_timeline.txt 2>&1; head -40 gpurun_out/r05_gemv_timeline.txt
timeout 600 python tools/gpu_attn_timeline.py > gpurun_out/r05_attention_timeline.txt 2>&1; head -16 gpurun_out/r05_attention_timeline.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k thirty_two --durations=1 2>&1 | tail -4
