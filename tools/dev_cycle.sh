#!/bin/bash
# dev loop: rebuild the HIP library, run tests+bench+rocprof on a GPU box, print the kernel summary
cd /root/repo || exit 1
python powerserve_amd/build.py 2>&1 | grep -v "^\[build\]" | head -20
timeout 3000 /usr/local/graft/bin/gpurun --timeout 1300 -- "bash tools/${1:-gpu_tests_spec.sh}" 2>&1 | grep -v "^\[bench\]" | tail -6 | cut -c1-420
python tools/prof_summary.py gpurun_out/prof_v3/v3_results.db | head -${2:-20}
