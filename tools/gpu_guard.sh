# the GPU suite and a random sweep with every device allocation ending at an unmapped page (PS_HIP_GUARD=1, csrc/api.hip: ps_dev_malloc): a kernel that
# reads or writes past the end of a buffer faults; the last test name in the log is the culprit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp PS_HIP_GUARD=1
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -1
timeout ${1:-900} python -m pytest tests -m gpu -x -v 2>&1 | grep -v "^$" | tail -6 | cut -c1-220 | tee gpurun_out/${R:-r06}_guard_pytest.txt
timeout 400 python tools/gpu_fuzz.py --seconds ${2:-120} --seed 4 --verbose 2>&1 | tail -3 | cut -c1-220 | tee gpurun_out/${R:-r06}_guard_fuzz.txt
