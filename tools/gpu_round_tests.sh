# a round's parity evidence: the whole GPU suite on the build that ships, then smoke() (R=r06 ...)
R=${R:-r06}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/${R}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/${R}_pytest_gpu.txt
