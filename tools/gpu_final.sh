cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
