# round 4: batch attention kernels take (pos0, bs) from the launch arguments on eager forwards
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_speculative.py tests/test_gpu_model.py tests/test_gpu_host.py -m gpu -q -x > $O/r04m_pytest.txt 2>&1; tail -2 $O/r04m_pytest.txt
timeout 600 python tools/bench_verify.py Q4_K 2,8,12,16 > $O/r04m_tree.txt 2> $O/r04m_tree.err; cut -c1-300 $O/r04m_tree.txt; tail -2 $O/r04m_tree.err
timeout 600 python tools/prefill_ab.py > $O/r04m_prefill_ab.txt 2>&1; cat $O/r04m_prefill_ab.txt
