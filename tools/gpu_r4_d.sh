# round 4, fourth GPU pass: pre-expanded stage images for the wide Q4_K mat-mul (parity at model level, prefill timing A/B)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_host.py -m gpu -q -x > $O/r04d_pytest_model.txt 2>&1; tail -3 $O/r04d_pytest_model.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k headline > $O/r04d_pytest_headline.txt 2>&1; tail -3 $O/r04d_pytest_headline.txt
timeout 600 python tools/prefill_ab.py > $O/r04d_prefill_ab.txt 2>&1
PS_NO_G4K_IMG=1 timeout 600 python tools/prefill_ab.py >> $O/r04d_prefill_ab.txt 2>&1
cat $O/r04d_prefill_ab.txt
