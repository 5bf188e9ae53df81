# round 5, call B: gemm4k2_kernel parity against gemm4k_kernel, then timing (prefill + the gate/up launch), both item orders
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/g4k2_check.py 2>&1 | tail -16
for v in 0 1; do for c in 0 4; do PS_G4K_V2=$v PS_G4K_CBX=$c timeout 300 python tools/prefill_ab.py 2>&1 | tail -1 | sed "s/^/v2 $v cbx $c: /"; done; done | tee gpurun_out/r05_g4k2.txt
