#!/usr/bin/env python3
"""Race screen for the fp16 perf-mode GEMM kernels: every variant on a set of shapes, REPS times each, max |error| against the k-ordered fp32 reference
each time (a schedule that reads a staged tile before it has landed shows as rare wrong tiles).  usage: f16_gemm_screen.py [variants] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import hip
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = hip.Ctx(0)
shapes = [(256, 256, 64), (256, 256, 128), (300, 512, 256), (512, 512, 4096), (2048, 4096, 4096), (2047, 6144, 4096), (1024, 4096, 14336), (2048, 28672, 4096)]
bad = 0
for v in variants:
    ctx.check(ctx.L.ps_hip_debug_set(4, v))
    for M, N, K in shapes:
        worst = 0.0
        for r in range(reps):
            us, err = C.c_double(), C.c_double()
            ctx.check(ctx.L.ps_hip_debug_f16_gemm(ctx.h, M, N, K, 2, 1.0 if r & 1 else 0.0, C.byref(us), C.byref(err)))
            worst = max(worst, err.value / (2.0 if r & 1 else 1.0))
        tol = 2e-6 * K + 1e-5
        ok = worst <= tol
        bad += not ok
        print(f"variant {v} M {M:5d} N {N:6d} K {K:6d}: worst error over {reps} runs {worst:.2e} (bound {tol:.2e}) {'ok' if ok else 'WRONG'}", flush=True)
ctx.check(ctx.L.ps_hip_debug_set(4, 0))
print("SCREEN", "PASS" if bad == 0 else f"FAIL ({bad})")
