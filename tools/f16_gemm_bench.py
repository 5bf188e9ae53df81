#!/usr/bin/env python3
"""GEMM shapes of the fp16 perf mode (csrc/perf16.hip) on synthetic operands through ps_hip_debug_f16_gemm: us per launch, TFLOP/s, max |error| against a
k-ordered fp32 reference, by kernel variant (ps_hip_debug_set(4, v): 1 = 128-token tiles, 2 = 256 x 256 tiles on the LDS-DMA path, 0 = by shape).
usage: f16_gemm_bench.py [variants, default 1,2,0] [M list, default 512,2048]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import hip
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 0]
Ms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 2048]
ctx = hip.Ctx(0)
shapes = [("gate/up", 28672, 4096), ("down", 4096, 14336), ("O", 4096, 4096), ("Q/K/V", 6144, 4096)]
for M in Ms:
    for name, N, K in shapes:
        line = f"M {M:5d} {name:8s} N {N:6d} K {K:6d}:"
        for v in variants:
            ctx.check(ctx.L.ps_hip_debug_set(4, v))
            us, err = C.c_double(), C.c_double()
            ctx.check(ctx.L.ps_hip_debug_f16_gemm(ctx.h, M, N, K, 20, 0.0, C.byref(us), C.byref(err)))
            line += f"   v{v}: {us.value:8.1f} us {2.0 * M * N * K / us.value / 1e6:7.1f} TF/s err {err.value:.2e}"
        print(line, flush=True)
ctx.check(ctx.L.ps_hip_debug_set(4, 0))
