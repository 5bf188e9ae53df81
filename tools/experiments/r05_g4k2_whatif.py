#!/usr/bin/env python3
"""Timing what-ifs of gemm4k2_kernel (library built by `tools/ab_build.py whatif k_gemm4k.hip -DG4K2_WHATIF=1`, run with PS_HIP_LIB=.../libps_hip_whatif.so
PS_G4K_V2=8): the gate/up chunk mat-mul of a 512-column sequence (event-bracketed replay over the 32 layers), one line per switch set.  Results are WRONG by
construction; only the times mean something.  bits: 1 no A-operand LDS reads, 2 no B-fragment loads, 4 producers store no operands, 8 one chain step per matrix
result instead of 16, 16 no matrix instructions, 32 producers: loads + barriers only, 64 no mins part."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(os.path.join(d, ".done")):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=4096, seed=1234)
    open(os.path.join(d, ".done"), "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=512, n_ctx=4096)
L = ctx.L
L.ps_hip_model_bench_matmul.restype = C.c_int
L.ps_hip_model_bench_matmul.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
L.ps_hip_last_matmul_kernel.restype = C.c_char_p
sets = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 24, 64, 91, 36, 127]  # (template instances of the A/B build; a what-if launch runs BEHIND the production launch: subtract line 0)
for which, name in ((1, "gate/up"), (4, "down")):
    for w in sets:
        L.ps_hip_debug_set(8, w)
        seq_ms, null_ms, n = C.c_double(), C.c_double(), C.c_int()
        ctx.check(L.ps_hip_model_bench_matmul(m.h, 3, which, 512, C.byref(seq_ms), C.byref(null_ms), C.byref(n)))
        print(f"{name} bs 512 whatif {w:3d} ({L.ps_hip_last_matmul_kernel().decode()}): {1e3 * seq_ms.value / n.value:7.1f} us per launch (quantizer launch included)", flush=True)
L.ps_hip_debug_set(8, 0)
