#!/usr/bin/env python3
"""Per-launch time of the 8B layer mat-muls for a narrow batch (ps_hip_model_bench_matmul: activation quantizer + mat-mul, 32 layers
back to back) under values of ps_hip_debug_set(3, par) and the what-if switch ps_hip_debug_set(2, flags).
usage: par_exp.py bs par:flags [par:flags ...]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
bs = int(sys.argv[1])
combos = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or [(1, 0)]
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_spec_llama-3.1-8b_Q4_K_1234_1024")
if not os.path.exists(d + "/.done"):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=1024, seed=1234); open(d + "/.done", "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=1024)
m.forward(np.arange(bs, dtype=np.int32) + 7, np.arange(bs), lm_head=False)
names = {2: "QKV", 3: "O", 1: "gate/up", 4: "down"}
for par, fl in combos:
    ctx.check(ctx.L.ps_hip_debug_set(3, par)); ctx.check(ctx.L.ps_hip_debug_set(2, fl))
    line = f"par {par} flags {fl}:"
    for which in (2, 3, 1, 4):
        seq, null, n = C.c_double(), C.c_double(), C.c_int()
        ctx.check(ctx.L.ps_hip_model_bench_matmul(m.h, 20, which, bs, C.byref(seq), C.byref(null), C.byref(n)))
        line += f"  {names[which]} {1e3 * seq.value / n.value:6.2f} us"
    print(line + f"  (last kernel {ctx.L.ps_hip_last_matmul_kernel().decode()})", flush=True)
ctx.check(ctx.L.ps_hip_debug_set(2, 0)); ctx.check(ctx.L.ps_hip_debug_set(3, 1))
