#!/usr/bin/env python3
"""Prefill timing for an A/B library build (PS_HIP_LIB): the bench model, 2047 tokens through ps_hip_model_prefill (chunks of 128, super-chunks as the bench),
three passes, plus the event-bracketed replay of the gate/up chunk mat-mul at 128 and 512 columns."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(os.path.join(d, ".done")):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=4096, seed=1234)
    open(os.path.join(d, ".done"), "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=512, n_ctx=4096)
mode = int(os.environ.get("PS_MODE", "0"))
if mode:
    m.set_mode(mode)
L = ctx.L
L.ps_hip_model_bench_matmul.restype = C.c_int
L.ps_hip_model_bench_matmul.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
prompt = np.random.default_rng(42).integers(0, m.cfg.vocab_size, 2048).astype(np.int32)
line = f"lib {os.path.basename(os.environ.get('PS_HIP_LIB', 'libps_hip.so'))} mode {mode}:"
for rep in range(3):
    m.reset(); ctx.sync()
    t0 = time.perf_counter()
    m.prefill(prompt[:2047], 128)
    ctx.sync()
    line += f" prefill {2047 / (time.perf_counter() - t0):8.0f} tok/s"
if not mode:
    for bs in (128, 512):
        seq_ms, null_ms, n = C.c_double(), C.c_double(), C.c_int()
        ctx.check(L.ps_hip_model_bench_matmul(m.h, 5, 1, bs, C.byref(seq_ms), C.byref(null_ms), C.byref(n)))
        line += f"  gate/up mat-mul bs {bs}: {1e3 * seq_ms.value / n.value:7.1f} us"
print(line, flush=True)
