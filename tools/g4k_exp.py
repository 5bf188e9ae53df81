#!/usr/bin/env python3
"""Prefill timer of the bench model: 2048 tokens in 128-token chunks, best of two passes per value of ps_hip_debug_set(2, v)
(the what-if switch the mat-mul kernels may read while experimenting; 0 = production).  usage: g4k_exp.py [v ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth

flags = [int(a) for a in sys.argv[1:]] or [0]
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(os.path.join(d, ".done")):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=4096, seed=1234)
    open(os.path.join(d, ".done"), "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=4096)
prompt = np.random.default_rng(42).integers(0, m.cfg.vocab_size, 2048).astype(np.int32)
for f in flags + [0]:
    ctx.check(ctx.L.ps_hip_debug_set(2, f))
    best = 1e9
    for rep in range(2):
        ctx.check(ctx.L.ps_hip_model_kv_truncate(m.h, 0))
        ctx.sync()
        t0 = time.perf_counter()
        done = 0
        while done < 2048:
            m.forward(prompt[done:done + 128], np.arange(done, done + 128), lm_head=False)
            done += 128
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"flags {f:3d}: prefill {2048 / best:8.1f} tok/s  ({best * 1e3:.1f} ms)", flush=True)
