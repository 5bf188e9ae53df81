#!/usr/bin/env python3
"""In-kernel timeline of the Q4_K prefill mat-mul (k_gemm4k.hip) on the bench model: consumer wave 0 and producer wave 8 of
every workgroup mark the clock after the barrier of steps 0..7 and of every 8th step after.
usage: gpu_g4k_timeline.py [key ...]   48 QKV, 49 O, 50 down, 52 gate/up"""
# (the in-kernel marks live in the timeline build of the library: python -m powerserve_amd.build --timeline)
import os as _os
_tl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "powerserve_amd", "lib", "libps_hip_timeline.so")
if "PS_HIP_LIB" not in _os.environ and _os.path.exists(_tl):
    _os.environ["PS_HIP_LIB"] = _tl
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth

keys = [int(a) for a in sys.argv[1:]] or [49, 50, 52, 48]
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(os.path.join(d, ".done")):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=4096, seed=1234)
    open(os.path.join(d, ".done"), "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=4096)
prompt = np.random.default_rng(42).integers(0, m.cfg.vocab_size, 512).astype(np.int32)
for c in range(2):
    m.forward(prompt[c * 128:(c + 1) * 128], np.arange(c * 128, (c + 1) * 128), lm_head=False)
NW = 1024
for key in keys:
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, None, 0))
    bs = int(os.environ.get('TL_BS', '128'))  # width of the recorded forward
    m.forward(prompt[256:256 + bs], np.arange(256, 256 + bs), lm_head=False)
    buf = np.zeros(NW * 64, dtype=np.uint64)
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, buf.ctypes.data_as(C.c_void_p), buf.size))
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, -1, None, 0))
    ctx.check(ctx.L.ps_hip_model_kv_truncate(m.h, 256))
    ev = buf.reshape(NW, 2, 32).astype(np.int64)
    ev = ev[ev[:, 0, 0] > 0]
    n = ev.shape[0]
    dt_ref = (ev[:, 0, 30] - ev[:, 0, 29]) / 100.0
    mhz = np.median((ev[:, 0, 31] - ev[:, 0, 0]) / dt_ref)
    t_in = (ev[:, 0, 29] - ev[:, 0, 29].min()) / 100.0
    t_out = (ev[:, 0, 30] - ev[:, 0, 29].min()) / 100.0
    print(f"key {key}: {n} workgroups recorded; clock {mhz:.0f} ticks/us; lifetime median {np.median(dt_ref):.2f} us; "
          f"entry p0/50/100 {np.percentile(t_in, [0, 50, 100]).round(2)}; exit p0/50/100 {np.percentile(t_out, [0, 50, 100]).round(2)}")
    first = ev[t_in < 1.0]  # the first round of workgroups
    for role, name in ((0, "consumer wave 0"), (1, "producer wave 8")):
        for sel, tag in ((first, "first round"), (ev, "all")):
            own = (sel[:, role, 1:29] - sel[:, role, 0:1]) / mhz
            ok = (sel[:, role, 1:29] > 0).all(axis=0)
            t = own[:, ok].mean(axis=0)
            steps = [g for g in range(400) if g < 8 or (g & 7) == 7][:len(t)]
            print(f"  {name}, {tag} ({sel.shape[0]} wgs): us since own entry after the barrier of step:")
            print("    " + " ".join(f"{g}:{x:.2f}" for g, x in zip(steps, t)))
            end = (sel[:, role, 31] - sel[:, role, 0]) / mhz
            print(f"    exit {end.mean():.2f}; per step between marks: " + " ".join(f"{(t[i + 1] - t[i]) / (steps[i + 1] - steps[i]):.2f}" for i in range(len(t) - 1)))
