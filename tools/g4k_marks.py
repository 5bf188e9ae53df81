#!/usr/bin/env python3
"""Raw in-kernel marks of the wide chunk mat-mul's consumer wave 0 (library built with -DG4K_MARK2: a mark behind the barrier AND one behind the step's
chains, steps 0..7): how much of a super-block step is the wave's own work and how much is waiting at the barrier.  usage: PS_HIP_LIB=... g4k_marks.py [key]"""
# (the in-kernel marks live in the timeline build of the library: python -m powerserve_amd.build --timeline)
import os as _os
_tl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "powerserve_amd", "lib", "libps_hip_timeline.so")
if "PS_HIP_LIB" not in _os.environ and _os.path.exists(_tl):
    _os.environ["PS_HIP_LIB"] = _tl
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
key = int(sys.argv[1]) if len(sys.argv) > 1 else 52
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(os.path.join(d, ".done")):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=4096, seed=1234); open(os.path.join(d, ".done"), "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=512, n_ctx=4096)
prompt = np.random.default_rng(42).integers(0, m.cfg.vocab_size, 1024).astype(np.int32)
m.forward(prompt[:512], np.arange(512), lm_head=False)
ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, None, 0))
m.forward(prompt[512:1024], np.arange(512, 1024), lm_head=False)
NW = 1024
buf = np.zeros(NW * 64, dtype=np.uint64)
ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, buf.ctypes.data_as(C.c_void_p), buf.size))
ev = buf.reshape(NW, 2, 32).astype(np.int64)
ev = ev[ev[:, 0, 0] > 0]
mhz = np.median((ev[:, 0, 31] - ev[:, 0, 0]) / ((ev[:, 0, 30] - ev[:, 0, 29]) / 100.0))
for role, name in ((0, "consumer wave 0"), (1, "producer wave 8")):
    t = ((ev[:, role, 1:25] - ev[:, role, 0:1]) / mhz)
    ok = (ev[:, role, 1:25] > 0).all(axis=0)
    mean = t[:, ok].mean(axis=0)
    print(f"key {key} {name}: {ev.shape[0]} workgroups, clock {mhz:.0f} ticks/us; marks (us since entry): " + " ".join(f"{x:.2f}" for x in mean))
    print("   deltas: " + " ".join(f"{b - a:.2f}" for a, b in zip(mean[:-1], mean[1:])))
