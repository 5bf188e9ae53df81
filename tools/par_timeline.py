#!/usr/bin/env python3
"""In-kernel timeline of gemm4k_par_kernel (k_gemm4k.hip) on the 8B model, 12 columns: wave 0 of every workgroup marks the clock seven times per
round (1 weights landed + header decoded, 2 transposed through LDS, 3 eight accumulator lanes, 4 mins lanes, 5 barrier, 6 chains, 7 barrier),
first four rounds.  usage: par_timeline.py [key ...]   48 QKV, 49 O, 50 down"""
# (the in-kernel marks live in the timeline build of the library: python -m powerserve_amd.build --timeline)
import os as _os
_tl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "powerserve_amd", "lib", "libps_hip_timeline.so")
if "PS_HIP_LIB" not in _os.environ and _os.path.exists(_tl):
    _os.environ["PS_HIP_LIB"] = _tl
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
keys = [int(a) for a in sys.argv[1:]] or [50, 49, 48]
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_spec_llama-3.1-8b_Q4_K_1234_1024")
if not os.path.exists(d + "/.done"):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=1024, seed=1234); open(d + "/.done", "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=1024)
bs = 12
toks = np.arange(bs, dtype=np.int32) + 7
m.forward(toks, np.arange(bs), lm_head=False)
NW = 1024
for key in keys:
    ctx.check(ctx.L.ps_hip_model_kv_truncate(m.h, 0))
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, None, 0))
    m.forward(toks, np.arange(bs), lm_head=False)
    buf = np.zeros(NW * 64, dtype=np.uint64)
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, buf.ctypes.data_as(C.c_void_p), buf.size))
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, -1, None, 0))
    ev = buf.reshape(NW, 64)[:, :32].astype(np.int64)
    ev = ev[ev[:, 0] > 0]
    dt_ref = (ev[:, 30] - ev[:, 29]) / 100.0
    mhz = np.median((ev[:, 31] - ev[:, 0]) / dt_ref)
    t_in = (ev[:, 29] - ev[:, 29].min()) / 100.0
    print(f"key {key} ({ctx.L.ps_hip_last_matmul_kernel().decode()}): {ev.shape[0]} workgroups; clock {mhz:.0f} ticks/us; lifetime median {np.median(dt_ref):.2f} us; entry p0/50/100 {np.percentile(t_in, [0, 50, 100]).round(2)}")
    own = (ev[:, 1:29] - ev[:, 0:1]) / mhz
    ok = (ev[:, 1:29] > 0).all(axis=0)
    t = own[:, ok].mean(axis=0)
    names = ["hdr", "trn", "acc", "min", "bar", "chn", "bar"]
    prev = 0.0
    for r in range(0, len(t), 7):
        seg = t[r:r + 7]
        print(f"  round {r // 7}: " + "  ".join(f"{names[i]} {x:.2f} (+{x - (seg[i - 1] if i else prev):.2f})" for i, x in enumerate(seg)))
        prev = seg[-1]
    print(f"  exit {((ev[:, 31] - ev[:, 0]) / mhz).mean():.2f}")
