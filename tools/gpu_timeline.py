#!/usr/bin/env python3
"""In-kernel timeline of the decode mat-vec on the bench model (ps_hip_debug_timeline).
usage: gpu_timeline.py [key ...]   key = epilogue*4 + prologue (1 = QKV, 2 = O/down, 5 = gate/up)"""
# (the in-kernel marks live in the timeline build of the library: python -m powerserve_amd.build --timeline)
import os as _os
_tl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "powerserve_amd", "lib", "libps_hip_timeline.so")
if "PS_HIP_LIB" not in _os.environ and _os.path.exists(_tl):
    _os.environ["PS_HIP_LIB"] = _tl
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import hip, synth

keys = [int(a) for a in sys.argv[1:]] or [5, 1, 2]
d = tempfile.mkdtemp(prefix="ps_tl_")
synth.write_model_dir(d, "llama-8b-dims-4l", 12, n_ctx=512, seed=1)
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=8, n_ctx=512)
ctx.check(ctx.L.ps_hip_debug_set(1, int(os.environ.get('G4_CFG', '0'))))
mode = int(os.environ.get('TL_MODE', '1'))  # 1 eager launches, 0 hipGraph replay
m.set_mode(mode)
if mode == 0:
    keys = []
m.forward(np.arange(8, dtype=np.int32) + 5, np.arange(8), lm_head=False)
NW = 1024
for key in keys:
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, None, 0))
    for _ in range(3):
        m.decode_greedy(7, 2)
    buf = np.zeros(NW * 64, dtype=np.uint64)
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, buf.ctypes.data_as(C.c_void_p), buf.size))
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, -1, None, 0))
    ev = buf.reshape(NW, 2, 32).astype(np.int64)
    used = ev[:, 0, 0] > 0
    ev = ev[used]
    n = ev.shape[0]
    # shader-clock ticks per microsecond from the 100 MHz reference pair
    dt_ref = (ev[:, 0, 30] - ev[:, 0, 29]) / 100.0
    dt_clk = (ev[:, 0, 31] - ev[:, 0, 0]).astype(np.float64)
    mhz = np.median(dt_clk / dt_ref)
    print(f"key {key}: {n} workgroups; s_memtime runs at {mhz:.0f} ticks/us; workgroup lifetime median {np.median(dt_ref):.2f} us")
    t_in = (ev[:, 0, 29] - ev[:, 0, 29].min()) / 100.0
    t_out = (ev[:, 0, 30] - ev[:, 0, 29].min()) / 100.0
    print(f"  100 MHz reference: first entry {ev[:, 0, 29].min() / 100.0:.2f} us, last exit {ev[:, 0, 30].max() / 100.0:.2f} us (absolute)")
    print("  entry time (us, 100 MHz reference) percentiles 0/25/50/75/100:", np.percentile(t_in, [0, 25, 50, 75, 100]).round(2),
          " exit:", np.percentile(t_out, [0, 25, 50, 75, 100]).round(2))
    hist, edges = np.histogram(t_in, bins=12)
    print("  entry histogram:", list(zip(edges[:-1].round(1), hist)))
    xcd = np.arange(n) % 8
    base = np.zeros(n)
    for x in range(8):  # counters are per XCD: times are relative to the first entry on the same XCD
        base[xcd == x] = ev[xcd == x, 0, 0].min()
    for role, name in ((0, "producer wave 0"), (1, "chain wave")):
        e = ev[:, role, :29]
        print(f"  {name}: event: mean / min / max since the XCD's first entry | mean since own entry (us)")
        for i in range(29):
            ok = e[:, i] > 0
            if not ok.any():
                continue
            v = (e[ok, i] - base[ok]) / mhz
            own = (e[ok, i] - ev[ok, 0, 0]) / mhz
            print(f"    {i:2d}: {v.mean():7.2f} {v.min():7.2f} {v.max():7.2f} | {own.mean():7.2f}   n={ok.sum()}")

    arr = ev[:, 1, 12:28]
    if (arr > 0).any():
        rel = (arr - ev[:, 0, 0][:, None]) / mhz
        rel[arr <= 0] = np.nan
        print("  waves 0..15: entry (gemv4), us since workgroup entry, mean:", np.nanmean(rel, axis=0).round(2))
# boundary between two consecutive kernels: gate/up (5) then down (2) of the same layer
if os.environ.get('TL_BOUNDARY', '0') != '1':
    sys.exit(0)
ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, 5 + 100 * (2 + 1), None, 0))
for _ in range(3):
    m.decode_greedy(7, 4)
buf = np.zeros(2 * NW * 64, dtype=np.uint64)
ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, 0, buf.ctypes.data_as(C.c_void_p), buf.size))
ev = buf.reshape(2, NW, 2, 32).astype(np.int64)
a, b = ev[0][ev[0][:, 0, 0] > 0], ev[1][ev[1][:, 0, 0] > 0]
# the buffers hold the LAST launch of each key: gate/up of the last layer, then down of the last layer
print(f"gate/up: first entry {a[:, 0, 29].min() / 100.0:.2f}  last exit {a[:, 0, 30].max() / 100.0:.2f} us")
print(f"down   : first entry {b[:, 0, 29].min() / 100.0:.2f}  last exit {b[:, 0, 30].max() / 100.0:.2f} us")
print(f"boundary (last exit of gate/up -> first entry of down): {(b[:, 0, 29].min() - a[:, 0, 30].max()) / 100.0:.2f} us")
