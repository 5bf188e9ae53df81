#!/usr/bin/env python3
"""In-kernel timeline of the single-token attention kernels (ps_hip_debug_timeline keys 40 = scores, 41 = soft-max + V.p)
on the 8B layer shape with a long cache.  usage: gpu_attn_timeline.py [n_prefill=2048]"""
# (the in-kernel marks live in the timeline build of the library: python -m powerserve_amd.build --timeline)
import os as _os
_tl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "powerserve_amd", "lib", "libps_hip_timeline.so")
if "PS_HIP_LIB" not in _os.environ and _os.path.exists(_tl):
    _os.environ["PS_HIP_LIB"] = _tl
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import hip, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = tempfile.mkdtemp(prefix="ps_atl_")
synth.write_model_dir(d, "llama-8b-dims-4l", 12, n_ctx=4096, seed=1)
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=4096)
ctx.check(ctx.L.ps_hip_debug_set(2, int(os.environ.get("G4_FLAGS", "0"))))
rng = np.random.default_rng(0)
for lo in range(0, P, 128):
    m.forward(rng.integers(0, 4096, 128), np.arange(lo, lo + 128), lm_head=False)
NW = 1024
names = {40: ["entry", "loads issued", "q + first K round landed", "end"],
         41: ["entry", "V + score loads issued", "scores landed, logits in LDS", "barrier", "exp + row sums", "barrier", "1/sum", "V landed, stored to LDS", "barrier", "chains + reduce done", "end"],
         42: ["entry", "position, q, hinted K requested", "position landed, rest of K and V requested, q -> LDS", "scores computed + stored", "stores drained, V parked, barrier",
              "counter complete, barrier", "scores gathered, max barrier passed", "exp + partial sums", "sum barrier passed", "matrix chains + barrier", "end"]}
names[43] = ["entry", "activation + first weight chunk requested", "activation quantized", "second chunk, K rows, V pieces requested", "barrier: activation in LDS", "last chunk produced",
             "rendezvous A passed (q, new K row, new V column in memory)", "q -> LDS, fresh K row / V lines, barrier", "scores computed + stored", "stores drained, rendezvous B, barrier",
             "scores gathered, max barrier passed", "exp + partial sums, barrier", "matrix chains + barrier", "end", "(producers: K rows, V pieces requested)", "(producers: K / V landed)"]
MODES = {40: 17, 41: 17, 42: 129, 43: 1}  # two launches (mode bit 4) / the one-launch attention behind its own QKV launch (bit 7) / the fused QKV + attention launch; + eager
KEYS = [int(k) for k in os.environ.get("TL_KEYS", "43,42,40,41").split(",")]
for key in KEYS:
    m.set_mode(MODES[key])
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, None, 0))
    for _ in range(3):
        m.decode_greedy(7, 2)
    buf = np.zeros(NW * 64, dtype=np.uint64)
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, key, buf.ctypes.data_as(C.c_void_p), buf.size))
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, -1, None, 0))
    ev = buf.reshape(NW, 64).astype(np.int64)
    ev = ev[(ev[:, 0] > 0) & (ev[:, 30] > 0)]
    n = ev.shape[0]
    last = 10 if key == 42 else (13 if key == 43 else len(names[key]) - 1)
    chain = buf.reshape(NW, 64).astype(np.int64)[:, 32:]
    chain = chain[chain[:, 0] > 0]
    dt_ref = (ev[:, 30] - ev[:, 29]) / 100.0
    mhz = np.median((ev[:, last] - ev[:, 0]) / np.maximum(dt_ref, 1e-3))
    t0 = ev[:, 29].min()
    print(f"key {key}: {n} workgroups, n_kv = {m.position}; s_memtime {mhz:.0f} ticks/us; workgroup lifetime median {np.median(dt_ref):.2f} us; "
          f"first entry -> last exit {(ev[:, 30].max() - t0) / 100.0:.2f} us; entry spread p50/p100 {np.percentile((ev[:, 29] - t0) / 100.0, [50, 100]).round(2)}")
    for i, nm in enumerate(names[key]):
        if not (ev[:, i] > 0).all():
            continue
        own = (ev[:, i] - ev[:, 0]) / mhz
        print(f"    {i:2d} {nm:34s} since own entry: mean {own.mean():6.2f}  min {own.min():6.2f}  max {own.max():6.2f} us")
    if key == 43 and len(chain):
        for i, nm in ((1, "chain wave: rows chained, rotated, stored"), (2, "chain wave: rendezvous A complete")):
            own = (chain[:, i] - chain[:, 0]) / mhz
            print(f"       {nm:44s} since own entry: mean {own.mean():6.2f}  min {own.min():6.2f}  max {own.max():6.2f} us")
