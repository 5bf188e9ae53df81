#!/usr/bin/env python3
"""Where a decode layer's time goes BETWEEN its launches: for every pair of consecutive launches of the last layer the in-kernel 100 MHz stamps of both
(first workgroup entry, last workgroup exit: ps_hip_debug_timeline with a two-key request) -> the kernel boundary as the hardware sees it, next to each
launch's own span.  Needs the timeline build (python -m powerserve_amd.build --timeline).
keys: 9 QKV, 42 one-launch attention, 3 O, 5 gate/up, 2 down.   usage: gpu_boundaries.py [n_prompt]   (TL_MODE=0 hipGraph replay (default), 1 eager)"""
import os as _os
_tl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "powerserve_amd", "lib", "libps_hip_timeline.so")
if "PS_HIP_LIB" not in _os.environ and _os.path.exists(_tl):
    _os.environ["PS_HIP_LIB"] = _tl
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import hip, synth

n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 2040
d = tempfile.mkdtemp(prefix="ps_tl_")
synth.write_model_dir(d, "llama-8b-dims-4l", 12, n_ctx=4096, seed=1)
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=4096)
ctx.check(ctx.L.ps_hip_debug_set(1, int(os.environ.get("G4_CFG", "0"))))
ctx.check(ctx.L.ps_hip_debug_set(2, int(os.environ.get("G4_FLAGS", "0"))))
mode = int(os.environ.get("TL_MODE", "0"))
prompt = np.random.default_rng(1).integers(0, m.cfg.vocab_size, n_prompt).astype(np.int32)
done = 0
while done < n_prompt:
    bs = min(128, n_prompt - done)
    m.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
    done += bs
NW = 1024
NAMES = {9: "QKV", 42: "attention", 3: "O", 5: "gate/up", 2: "down"}
spans, gaps = {}, {}
for k1, k2 in ((9, 42), (42, 3), (3, 5), (5, 2)):
    m.set_mode(16); m.set_mode(mode)  # drop the captured graph: the timeline pointers are baked into the launches
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, k1 + 100 * (k2 + 1), None, 0))
    g, s1, s2 = [], [], []
    for rep in range(6):
        m.decode_greedy(7, 3)
        buf = np.zeros(2 * NW * 64, dtype=np.uint64)
        ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, k1 + 100 * (k2 + 1), buf.ctypes.data_as(C.c_void_p), buf.size))
        ev = buf.reshape(2, NW, 64).astype(np.int64)
        a, b = ev[0][ev[0][:, 0] > 0], ev[1][ev[1][:, 0] > 0]
        if rep == 0 or not len(a) or not len(b):
            continue
        xa, xb = np.maximum(a[:, 30], a[:, 62]).max(), np.maximum(b[:, 30], b[:, 62]).max()  # (mat-vec: the chain wave's record [32..63] ends last)
        g.append((b[:, 29].min() - xa) / 100.0)
        s1.append((xa - a[:, 29].min()) / 100.0)
        s2.append((xb - b[:, 29].min()) / 100.0)
    ctx.check(ctx.L.ps_hip_debug_timeline(ctx.h, -1, None, 0))
    gaps[(k1, k2)] = float(np.median(g))
    spans.setdefault(k1, []).append(float(np.median(s1)))
    spans.setdefault(k2, []).append(float(np.median(s2)))
    print(f"{NAMES[k1]:>10s} -> {NAMES[k2]:<10s}: last exit -> first entry {np.median(g):5.2f} us (min {min(g):.2f} max {max(g):.2f});  spans {np.median(s1):5.2f} / {np.median(s2):5.2f} us", flush=True)
print(f"n_kv ~ {m.position}; mode {'hipGraph replay' if mode == 0 else 'eager'}; in-kernel spans (first entry -> last exit, timeline build): " +
      ", ".join(f"{NAMES[k]} {np.mean(v):.2f}" for k, v in spans.items()))
print("sum of spans + the four measured boundaries (+ down -> next QKV, not measured: taken as the mean of the others): "
      f"{sum(np.mean(v) for v in spans.values()) + sum(gaps.values()) * 5 / 4:.1f} us per layer")
