#!/usr/bin/env python3
"""Time the target-side pieces of a speculative iteration on the 8B shape: a bs-wide tree verify (forward_tree with
lm_head + arg-max, no logits copy) for several widths, next to a single-token step.  usage: bench_verify.py [wtype [widths [par,...]]]
(par: values of ps_hip_debug_set(3, .) to time side by side -- the narrow-batch few-tile mat-mul form, include/ps_hip.h)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
wt = sys.argv[1] if len(sys.argv) > 1 else "Q4_K"
tmp = os.environ.get("TMPDIR", "/tmp")
d = os.path.join(tmp, f"ps_spec_llama-3.1-8b_{wt}_1234_1024")
if not os.path.exists(d + "/.done"):
    wtid = {"Q4_K_M": synth.Q4_K_M, "Q5_K_M": synth.Q5_K_M}.get(wt) or gguf.NAME_TYPE[wt]
    synth.write_model_dir(d, "llama-3.1-8b", wtid, n_ctx=1024, seed=1234); open(d + "/.done", "w").write("ok")
ctx = hip.Ctx(0)
t = hip.Model(ctx, d, max_batch=128, n_ctx=1024)
P = 256
prompt = np.random.default_rng(42).integers(0, t.cfg.vocab_size, P).astype(np.int32)
t.forward(prompt[:128], np.arange(128), lm_head=False); t.forward(prompt[128:], np.arange(128, P), lm_head=False)
widths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 12, 16, 24, 32]
pars = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]
for par in pars:
    if par is not None: assert ctx.L.ps_hip_debug_set(3, par) == 0
    out = {}
    for bs in widths:
        toks = np.arange(bs, dtype=np.int32) + 5
        pos = np.array([t.position] + [t.position + 1] * (bs - 1), dtype=np.int32)
        tree = np.eye(bs, dtype=np.uint8); tree[:, 0] = 1
        for _ in range(3): t.forward_tree(toks, pos, tree, lm_head=True, want_logits=False, advance=False)
        ctx.sync(); t0 = time.perf_counter()
        R = 20
        for _ in range(R): t.forward_tree(toks, pos, tree, lm_head=True, want_logits=False, advance=False)
        ctx.sync(); out[bs] = 1e3 * (time.perf_counter() - t0) / R
    res = {"workload": f"llama-3.1-8b {wt}, KV prefix {P}, tree forward incl. lm_head + arg-max", "ms_per_forward_by_width": out}
    if par is not None: res["gemm4k_par"] = par
    print(json.dumps(res), flush=True)
