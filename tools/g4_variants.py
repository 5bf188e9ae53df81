#!/usr/bin/env python3
"""Wave-configuration sweep of the decode mat-vec (k_gemv4.hip) on the bench model: for every configuration
(ps_hip_debug_set(1, cfg)) replay each launch family of a token between HIP events (ps_hip_model_bench_gemv) and one
real greedy decode.  usage: g4_variants.py [cfg ...]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth

cfgs = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7]  # cfg + 100 * what-if flags
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(os.path.join(d, ".done")):
    synth.write_model_dir(d, "llama-3.1-8b", gguf.NAME_TYPE["Q4_K"], n_ctx=4096, seed=1234)
    open(os.path.join(d, ".done"), "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=4096)
L = ctx.L
L.ps_hip_model_bench_gemv.restype = C.c_int
L.ps_hip_model_bench_gemv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
prompt = np.random.default_rng(42).integers(0, m.cfg.vocab_size, 2048).astype(np.int32)
done = 0
while done < 2047:
    bs = min(128, 2047 - done)
    m.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
    done += bs
pos0 = m.position
mb = {2: 14.16, 3: 9.44, 1: 66.06, 4: 33.03, 5: 295.5, 0: 4221.4}
names = {2: "QKV", 3: "O", 1: "gate/up", 4: "down", 5: "lm_head", 0: "all"}
ref_ids = None
for cfg in cfgs:
    ctx.check(L.ps_hip_debug_set(1, cfg % 100))
    ctx.check(L.ps_hip_debug_set(2, cfg // 100))
    line = f"cfg {cfg}:"
    for which in (2, 3, 1, 4, 5, 0):
        seq_ms, null_ms, n = C.c_double(), C.c_double(), C.c_int()
        ctx.check(L.ps_hip_model_bench_gemv(m.h, 10, which, C.byref(seq_ms), C.byref(null_ms), C.byref(n)))
        us = 1e3 * seq_ms.value / n.value
        line += f"  {names[which]} {us:6.2f} us ({mb[which] / (n.value if which == 0 else 1) / us * 1e-0:5.2f} TB/s)" if which else f"  all {seq_ms.value:6.3f} ms ({mb[0] / seq_ms.value * 1e-3:5.2f} TB/s)"
    # a real decode (graph replay): 32 tokens from the same state
    MODE = int(os.environ.get("G4V_MODE", "0"))  # (128: the Q / K / V mat-vec and the single-token attention as two launches)
    m.set_mode(MODE)
    ctx.check(L.ps_hip_model_kv_truncate(m.h, pos0))
    m.set_mode(16 | MODE); m.set_mode(MODE)  # drop the captured graph: the launch plan changed
    m.decode_greedy(int(prompt[-1]), 4)
    ctx.check(L.ps_hip_model_kv_truncate(m.h, pos0))
    ctx.sync()
    t0 = time.perf_counter()
    ids = m.decode_greedy(int(prompt[-1]), 32)
    ctx.sync()
    dt = time.perf_counter() - t0
    if ref_ids is None:
        ref_ids = ids
    line += f"  decode {32 / dt:6.1f} tok/s ids {'same' if np.array_equal(ids, ref_ids) else 'DIFFERENT'}"
    print(line, flush=True)
