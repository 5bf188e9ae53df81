#!/usr/bin/env python3
"""gemm4k2_kernel (round 5) against gemm4k_kernel (bit-exact against the oracle since round 2) on the same inputs: bitwise equality of the op on wide,
ragged, extreme shapes (EPI 0, with one shape also against the oracle itself), and of a whole Q4_K model prefill (EPI 1: gate/up with SiLU, residuals)."""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from powerserve_amd import hip, synth
ctx = hip.Ctx(0)
L = ctx.L
L.ps_hip_last_matmul_kernel.restype = C.c_char_p
bad = 0
V2 = int(os.environ.get("G4K2_V", "1"))  # 1 / 8 / 16: the producers' ring depth
def mm(W, x, N):
    dx, dy = ctx.to_device(x), ctx.empty((x.shape[0], N))
    ctx.check(L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    return dy.numpy(), L.ps_hip_last_matmul_kernel().decode()
for (K, N, bs) in [(4096, 512, 128), (1024, 96, 120), (14336, 64, 113), (2048, 288, 128), (1024, 64, 160), (2048, 96, 200), (1024, 8224, 70), (4096, 256, 512), (2048, 32, 130), (1024, 64, 17), (4096, 4096, 512)]:
    rng = np.random.default_rng(K + N + bs)
    w = synth.random_blocks(rng, 12, N, K)
    x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
    x[min(3, bs - 1), 256:512] = 0.0
    W = ctx.upload_weight(12, w, K, N)
    L.ps_hip_debug_set(7, 0); y1, k1 = mm(W, x, N)
    L.ps_hip_debug_set(7, V2); y2, k2 = mm(W, x, N)
    ok = np.array_equal(y1.view(np.uint32), y2.view(np.uint32))
    bad += not ok
    print(f"K {K} N {N} bs {bs}: {k1} vs {k2}: {'bit-equal' if ok else 'DIFFERENT ' + str(np.argwhere(y1 != y2)[:6].tolist())}", flush=True)
    if (K, N, bs) == (1024, 96, 120):
        from oracle import binding as B
        want = B.Oracle().mul_mat(12, w, K, N, x)
        ok = np.array_equal(y2.view(np.uint32), want.view(np.uint32)); bad += not ok
        print("   vs oracle:", "bit-equal" if ok else "DIFFERENT")
    W.free()
# model level (EPI 1 + residual epilogues): caches and logits, v1 vs v2
with tempfile.TemporaryDirectory() as d:
    synth.write_model_dir(d, "small-llama-hs128", 12, n_ctx=512, seed=3)
    prompt = np.random.default_rng(1).integers(0, 1024, 301)
    res = []
    for v in (0, V2):
        L.ps_hip_debug_set(7, v)
        m = hip.Model(ctx, d, max_batch=128, n_ctx=512)
        m.prefill(prompt[:300], 64)
        lg, _ = m.forward([int(prompt[300])], [300], lm_head=True)
        res.append((m.k_cache(1)[:300].copy(), m.v_cache(1)[:, :300].copy(), lg.copy()))
        m.close()
    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(res[0], res[1])); bad += not ok
    print("model prefill (gate/up SiLU epilogue, residual epilogues):", "bit-equal" if ok else "DIFFERENT")
L.ps_hip_debug_set(7, 0)
print("g4k2 check:", "ALL EQUAL" if not bad else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
