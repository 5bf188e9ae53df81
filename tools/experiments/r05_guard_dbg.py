"""PS_HIP_GUARD debugging: the quantizer test's exact sequence, with the differing bytes printed."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from powerserve_amd import hip
from oracle import binding as B
o = B.Oracle(); ctx = hip.Ctx(0)
for vdt in (8, 15):
    for K in (32, 256, 896, 2048, 4096, 4864, 14336):
        if vdt == 15 and K % 256: continue
        rng = np.random.default_rng(K + vdt)
        rows = 5
        x = rng.standard_normal((rows, K)).astype(np.float32)
        x[0] *= 0.01
        x[1] *= 30.0
        x[2, : min(K, 256)] = 0.0
        x[3, 3] = 7.5; x[3, 9] = -7.5; x[3, 40 % K] = 7.5
        x[4, ::2] = 1e-40
        dx = ctx.to_device(x)
        rs = ctx.L.ps_hip_row_size(vdt, K)
        out = ctx.empty((rows, rs), np.uint8)
        for rep in range(2):
            ctx.check(ctx.L.ps_hip_quantize_act(ctx.h, vdt, dx.ptr, K, rows, out.ptr))
            got = out.numpy()
            bad = []
            for r in range(rows):
                want = o.from_float(vdt, x[r])
                nz = np.flatnonzero(got[r] != want)
                if nz.size: bad.append((r, nz[:6].tolist(), int(nz.size), got[r][nz[:6]].tolist(), want[nz[:6]].tolist()))
            print("vdt", vdt, "K", K, "rep", rep, "x %x out %x" % (dx.ptr, out.ptr), "bad:", bad[:3], flush=True)
