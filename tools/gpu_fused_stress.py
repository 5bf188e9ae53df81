#!/usr/bin/env python3
"""Long runs of the fused QKV + attention launch (k_qkvattn.hip) against the two launches it replaces (mode bit 7: gemv4 QKV + attn_decode2, themselves pinned to the
oracle by the GPU suite) at the 8B layer shape: greedy decode from several cache lengths so that the new position walks through every residue of 8 (K row groups) and 32
(V lines, score slices), hipGraph replay and eager; ids, the last step's logits and every layer's K / V cache rows compared on bits.  A rendezvous that let one workgroup
through early, a stale line or a missed drain shows up as a differing bit somewhere in a few thousand steps.
usage: gpu_fused_stress.py [steps_per_run=400]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import hip, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
fails = 0
for preset, n_ctx in (("llama-8b-dims-4l", 4096), ("llama-1b-dims-2l", 2048)):
    d = tempfile.mkdtemp(prefix="ps_fs_")
    synth.write_model_dir(d, preset, 12, n_ctx=n_ctx, seed=7)
    ctx = hip.Ctx(0)
    ctx.check(ctx.L.ps_hip_debug_set(7, 1))  # the fused launch wherever it is covered (the head-size-64 instance is not dispatched by default)
    m = hip.Model(ctx, d, max_batch=128, n_ctx=n_ctx)
    L = m.cfg.n_layers
    rng = np.random.default_rng(3)
    for start in (0, 1, 7, 33, 1000, n_ctx - steps - 3):
        prompt = rng.integers(0, m.cfg.vocab_size, max(start, 1)).astype(np.int32)
        res = {}
        for mode in (0, 128, 1, 129):  # fused / two launches, hipGraph replay; the same eager
            m.set_mode(mode)
            m.reset()
            done = 0
            while done < start:
                bs = min(128, start - done)
                m.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
                done += bs
            t0 = time.perf_counter()
            ids = m.decode_greedy(int(prompt[-1]), steps)
            dt = time.perf_counter() - t0
            lg, _ = m.forward([int(ids[-1])], [m.position], lm_head=True)
            n = m.position
            kv = [(m.k_cache(l)[:n].copy(), m.v_cache(l)[:, :n].copy()) for l in range(L)]
            res[mode] = (ids, np.asarray(lg[0]).copy(), kv, dt)
        for a, b in ((0, 128), (1, 129), (0, 1)):
            ok = np.array_equal(res[a][0], res[b][0]) and np.array_equal(res[a][1].view(np.uint32), res[b][1].view(np.uint32))
            ok = ok and all(np.array_equal(x[0].view(np.uint32), y[0].view(np.uint32)) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)) for x, y in zip(res[a][2], res[b][2]))
            if not ok:
                fails += 1
                first = int(np.argmax(np.append(res[a][0] != res[b][0], True)))
                print(f"MISMATCH {preset} start {start}: modes {a} vs {b}, first differing id at step {first}", flush=True)
        print(f"{preset} start {start:5d} + {steps} steps: fused {steps / res[0][3]:7.1f} tok/s, two launches {steps / res[128][3]:7.1f} tok/s (hipGraph replay); "
              f"{'ok' if not fails else 'FAILURES SO FAR: ' + str(fails)}", flush=True)
    m.close()
print("fused stress:", "PASS" if not fails else f"{fails} MISMATCHES")
sys.exit(1 if fails else 0)
