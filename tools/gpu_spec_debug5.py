"""Localise a batch-vs-single-token discrepancy: same token at the same position through bs=1 and bs=3, last-layer scratch."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from powerserve_amd import gguf, hip, synth
preset, wt, P, STEP, BS = "small-llama-hs128", 12, 250, 4, 3
tmp = os.environ.get("TMPDIR", "/tmp")
d = os.path.join(tmp, f"ps_dbg4_{preset}_{wt}")
mj = synth.write_model_dir(d, preset, wt, n_ctx=512, seed=5)
ctx = hip.Ctx(0)
t = hip.Model(ctx, d, max_batch=128, n_ctx=512)
prompt = np.random.default_rng(42).integers(0, t.cfg.vocab_size, P)
def prefill():
    t.reset(); done = 0
    while done < P - 1:
        bs = min(128, P - 1 - done); t.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False); done += bs
prefill()
cur = int(prompt[-1]); ids = []
for s in range(STEP):
    lg, am = t.forward([cur], [t.position], lm_head=True); cur = int(am[0]); ids.append(cur)
p0 = t.position
print("pos0", p0, "layers", t.cfg.n_layers, "heads", t.cfg.n_heads, t.cfg.n_kv_heads)
lg1, _ = t.forward([cur], [p0], lm_head=True)
s1 = [t.scratch(w, 1).copy() for w in range(5)]
t.kv_rollback(1)
for bs in (2, BS, 5):
    lgb, _ = t.forward([cur] + [7] * (bs - 1), np.arange(p0, p0 + bs), lm_head=True)
    sb = [t.scratch(w, bs).copy() for w in range(5)]
    t.kv_rollback(bs)
    print(f"bs {bs}: logits {np.abs(lgb[0]-lg1[0]).max():.3e}", " ".join(f"{n} {np.abs(sb[w][0]-s1[w][0]).max():.3e}" for w, n in enumerate(["x", "q", "att", "hb"])))
    H, nctx = t.cfg.n_heads, t.cfg.seq_len
    sc1 = s1[4][0].reshape(H, nctx)[:, :p0 + 1]; scb = sb[4][0].reshape(H, nctx)[:, :p0 + 1]
    print("   scores", np.abs(sc1 - scb).max())
    da = np.abs(sb[2][0] - s1[2][0]); 
    if da.max() > 0: print("   att bad idx", np.nonzero(da > 1e-7)[0][:40], da.max())
