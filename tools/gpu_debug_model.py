import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as B
from powerserve_amd import hip, synth, gguf

def load_tensors(path):
    rd = gguf.GGUFReader(path)
    return {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne)+[1])[1]) for n, ti in rd.tensors.items()}

ctx = hip.Ctx(0); o = B.Oracle()
for preset, wt in (("tiny-llama", 2), ("tiny-llama", 8), ("tiny-llama", 12), ("tiny-qwen2", 8), ("small-llama", 2), ("small-llama-hs128", 12)):
    for seed in (1, 2, 3):
        d = tempfile.mkdtemp()
        mj = synth.write_model_dir(d, preset, wt, n_ctx=128, seed=seed)
        cfg = B.make_config(mj["llm_config"])
        om = o.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), 4)
        gm = hip.Model(ctx, d, max_batch=16)
        prompt = np.random.default_rng(42).integers(0, cfg.vocab_size, 21)
        ids, logits, *_ = om.generate(prompt, 8, 24, want_logits=True)
        gm.reset()
        gm.forward(prompt[:16], np.arange(16), lm_head=False); gm.forward(prompt[16:20], np.arange(16, 20), lm_head=False)
        cur = int(prompt[-1]); errs = []
        for s in range(24):
            lg, am = gm.forward([cur], [gm.position], True)
            errs.append(np.abs(lg[0]-logits[s]).max()/np.abs(logits[s]).max()); cur = int(ids[s])
        kerr = np.abs(gm.k_cache(0)[:44]-om.k_cache(0)[:44]).max()/np.abs(om.k_cache(0)[:44]).max()
        gids = gm.generate(prompt, 8, 24)
        print(preset, wt, seed, "max rel %.2e median %.2e k0err %.1e ids_equal %s top2gap %.3f" % (max(errs), np.median(errs), kerr, np.array_equal(gids, ids), np.min(np.sort(logits,1)[:,-1]-np.sort(logits,1)[:,-2])), flush=True)
        gm.close(); om.close()
