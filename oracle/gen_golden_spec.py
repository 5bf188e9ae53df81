"""Generates tests/golden/token_tree.npz from the REAL reference token tree (src/speculative/token_tree.cpp compiled into
oracle/_ref/libps_ref.so, driven by the scripted models of oracle/ref_token_tree.cpp).  Run in the dev container:
    python oracle/gen_golden_spec.py
Each case records the emitted tokens, every iteration's tree (token, position, parent per node; the attention mask)
and the complete sequence of model / KV-cache calls the reference made."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as B  # noqa: E402

# name: (config kwargs, script kwargs, prefix length, iterations)
CASES = {
    "default_agree": (dict(), dict(target_w=0.0, draft_w=0.0), 5, 12),                       # draft == target: long accepted paths
    "default_close": (dict(), dict(target_w=6.0, draft_w=6.0), 7, 16),                       # partial agreement
    "default_far": (dict(), dict(target_w=48.0, draft_w=48.0), 3, 10),                       # mostly rejected: catch-up forwards
    "peaky_deep": (dict(min_prob=0.05), dict(shared_w=160.0, target_w=4.0, draft_w=4.0), 5, 10),  # chains many levels deep
    "wide_bs16": (dict(draft_batch_size=16, max_fan_out=4, min_prob=0.05), dict(target_w=4.0, draft_w=4.0), 9, 12),
    "narrow_bs8_no_early_stop": (dict(draft_batch_size=8, early_stop=0, max_fan_out=2), dict(target_w=4.0, draft_w=4.0), 4, 12),
    "flat_sampler": (dict(temperature=4.0, p_base=0.5, top_k=8, min_prob=0.0), dict(target_w=2.0, draft_w=2.0, shared_w=24.0), 6, 10),
    "tiny_vocab_ties": (dict(draft_batch_size=10, top_k=6), dict(target_w=0.0, draft_w=0.0, vocab=6, shared_w=0.0), 2, 6),  # all logits 0: every tie rule
    "no_prefix": (dict(), dict(target_w=8.0, draft_w=8.0), 0, 8),
}


def make(cfg_kw, script_kw):
    c = dict(draft_batch_size=12, top_k=15, max_fan_out=3, early_stop=1, temperature=1.5, p_base=0.9, min_prob=0.2)
    c.update(cfg_kw)
    s = dict(shared_seed=0x1234, target_seed=0xAAAA, draft_seed=0xBBBB, shared_w=64.0, target_w=1.0, draft_w=1.0, vocab=64, n_ctx=512)
    s.update(script_kw)
    return c, s


def main():
    ref = B.Ref()
    out = {}
    for name, (ckw, skw, n_prefix, iters) in CASES.items():
        c, s = make(ckw, skw)
        cfg = B.SpecConfig(c["draft_batch_size"], c["top_k"], c["max_fan_out"], c["early_stop"], c["temperature"], c["p_base"], c["min_prob"])
        scr = B.Script(s["shared_seed"], s["target_seed"], s["draft_seed"], s["shared_w"], s["target_w"], s["draft_w"], s["vocab"], s["n_ctx"])
        prefix = (np.arange(n_prefix) * 7 + 3) % s["vocab"]
        r = B.ref_token_tree_run(ref, cfg, scr, prefix, root_token=1, n_iterations=iters)
        out[f"{name}/cfg"] = np.array([c[k] for k in ("draft_batch_size", "top_k", "max_fan_out", "early_stop", "temperature", "p_base", "min_prob")], dtype=np.float64)
        out[f"{name}/script"] = np.array([s[k] for k in ("shared_seed", "target_seed", "draft_seed", "shared_w", "target_w", "draft_w", "vocab", "n_ctx")], dtype=np.float64)
        out[f"{name}/prefix"] = prefix.astype(np.int32)
        out[f"{name}/iters"] = np.array([iters], dtype=np.int32)
        for k, v in r.items():
            out[f"{name}/{k}"] = v
        depth = int((r["tree"][:, :, 1].max(axis=1) - r["tree"][:, 0, 1]).max())
        print(f"{name:28s} {len(r['tokens']):4d} tokens in {iters} iterations ({len(r['tokens']) / iters:.2f}/it), {len(r['events'])} calls, deepest tree {depth}")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "token_tree.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
