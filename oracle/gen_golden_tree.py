#!/usr/bin/env python3
"""Golden vectors for the token-tree forward: tests/golden/tree_forward.npz.

Source: oracle/ref_ops_forward.py = the REAL reference's compiled operators (oracle/_ref) given the tree mask — pinned to
the real LlamaModel / Qwen2Model::forward in the causal case by tests/test_oracle_vs_ref.py.  Run in the dev container:
    python oracle/gen_golden_tree.py
Each case: a synthetic model (powerserve_amd/synth.py, seed recorded: the GPU test regenerates the identical file and
checks its sha256), a prefix, two hidden cache slots, a branching 12-node tree with RoPE positions = prefix + depth, the
logits of every node; then the accepted path 0 -> 1 -> 4 is compacted (KVCacheInterface::move), the cache advanced by 3
and one more token decoded behind it (single-token forward with hidden slots) — its logits, too."""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as B  # noqa: E402
from oracle.ref_ops_forward import RefOpsModel  # noqa: E402
from powerserve_amd import gguf, synth  # noqa: E402
from test_oracle_vs_ref import TREE, TREE_DEPTH  # noqa: E402

CASES = [("tiny-llama", 8, 96, 37), ("tiny-qwen2", 2, 96, 41), ("small-llama-hs128", 12, 160, 70), ("small-llama", 1015, 160, 101)]
HIDDEN = (5, 33)
ACCEPT = (0, 1, 4)


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    r = B.Ref(2)
    out = {}
    for ci, (preset, t, n_ctx, P) in enumerate(CASES):
        with tempfile.TemporaryDirectory() as d:
            seed = 700 + ci
            mj = synth.write_model_dir(d, preset, t, n_ctx=n_ctx, seed=seed)
            cfg = B.make_config(mj["llm_config"])
            path = os.path.join(d, "ggml", "weights.gguf")
            rd = gguf.GGUFReader(path)
            tensors = {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}
            m = RefOpsModel(r, cfg, mj["model_arch"], tensors)
            rng = np.random.default_rng(seed)
            prefix = rng.integers(0, cfg.vocab_size, P).astype(np.int32)
            done = 0
            while done < P:
                bs = min(32, P - done)
                m.forward_causal(prefix[done:done + bs], np.arange(done, done + bs), False)
                done += bs
            kv_vis = np.ones(n_ctx, dtype=np.uint8)
            kv_vis[list(HIDDEN)] = 0
            toks = rng.integers(0, cfg.vocab_size, 12).astype(np.int32)
            rope = (P + TREE_DEPTH).astype(np.int32)
            logits = m.forward_tree(toks, rope, TREE, kv_vis, True, advance=False)
            # accept 0 -> 1 -> 4: slots P, P + 1, P + 4 become P, P + 1, P + 2
            for L in range(cfg.n_layers):
                m.k_cache[L][P + 2] = m.k_cache[L][P + 4]
                m.v_cache[L][:, P + 2] = m.v_cache[L][:, P + 4]
            m.position = P + 3
            nxt = np.array([int(np.argmax(logits[4]))], dtype=np.int32)
            step = m.forward_tree(nxt, np.array([P + 3], dtype=np.int32), None, kv_vis, True, advance=False)
            k = f"c{ci}_"
            out.update({k + "preset": preset, k + "wt": t, k + "n_ctx": n_ctx, k + "seed": seed, k + "gguf_sha256": sha(path), k + "prefix": prefix,
                        k + "hidden": np.array(HIDDEN), k + "tokens": toks, k + "rope": rope, k + "tree": TREE, k + "logits": logits,
                        k + "accept": np.array(ACCEPT), k + "next": nxt, k + "step_logits": step})
    out["n_cases"] = len(CASES)
    path = os.path.join(ROOT, "tests", "golden", "tree_forward.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
