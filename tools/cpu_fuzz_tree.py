#!/usr/bin/env python3
"""Seeded random sweep of the oracle's token-tree forward (pso_model_forward_tree) and of its causal forwards with K-quant weights against the reference's
own compiled OPERATORS sequenced by oracle/ref_ops_forward.py (the reference's loader takes no K-quants and its models have no tree entry point; its
operators take both), dev container only: random preset / weight type (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, the K_M mixes), prefix length and chunking, tree shape
(2-16 nodes), RoPE positions = prefix + depth, hidden cache slots; logits of every node and the appended K / V rows on bits.
usage: cpu_fuzz_tree.py <seed> <seconds>"""
import os, sys, tempfile, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402
from oracle.ref_ops_forward import RefOpsModel  # noqa: E402
from powerserve_amd import gguf, synth  # noqa: E402

MODELS = [("tiny-llama", [2, 8, 12, 13, 14, 1015, 1017]), ("tiny-qwen2", [2, 8, 12]), ("small-llama-draft", [2, 12]), ("odd-llama-hs96", [8, 12, 14]),
          ("odd-llama-hs32", [2, 12]), ("odd-qwen2-r3", [2, 8]), ("odd-llama-r5", [8]), ("odd-llama-r6", [12, 13]), ("odd-llama-r8", [2, 12])]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def load(path):
    rd = gguf.GGUFReader(path)
    return {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}


def main(seed, seconds):
    o, r = B.Oracle(), B.Ref(2)
    rng = np.random.default_rng(seed)
    n, fails, t_end = 0, [], time.time() + seconds
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() < t_end and len(fails) < 5:
            preset, wts = MODELS[int(rng.integers(0, len(MODELS)))]
            wt = int(rng.choice(wts))
            n_ctx = int(rng.choice([40, 64, 96, 132]))
            n_tree = int(rng.integers(2, 17))
            P = int(rng.integers(1, n_ctx - n_tree))
            chunk = int(rng.choice([1, 3, 8, 20, 33, 64]))
            mseed = int(rng.integers(0, 1 << 30))
            tag = f"{preset} wt={wt} n_ctx={n_ctx} P={P} chunk={chunk} tree={n_tree} seed={mseed}"
            d = os.path.join(tmp, f"m{n}")
            fs, af = synth.ROPE_DRAWS[mseed % len(synth.ROPE_DRAWS)]
            mj = synth.write_model_dir(d, preset, wt, n_ctx=n_ctx, seed=mseed, rope_freq_scale=fs, rope_attn_factor=af)
            cfg = B.make_config(mj["llm_config"])
            tensors = load(os.path.join(d, "ggml", "weights.gguf"))
            om = o.model(cfg, mj["model_arch"], tensors, n_threads=4)
            rm = RefOpsModel(r, cfg, mj["model_arch"], tensors)
            prefix = rng.integers(0, cfg.vocab_size, P)
            ok = True
            for lo in range(0, P, chunk):
                hi = min(P, lo + chunk)
                a = om.forward(prefix[lo:hi], np.arange(lo, hi), True)
                b = rm.forward_causal(prefix[lo:hi], np.arange(lo, hi), True)
                if not np.array_equal(bits(a), bits(b)):
                    fails.append(f"causal chunk at {lo}: {tag}"); ok = False
                    break
            if ok:
                par = [-1] + [int(rng.integers(0, i)) for i in range(1, n_tree)]
                vis = np.zeros((n_tree, n_tree), np.uint8)
                depth = np.zeros(n_tree, np.int32)
                for i in range(n_tree):
                    j = i
                    while j >= 0:
                        vis[i, j] = 1
                        j = par[j]
                    depth[i] = 0 if par[i] < 0 else depth[par[i]] + 1
                kv_vis = np.ones(cfg.seq_len, np.uint8)
                if P > 1:
                    kv_vis[rng.choice(P, size=min(P - 1, int(rng.integers(0, 4))), replace=False)] = 0
                toks = rng.integers(0, cfg.vocab_size, n_tree)
                want = rm.forward_tree(toks, P + depth, vis, kv_vis, True, advance=False)
                got = om.forward_tree(toks, P + depth, vis, kv_vis, True, advance=False)
                if not np.array_equal(bits(got), bits(want)):
                    fails.append(f"tree logits: {tag}")
                else:
                    for L in range(cfg.n_layers):
                        if not (np.array_equal(bits(om.k_cache(L)[:P + n_tree]), bits(rm.k_cache[L][:P + n_tree])) and
                                np.array_equal(bits(om.v_cache(L)[:, :P + n_tree]), bits(rm.v_cache[L][:, :P + n_tree]))):
                            fails.append(f"cache rows layer {L}: {tag}")
                            break
            om.close()
            for f in ("ggml/weights.gguf", "model.json"):
                os.remove(os.path.join(d, f))
            n += 1
    print(f"cpu_fuzz_tree seed {seed}: {n} models (chunked causal forwards + a random token tree behind hidden slots, all weight types), oracle vs the reference's operators on bits; {len(fails)} failures")
    for f in fails:
        print("FAIL", f)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, float(sys.argv[2]) if len(sys.argv) > 2 else 60))
