#!/usr/bin/env python3
"""Seeded random sweep of the product's token tree (csrc/host/speculative.cpp) against the reference's compiled src/speculative/token_tree.cpp (oracle/_ref),
dev container only: the live comparison of tests/test_token_tree_vs_ref.py (node order, masks, model / KV-cache call sequence, emitted tokens) over many seeds.
usage: cpu_fuzz_token_tree.py <first_seed> <n_seeds>"""
import os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_token_tree_vs_ref as T  # noqa: E402
from oracle import binding as B  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
BATCH = 30  # per process: every run of the reference's tree keeps one of ggml's 64 context slots (its tokenizer's GGUF)
if n > BATCH:
    import subprocess
    rc, tot_bad = 0, 0
    for lo in range(first, first + n, BATCH):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(lo), str(min(BATCH, first + n - lo))], capture_output=True, text=True, timeout=900)
        last = [l for l in r.stdout.splitlines() if l.startswith(("cpu_fuzz_token_tree", "FAIL"))]
        if r.returncode != 0 or not last:
            rc = 1
            print("\n".join(last) or (r.stdout + r.stderr)[-400:])
            tot_bad += 1
    print(f"cpu_fuzz_token_tree seeds {first}..{first + n - 1}: {n} configurations against the reference's token_tree.cpp (nodes, masks, call sequence, tokens); {tot_bad} failing batches")
    sys.exit(rc)
ref = B.Ref(2)
bad = []
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    cfg = dict(draft_batch_size=int(rng.integers(2, 24)), top_k=int(rng.integers(1, 24)), max_fan_out=int(rng.integers(1, 6)), early_stop=int(rng.integers(0, 2)),
               temperature=float(np.float32(rng.uniform(0.3, 3.0))), p_base=float(np.float32(rng.uniform(0.2, 1.0))), min_prob=float(np.float32(rng.uniform(0.0, 0.5))))
    scr = dict(shared_seed=int(rng.integers(1, 2**40)), target_seed=int(rng.integers(1, 2**40)), draft_seed=int(rng.integers(1, 2**40)),
               shared_w=float(np.float32(rng.uniform(5, 150))), target_w=float(np.float32(rng.uniform(0, 30))), draft_w=float(np.float32(rng.uniform(0, 30))),
               vocab=int(rng.integers(8, 300)), n_ctx=512)
    prefix, iters = rng.integers(0, scr["vocab"], int(rng.integers(0, 20))), int(rng.integers(1, 16))
    try:
        r = B.ref_token_tree_run(ref, B.SpecConfig(*[cfg[k] for k in T.CFG_KEYS]), B.Script(*[scr[k] for k in T.SCR_KEYS]), prefix, 1, iters)
        tokens, trees, stats, log, _ = T.run_product(cfg, scr, prefix, iters)
        T.compare(r, tokens, trees, stats, log, cfg["draft_batch_size"])
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print(f"cpu_fuzz_token_tree seeds {first}..{first + n - 1}: {n} configurations against the reference's token_tree.cpp (nodes, masks, call sequence, tokens); {len(bad)} failures")
for b in bad[:10]:
    print("FAIL", b)
sys.exit(1 if bad else 0)
