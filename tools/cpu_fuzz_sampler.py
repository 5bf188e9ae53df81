#!/usr/bin/env python3
"""Seeded random sweep of the host-side sampler chain (csrc/host/sampler.cpp) against the reference's own sampler classes compiled into oracle/_ref
(src/sampler/sampler.cpp, prob_array.cpp, chained as sampler_chain.cpp:19-51), dev container only: random configurations (top-k, top-p, temperature,
the penalties, seeds, vocabulary sizes) x random logits incl. exact ties, token sequences must be identical.
usage: cpu_fuzz_sampler.py <first_seed> <n_seeds>"""
import ctypes as C, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402
from powerserve_amd import host  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
ref = B.Ref(1)
bad, total = [], 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    n_vocab = int(rng.choice([7, 33, 256, 1000, 5003]))
    kw = dict(seed=int(rng.integers(0, 2**31)), temperature=float(np.float32(rng.choice([0.05, 0.3, 0.8, 1.0, 1.7, 4.0]))), top_p=float(np.float32(rng.choice([0.05, 0.5, 0.9, 0.95, 1.0]))),
              top_k=int(rng.choice([1, 2, 12, 40, 1000, 100000])), penalty_last_n=int(rng.choice([0, 4, 16, 64, 300])), penalty_repeat=float(np.float32(rng.choice([1.0, 1.1, 1.5]))),
              penalty_freq=float(np.float32(rng.choice([0.0, 0.2]))), penalty_present=float(np.float32(rng.choice([0.0, 0.3]))), penalize_nl=bool(rng.integers(0, 2)),
              linefeed_id=int(rng.integers(0, n_vocab)), ignore_eos=bool(rng.integers(0, 2)), special_eos_id=int(rng.integers(0, n_vocab)))
    try:
        cfg = host.SamplerCfg.make(n_vocab, **kw)
    except TypeError:  # (a keyword this build's SamplerCfg.make does not take)
        kw = {k: v for k, v in kw.items() if k in host.SamplerCfg.make.__code__.co_varnames}
        cfg = host.SamplerCfg.make(n_vocab, **kw)
    mine = host.Sampler(cfg)
    rc = B.SamplerCfg.from_buffer_copy(bytes(cfg))
    rh = ref.L.ref_sampler_create(C.byref(rc))
    got, want = [], []
    for s in range(int(rng.integers(20, 150))):
        lg = (rng.standard_normal(n_vocab) * float(rng.choice([0.5, 2.5, 8.0]))).astype(np.float32)
        if s % 5 == 0:
            lg[rng.integers(0, n_vocab, 3)] = lg.max()
        got.append(mine.sample(lg))
        want.append(ref.L.ref_sampler_sample(rh, lg.ctypes.data, n_vocab))
    ref.L.ref_sampler_free(rh)
    mine.close()
    total += len(got)
    if got != want:
        bad.append((seed, kw, next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)))
print(f"cpu_fuzz_sampler seeds {first}..{first + n - 1}: {n} configurations, {total} sampled tokens against the reference's sampler chain; {len(bad)} failures")
for b in bad[:5]:
    print("FAIL", b)
sys.exit(1 if bad else 0)
