#!/usr/bin/env python3
"""Seeded random sweep of the CPU oracle (oracle/ps_oracle.c) against the REAL reference compiled from /root/reference (oracle/_ref), dev container only:
whole-model generate() of LlamaModel / Qwen2Model::forward at random (preset incl. the odd head sizes / GQA ratios, Q4_0 / Q8_0 — the reference's loader
takes no K-quants —, context window, prompt length, prefill chunk, decode length), ids and every step's logits on bits; with --fast the stock-flags build
(libps_ref_fast.so) against the oracle's contract mode.  tools/gpu_fuzz.py compares the GPU with the oracle on the same kind of draws: together they tie
the GPU's bits to the reference's on shapes nobody named.

usage: cpu_fuzz_oracle.py [--seconds 60] [--seed 1] [--max-draws 0] [--fast]"""
import argparse, os, sys, tempfile, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tensors  # noqa: E402
from oracle import binding as B  # noqa: E402
from powerserve_amd import synth  # noqa: E402

PRESETS = ["tiny-llama", "tiny-qwen2", "small-llama", "small-llama-draft", "odd-llama-hs96", "odd-llama-hs32", "odd-qwen2-r3", "odd-llama-r5", "odd-llama-r6", "odd-llama-r8"]


def one(preset, wt, n_ctx, P, chunk, steps, mseed, fast):
    """one draw, in a process of its own: the reference leaks a ggml context and its spinning pool threads per model (its destructor calls gguf_free only),
    so a process that builds model after model runs out of contexts and of cores"""
    o = B.Oracle()
    r = B.Ref(2, so=B.REF_FAST_SO) if fast else B.Ref(2)
    o.L.pso_set_contract(1 if fast else 0)
    with tempfile.TemporaryDirectory() as d:
        fs, af = synth.ROPE_DRAWS[mseed % len(synth.ROPE_DRAWS)]  # (rope_freq_scale, rope_attn_factor: src/core/config.cpp:96,98)
        mj = synth.write_model_dir(d, preset, wt, n_ctx=n_ctx, seed=mseed, rope_freq_scale=fs, rope_attn_factor=af)
        cfg = B.make_config(mj["llm_config"])
        path = os.path.join(d, "ggml/weights.gguf")
        om = o.model(cfg, mj["model_arch"], load_tensors(path), n_threads=4)
        rm = r.model(path, mj["model_arch"], cfg, 2)
        prompt = np.random.default_rng(mseed + 1).integers(0, cfg.vocab_size, P)
        ids, lg, *_ = om.generate(prompt, chunk, steps, want_logits=True)
        rids, rlg, *_ = rm.generate(prompt, chunk, steps, want_logits=True)
        same = np.array_equal(ids, rids) and np.array_equal(lg.view(np.uint32), rlg.view(np.uint32))
        print("OK" if same else f"FAIL ids equal {np.array_equal(ids, rids)}, {int((lg != rlg).sum())} of {lg.size} logits differ", flush=True)
    os._exit(0)  # (no destructors: the reference's pool threads are still spinning)


def run(seconds=60.0, seed=1, max_draws=0, fast=False, verbose=False):
    import subprocess
    if not (B.have_ref_fast() if fast else B.have_ref()):
        raise SystemExit("the reference library is not built (needs /root/reference): python -c 'from oracle import binding; binding.build()'")
    rng = np.random.default_rng(seed)
    fails, n = [], 0
    t_end = time.time() + seconds
    while time.time() < t_end and len(fails) < 5 and (max_draws <= 0 or n < max_draws):
        preset = PRESETS[int(rng.integers(0, len(PRESETS)))]
        wt = int(rng.choice([2, 8]))
        n_ctx = int(rng.choice([36, 64, 96, 132, 200, 300]))
        steps = int(rng.integers(1, 8))
        P = int(rng.integers(2, n_ctx - steps))
        chunk = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 33, 64, 128]))
        mseed = int(rng.integers(0, 1 << 30))
        tag = f"{preset} wt={wt} n_ctx={n_ctx} P={P} chunk={chunk} steps={steps} seed={mseed}"
        if verbose:
            print("draw", tag, flush=True)
        cmd = [sys.executable, os.path.abspath(__file__), "--one", preset, str(wt), str(n_ctx), str(P), str(chunk), str(steps), str(mseed)] + (["--fast"] if fast else [])
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            last = (out.stdout.strip().splitlines() or ["no output: " + out.stderr[-300:]])[-1]
        except subprocess.TimeoutExpired:
            last = "FAIL timed out (the reference's spin-barrier pool on a busy host?)"
        if not last.startswith("OK"):
            fails.append(f"{tag}: {last}")
        n += 1
    return n, fails


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-draws", type=int, default=0)
    ap.add_argument("--fast", action="store_true", help="the reference as its own CMake flags build it (-ffp-contract=fast) against the oracle's contract mode")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--one", nargs=7, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.one:
        one(a.one[0], *[int(v) for v in a.one[1:]], a.fast)
    n, fails = run(a.seconds, a.seed, a.max_draws, a.fast, a.verbose)
    print(f"cpu_fuzz_oracle seed {a.seed}{' (stock-flags build)' if a.fast else ''}: {n} models, ids and logits on bits; {len(fails)} failures")
    for f in fails:
        print("FAIL", f)
    sys.exit(1 if fails else 0)
