#!/usr/bin/env python3
"""Seeded random sweep of the HIP path against the CPU oracle (the checker; nothing here is shipped or timed), through the C-ABI.

The parametrized tests name their shapes; this draws them: quantized mat-muls of every weight type at random (K, N, columns, scales),
and whole models at random (preset, weight type, context window, prompt length, prefill chunking, decode length) followed by a random
token tree behind random hidden cache slots.  Every comparison is on bits (ids, logits, cache rows).

usage: gpu_fuzz.py [--seconds 240] [--seed 1] [--ops-share 0.3]
Prints one line per failure (with the draw that reproduces it) and a summary; exit code 1 on any mismatch."""
import argparse, ctypes as C, os, sys, tempfile, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tensors  # noqa: E402
from oracle import binding as B  # noqa: E402
from powerserve_amd import hip, host, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=240)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--ops-share", type=float, default=0.3)
ap.add_argument("--max-draws", type=int, default=0, help="stop after this many draws (0: by --seconds only): a fixed, repeatable sequence")
ap.add_argument("--big-ctx", action="store_true", help="context windows of 4608 / 6144 slots with caches longer than 4096 tokens (the two-launch single-token attention, the LDS-row soft-max)")
ap.add_argument("--odd-share", type=float, default=0.0, help="share of model draws from the odd head-size / GQA-ratio presets")
ap.add_argument("--replay", default="", help="one model draw instead of the sweep: 'preset wt n_ctx P chunk max_batch steps tree seed' (values of a draw line)")
ap.add_argument("--verbose", action="store_true", help="print every draw before it runs (the last line names a draw that killed the process)")
args = ap.parse_args()

oracle, ctx = B.Oracle(), hip.Ctx(0)
rng = np.random.default_rng(args.seed)
fails, n_ops, n_models, n_trees, n_host = [], 0, 0, 0, 0
t_end = time.time() + args.seconds


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def op_case():
    global n_ops
    wt = int(rng.choice([2, 8, 12, 13, 14]))
    kq = wt in (12, 13, 14)
    K = int(rng.integers(1, 17)) * 256 if kq else int(rng.integers(1, 130)) * 32
    if rng.random() < 0.15:
        K = int(rng.choice([4096, 14336, 2048, 8192] if kq else [896, 4864, 2048, 4096]))
    N = int(rng.integers(1, 700))
    bs = int(rng.choice([1, 1, 1, 2, 3, 4, 7, 8, 9, 12, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 140]))
    if K * N * bs > 6e8:  # (the oracle is a scalar loop)
        bs = max(1, int(6e8 // (K * N)))
    w = synth.random_blocks(rng, wt, N, K)
    x = (rng.standard_normal((bs, K)) * rng.choice([0.01, 0.3, 1.0, 5.0, 40.0], (bs, 1))).astype(np.float32)
    if rng.random() < 0.2:
        x[rng.integers(0, bs), : min(K, 256)] = 0.0  # an all-zero activation block
    want = oracle.mul_mat(wt, w, K, N, x)
    W = ctx.upload_weight(wt, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    if not bits_equal(got, want):
        fails.append(f"mul_mat wt={wt} K={K} N={N} bs={bs}: {int((got != want).sum())} of {got.size} differ, first {np.argwhere(got != want)[:3].tolist()}")
    W.free(); dx.free(); dy.free()
    n_ops += 1


MODELS = [("tiny-llama", [2, 8, 12, 13, 14, 1015, 1017]), ("tiny-qwen2", [2, 8, 12]), ("small-llama", [2, 8, 12, 1015]),
          ("small-llama-hs128", [12, 13, 14, 1015, 1017, 2, 8]), ("small-llama-draft", [2, 12]), ("wide-llama", [12])]
# head sizes 32 / 96 and 1, 3, 5, 6, 8 query heads per kv head (synth.PRESETS "odd-*"; 384 and 320 are not multiples of 256: Q4_0 / Q8_0 only)
ODD = [("odd-llama-hs96", [2, 8, 12, 14]), ("odd-llama-hs32", [2, 8, 12]), ("odd-qwen2-r3", [2, 8]), ("odd-llama-r5", [2, 8]), ("odd-llama-r6", [8, 12, 13]), ("odd-llama-r8", [2, 8, 12])]


def random_tree(n):
    """parents of n nodes (node 0 the root), the visibility matrix (a node sees its ancestors and itself) and depths"""
    par = [-1] + [int(rng.integers(0, i)) for i in range(1, n)]
    vis = np.zeros((n, n), np.uint8)
    depth = [0] * n
    for i in range(n):
        j = i
        while j >= 0:
            vis[i, j] = 1
            j = par[j]
        depth[i] = 0 if par[i] < 0 else depth[par[i]] + 1
    return vis, np.array(depth, np.int32)


def stage(name):
    if args.verbose:
        print("  done:", name, flush=True)


def model_case(tmp):
    global n_models, n_trees, n_host
    pool = ODD if rng.random() < args.odd_share else MODELS
    preset, wts = pool[int(rng.integers(0, len(pool)))]
    wt = int(rng.choice(wts))
    n_ctx = int(rng.choice([64, 96, 160, 300, 520, 1100, 2100]))
    steps = int(rng.integers(2, 11))
    n_tree = int(rng.integers(2, 17))
    P = int(rng.integers(2, n_ctx - steps - n_tree - 1))
    if n_ctx >= 1100 and rng.random() < 0.6:
        P = int(rng.integers(n_ctx // 2, n_ctx - steps - n_tree - 1))  # (long caches: the attention's tails and slices)
    chunk = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 31, 32, 33, 64, 100, 128]))
    max_batch = int(rng.choice([chunk, max(chunk, 16), 128, 256, 512]))
    max_batch = max(max_batch, chunk, n_tree)
    seed = int(rng.integers(0, 1 << 30))
    if args.big_ctx:
        preset, wts = [("tiny-llama", [8, 12]), ("tiny-qwen2", [2, 8]), ("small-llama-draft", [2, 12]), ("small-llama-hs128", [12, 8])][int(rng.integers(0, 4))]
        wt = int(rng.choice(wts))
        n_ctx = int(rng.choice([4608, 6144]))
        P = int(rng.integers(4100, n_ctx - steps - n_tree - 1))
        chunk = int(rng.choice([33, 100, 128]))
        max_batch = int(rng.choice([128, 256, 512]))
    if args.replay:
        f = args.replay.split()
        preset, (wt, n_ctx, P, chunk, max_batch, steps, n_tree, seed) = f[0], [int(v) for v in f[1:9]]
    tag = f"{preset} wt={wt} n_ctx={n_ctx} P={P} chunk={chunk} max_batch={max_batch} steps={steps} tree={n_tree} seed={seed}"
    if args.verbose:
        print("draw", tag, flush=True)
    d = os.path.join(tmp, f"m{n_models}")
    fs, af = synth.ROPE_DRAWS[seed % len(synth.ROPE_DRAWS)]  # (rope_freq_scale, rope_attn_factor: src/core/config.cpp:96,98)
    mj = synth.write_model_dir(d, preset, wt, n_ctx=n_ctx, seed=seed, rope_freq_scale=fs, rope_attn_factor=af)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=8)
    gm = hip.Model(ctx, d, max_batch=max_batch)
    try:
        prompt = rng.integers(0, cfg.vocab_size, P)
        want_ids, want_lg, *_ = om.generate(prompt, chunk, steps, want_logits=True)
        got_ids = gm.generate(prompt, chunk, steps)
        stage("generate")
        if not np.array_equal(got_ids, want_ids):
            fails.append(f"generate ids: {tag}: {got_ids.tolist()} vs {want_ids.tolist()}")
            return
        # the same prompt through ps_hip_model_prefill (several reference chunks per launch sequence when max_batch allows), then every
        # step's logits, teacher-forced.  The SAME chunking: the reference's results depend on it (a soft-max row of n_kv = chunk end
        # entries takes libm's expf on its last n_kv % 8 and the V.p dot adds its last n_kv % 32 products after the chains)
        gm.reset()
        if P > 1:
            gm.prefill(prompt[:-1], chunk)
        ctx.sync(); stage("prefill")
        cur = int(prompt[-1])
        for s in range(steps):
            lg, am = gm.forward([cur], [gm.position], lm_head=True)
            if not bits_equal(lg[0], want_lg[s]) or int(am[0]) != int(want_ids[s]):
                fails.append(f"decode logits step {s}: {tag}: {int((lg[0] != want_lg[s]).sum())} logits differ")
                return
            cur = int(want_ids[s])
        stage("decode steps")
        n = gm.position
        for L in (0, cfg.n_layers - 1):
            if not (np.array_equal(gm.k_cache(L)[:n], om.k_cache(L)[:n]) and np.array_equal(gm.v_cache(L)[:, :n], om.v_cache(L)[:, :n])):
                fails.append(f"cache rows layer {L}: {tag}")
                return
        n_models += 1
        # a random token tree behind random hidden cache slots
        vis, depth = random_tree(n_tree)
        toks = rng.integers(0, cfg.vocab_size, n_tree)
        kv_vis = np.ones(cfg.seq_len, np.uint8)
        hidden = [int(h) for h in rng.choice(n, size=min(n - 1, int(rng.integers(0, 4))), replace=False)] if n > 1 else []
        for h in hidden:
            kv_vis[h] = 0
            gm.kv_mask(h, False)
        rope = n + depth
        want = om.forward_tree(toks, rope, vis, kv_vis, True, advance=False)
        got, am = gm.forward_tree(toks, rope, vis, lm_head=True, want_logits=True, advance=False)
        if not bits_equal(got, want) or not np.array_equal(am, want.argmax(axis=1)):
            fails.append(f"tree forward: {tag} hidden={hidden}: {int((got != want).sum())} of {got.size} logits differ")
            return
        n_trees += 1
        stage("tree")
        # the reference-API path (Graph builders -> Executor / HIPBackend::plan) on the same files: lowered to the fused launch plan, or op by op
        gm.close()
        hm = host.HostModel(d, 0, max_batch)
        try:
            fused = bool(rng.random() < 0.6) or P > 600  # (op by op: a launch per operator and column block; short prompts only)
            if args.verbose:
                print("  host fused", fused, flush=True)
            hm.set_fused(fused)
            ids = hm.generate(prompt, chunk, steps)
            if not np.array_equal(ids, want_ids):
                fails.append(f"host generate (fused={fused}): {tag}: {ids.tolist()} vs {want_ids.tolist()}")
                return
            lg = hm.forward([int(want_ids[-1])], [hm.position], lm_head=True)  # one more step: its logits on bits
            want1 = om.forward([int(want_ids[-1])], [om.position], True)
            if not bits_equal(lg[0], want1[0]):
                fails.append(f"host logits (fused={fused}): {tag}: {int((lg[0] != want1[0]).sum())} differ")
                return
            n_host += 1
        finally:
            hm.close()
    finally:
        gm.close(); om.close()
        for f in ("ggml/weights.gguf", "model.json"):
            try:
                os.remove(os.path.join(d, f))
            except OSError:
                pass


with tempfile.TemporaryDirectory() as tmp:
    n_draws = 0
    while time.time() < t_end and len(fails) < 10 and (args.max_draws <= 0 or n_draws < args.max_draws):
        n_draws += 1
        try:
            if args.replay:
                model_case(tmp)
                break
            if rng.random() < args.ops_share:
                op_case()
            else:
                model_case(tmp)
        except Exception as e:  # a refused shape is a finding too: report the draw, keep going
            fails.append(f"exception: {type(e).__name__}: {str(e)[:300]}")
print(f"gpu_fuzz seed {args.seed}: {n_ops} mat-muls, {n_models} models (generate + per-step logits + cache rows), {n_trees} tree forwards, {n_host} runs of the reference-API path; {len(fails)} failures")
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
