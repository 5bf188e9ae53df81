"""End-to-end parity of the fused model path (ps_hip_model_*) against the CPU oracle on identical GGUF weights.

north_star bar: bit-exact greedy token ids, logits within 1e-3 relative.
"""
import os

import numpy as np
import pytest

from conftest import load_tensors, rel_err

pytestmark = pytest.mark.gpu


CASES = [("tiny-llama", 2), ("tiny-llama", 8), ("tiny-llama", 12), ("tiny-qwen2", 8), ("tiny-qwen2", 2),
         ("small-llama", 2), ("small-llama-hs128", 12),
         ("small-llama-hs128", 14), ("small-llama-hs128", 1015), ("tiny-llama", 1015),  # 14: pure Q6_K; 1015: synth.Q4_K_M mix
         ("small-llama-hs128", 13), ("tiny-llama", 1017),                                # 13: pure Q5_K; 1017: synth.Q5_K_M mix
         # head sizes 96 / 32 and 3, 5, 6, 8 (and 1) query heads per kv head: no public config of the survey has them, the backend accepts them
         ("odd-llama-hs96", 12), ("odd-llama-hs32", 8), ("odd-qwen2-r3", 2), ("odd-llama-r5", 8), ("odd-llama-r6", 13), ("odd-llama-r8", 2)]


# rope_freq_scale / rope_attn_factor (src/core/config.cpp:96,98 -> ggml.c:15344-15358) off 1.0: Q4_K through the fused QKV epilogue, Qwen2 through the NEOX kernel
SCALED_ROPE = [("small-llama-hs128", 12, 0.5, 1.25), ("tiny-qwen2", 8, 0.25, 0.8), ("tiny-llama", 1015, 0.25, 1.0)]


@pytest.mark.parametrize("preset,wt,fs,af", [(p, w, 1.0, 1.0) for p, w in CASES] + SCALED_ROPE)
def test_generate_matches_oracle(ctx, oracle, tmp_path, preset, wt, fs, af):
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=128, seed=wt + len(preset), rope_freq_scale=fs, rope_attn_factor=af)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    gm = hip.Model(ctx, d, max_batch=16)
    rng = np.random.default_rng(42)
    prompt = rng.integers(0, cfg.vocab_size, 21)
    steps = 24
    want_ids, want_logits, *_ = om.generate(prompt, 8, steps, want_logits=True)
    # prefill in chunks (bs 8, 8, 4) + greedy decode through the captured graph
    got_ids = gm.generate(prompt, 8, steps)
    assert np.array_equal(got_ids, want_ids), (got_ids, want_ids)
    # eager mode must agree with graph replay
    gm.set_mode(1)
    assert np.array_equal(gm.generate(prompt, 8, steps), want_ids)
    gm.set_mode(0)
    # logits of every decode step: teacher-force the oracle's ids one token at a time
    gm.reset()
    gm.forward(prompt[:-1][:16], np.arange(16), lm_head=False)
    gm.forward(prompt[:-1][16:], np.arange(16, 20), lm_head=False)
    cur = int(prompt[-1])
    worst = 0.0
    for s in range(steps):
        lg, am = gm.forward([cur], [gm.position], lm_head=True)
        worst = max(worst, rel_err(lg[0], want_logits[s]))
        assert int(am[0]) == int(want_ids[s])
        cur = int(want_ids[s])
    # every kernel reproduces the reference's accumulation order: logits are bit-exact, not merely < 1e-3
    assert worst == 0.0, worst
    # KV cache contents (K rows, transposed V) match the reference layout
    n = gm.position
    assert np.array_equal(gm.k_cache(0)[:n], om.k_cache(0)[:n])
    assert np.array_equal(gm.v_cache(1)[:, :n], om.v_cache(1)[:, :n])
    gm.close()
    om.close()


def test_one_launch_attention_is_bit_identical(ctx, oracle, tmp_path):
    """Single-token attention as ONE launch (attn_decode2: scores stored write-through, per-kv-head ticket rendezvous; the
    default) and as two launches (mode bit 4): the same ids and logits as the oracle, eager and hipGraph replay, past a few
    position groups so that several workgroups contribute scores."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 12, n_ctx=256, seed=11)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=8)
    prompt = np.random.default_rng(5).integers(0, cfg.vocab_size, 150)
    want_ids, want_logits, *_ = om.generate(prompt, 32, 10, want_logits=True)
    for mode in (0, 1, 16, 17):  # default = one launch (attn_decode2); 16: two launches; +1: eager
        gm = hip.Model(ctx, d, max_batch=32)
        gm.set_mode(mode)
        assert np.array_equal(gm.generate(prompt, 32, 10), want_ids), mode
        gm.reset()
        done = 0
        while done < prompt.size - 1:
            bs = min(32, prompt.size - 1 - done)
            gm.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
            done += bs
        cur = int(prompt[-1])
        for s in range(10):
            lg, am = gm.forward([cur], [gm.position], lm_head=True)
            assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want_logits[s]).view(np.uint32)), (mode, s)
            cur = int(want_ids[s])
        gm.close()
    om.close()


@pytest.mark.parametrize("preset,n_ctx,P", [("small-llama-hs128", 4096, 2090), ("small-llama-hs128", 2048, 1031), ("small-llama", 4096, 2215),
                                            ("small-llama-hs128", 160, 97), ("llama-1b-dims-2l", 1024, 515), ("small-llama-hs128", 1024, 31)])
def test_decode_attention_long_cache_matches_oracle(ctx, oracle, tmp_path, preset, n_ctx, P):
    """The single-token attention (round 6: fused with the Q / K / V mat-vec into qkv_attn_kernel where the shape allows it -- head size 128 and, with
    llama-1b-dims-2l, 64; the 31-token prompt walks the new position across a 128-byte line of the V rows and a group of 8 K rows -- otherwise attn_decode2:
    scores exchanged through a counter rendezvous, soft-max in registers, V.p on the matrix cores) behind a LONG cache — several position rounds per workgroup, hinted and un-hinted K rows, both gather
    trips, the n_kv % 8 / n_kv % 32 leftovers walking through all their values — bit-exact against the oracle, eager and
    hipGraph replay, and identical to the two-launch kernels (mode bit 4) also with hidden cache slots (kv_mask), which the
    CPU executor cannot express."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, 12, n_ctx=n_ctx, seed=9)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=16)
    prompt = np.random.default_rng(17).integers(0, cfg.vocab_size, P)
    steps = 40
    want_ids, want_logits, *_ = om.generate(prompt, 128, steps, want_logits=True)
    assert ctx.L.ps_hip_debug_set(7, 1) == 0  # (the fused QKV + attention launch wherever it is covered: also its head-size-64 instance)
    gm = hip.Model(ctx, d, max_batch=128, n_ctx=n_ctx)
    assert np.array_equal(gm.generate(prompt, 128, steps), want_ids)           # hipGraph replay
    gm.set_mode(1)
    assert np.array_equal(gm.generate(prompt, 128, steps), want_ids)           # eager launches
    gm.set_mode(0)
    gm.kv_rollback(steps)                                                      # back to the end of the prompt
    cur = int(prompt[-1])
    for s in range(steps):                                                     # eager single-token forwards: exact host hint
        lg, am = gm.forward([cur], [gm.position], lm_head=True)
        assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want_logits[s]).view(np.uint32)), s
        cur = int(want_ids[s])
    # hidden slots: the two attention plans agree bit for bit (and differ from the unmasked logits)
    hidden = [3, 40, P // 2, P - 2]
    outs = []
    for mode in (0, 128, 16):  # the fused QKV + attention launch / its two launches (QKV mat-vec, attn_decode2) / scores + soft-max.V.p behind the QKV mat-vec
        gm.set_mode(mode)
        gm.kv_rollback(5)
        for h in hidden:
            gm.kv_mask(h, False)
        res = []
        for s in range(5):
            lg, am = gm.forward([int(want_ids[steps - 6 + s])], [gm.position], lm_head=True)  # (teacher forcing: step k is fed id k - 1)
            res.append(np.asarray(lg[0]).copy())
        for h in hidden:
            gm.kv_mask(h, True)
        outs.append(np.stack(res))
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)) and np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
    assert not np.array_equal(outs[0][0], np.asarray(want_logits[steps - 5]))                     # (unmasked, these would be step steps - 5's logits)
    assert ctx.L.ps_hip_debug_set(7, 0) == 0
    gm.close()
    om.close()


@pytest.mark.parametrize("preset", ["small-llama-hs128", "llama-1b-dims-2l"])
def test_fused_qkv_attention_from_an_empty_cache(ctx, oracle, tmp_path, preset):
    """qkv_attn_kernel (k_qkvattn.hip: the Q / K / V mat-vec and the single-token attention in ONE launch) one token at a time from position 0: no cached row at
    all, then positions walking through the first groups of 8 K rows and across the first 128-byte lines of the V rows (32, 64) -- what it asks of the cache before
    its rendezvous (rows below the position, whole lines below it) and what after (the new row, the line with the new column) changes at every one of them.
    Logits on bits against the oracle, the same from the two launches it replaces (mode bit 7), and the captured step from an empty cache."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, 12, n_ctx=256, seed=13)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=16)
    toks = np.random.default_rng(2).integers(0, cfg.vocab_size, 70)
    want = [om.forward([int(t)], [i], True)[0] for i, t in enumerate(toks)]
    assert ctx.L.ps_hip_debug_set(7, 1) == 0  # (the head-size-64 instance fills half the chip and is not dispatched by default: taken here wherever it is covered)
    gm = hip.Model(ctx, d, max_batch=8, n_ctx=256)
    for mode in (1, 129):  # eager: fused / two launches
        gm.set_mode(mode)
        gm.reset()
        for i, t in enumerate(toks):
            lg, _ = gm.forward([int(t)], [i], lm_head=True)
            assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want[i]).view(np.uint32)), (mode, i)
    om.reset()
    want_ids, *_ = om.generate(toks[:1], 8, 40)
    gm.set_mode(0)
    gm.reset()
    assert np.array_equal(gm.generate(toks[:1], 8, 40), want_ids)  # hipGraph replay from position 0
    assert ctx.L.ps_hip_debug_set(7, 0) == 0
    gm.close()
    om.close()


@pytest.mark.parametrize("preset,wt", [("small-llama-hs128", 12), ("llama-1b-dims-2l", 12), ("llama-1b-dims-2l", 2), ("tiny-qwen2", 8)])
def test_kv_cache_policy_hint_changes_no_bit(ctx, oracle, tmp_path, preset, wt):
    """The single-token attention reads the cached K rows / V channels with plain or with non-temporal loads (and, streaming, its new K row alone) depending on the cache's
    size (csrc/model.hip, ps_hip_debug_set(9, v)): both forms of the fused Q / K / V + attention launch and of the attention launch of its own, forced, against the oracle --
    a prompt whose length puts the new position at several places of its group of eight K rows, per-step logits through eager single-token forwards and greedy ids through
    the captured step."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=256, seed=29)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=16)
    prompt = np.random.default_rng(8).integers(0, cfg.vocab_size, 45)
    steps = 21
    want_ids, want_logits, *_ = om.generate(prompt, 32, steps, want_logits=True)
    assert ctx.L.ps_hip_debug_set(7, 1) == 0  # (the fused launch wherever it is covered)
    gm = hip.Model(ctx, d, max_batch=32, n_ctx=256)
    try:
        for force in (0, 1, -1):
            assert ctx.L.ps_hip_debug_set(9, force) == 0
            for mode in (0, 128):  # fused where covered / the two launches; set_mode drops the captured step, the next one is captured with this policy
                gm.set_mode(16 | mode); gm.set_mode(mode)
                gm.reset()
                assert np.array_equal(gm.generate(prompt, 32, steps), want_ids), (force, mode)
                gm.kv_rollback(steps)
                cur = int(prompt[-1])
                for s_ in range(steps):
                    lg, _ = gm.forward([cur], [gm.position], lm_head=True)
                    assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want_logits[s_]).view(np.uint32)), (force, mode, s_)
                    cur = int(want_ids[s_])
    finally:
        ctx.L.ps_hip_debug_set(9, -1); ctx.L.ps_hip_debug_set(7, 0)
    gm.close()
    om.close()


@pytest.mark.parametrize("preset,wt,P", [("llama-8b-dims-4l", 12, 161), ("llama-8b-dims-4l", 1015, 140), ("llama-1b-dims-2l", 2, 300), ("qwen2-0.5b-dims-2l", 8, 300)])
def test_real_layer_shapes_match_oracle(ctx, oracle, tmp_path, preset, wt, P):
    """The BASELINE.json configurations at their REAL layer dimensions (a few layers, a 4096-token vocabulary): Llama-3.1-8B
    (4096 / 14336, 32 / 8 heads of 128; pure Q4_K and the Q4_K_M mix), Llama-3.2-1B (2048 / 8192, 32 / 8 heads of 64; Q4_0) and
    Qwen2-0.5B (896 / 4864, 14 / 2 heads of 64, biases, NEOX; Q8_0) — prefill in chunks of 128 (a full chunk and a ragged one),
    greedy ids through the captured step, and every step's logits through eager single-token forwards, bit-exact against
    the oracle.  These are the shapes the production kernels are dispatched for (gemv4, gemm4k, attn_decode2, ...)."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=512, seed=77)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=min(32, os.cpu_count() or 8))
    prompt = np.random.default_rng(4).integers(0, cfg.vocab_size, P)
    steps = 8
    want_ids, want_logits, *_ = om.generate(prompt, 128, steps, want_logits=True)
    gm = hip.Model(ctx, d, max_batch=128, n_ctx=512)
    assert np.array_equal(gm.generate(prompt, 128, steps), want_ids)
    gm.kv_rollback(steps)
    cur = int(prompt[-1])
    for s in range(steps):
        lg, am = gm.forward([cur], [gm.position], lm_head=True)
        assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want_logits[s]).view(np.uint32)), (s, rel_err(lg[0], want_logits[s]))
        cur = int(want_ids[s])
    # a batch with logits behind the prompt (the chunk kernels with lm_head)
    om.rollback(steps)
    gm.kv_rollback(steps)
    toks = np.random.default_rng(5).integers(0, cfg.vocab_size, 9)
    want = om.forward(toks, np.arange(P - 1, P + 8), True)
    got, _ = gm.forward(toks, np.arange(P - 1, P + 8), lm_head=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rel_err(got, want)
    gm.close()
    om.close()


def test_tree_mask_plumbing(ctx, tmp_path):
    """The batch forward takes an optional [bs][bs] tree mask (speculative verify; the reference's CPU
    executor ignores mask objects, executor.cpp:210-224, so there is no CPU golden for a real tree).  A tree mask that
    IS the causal triangle must reproduce the causal forward bit for bit, and a branching tree must make a node blind to
    its sibling: the sibling's logits equal those of the chain without it."""
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "small-llama-hs128", 12, n_ctx=64, seed=4)
    gm = hip.Model(ctx, d, max_batch=16)
    rng = np.random.default_rng(8)
    prompt = rng.integers(0, gm.cfg.vocab_size, 6)
    toks = rng.integers(0, gm.cfg.vocab_size, 5)

    def run(tokens, pos, tree):
        gm.reset()
        gm.forward(prompt, np.arange(6), lm_head=False)
        lg, _ = gm.forward(tokens, pos, lm_head=True, tree=tree)
        return lg

    causal = run(toks, np.arange(6, 11), None)
    tri = np.tril(np.ones((5, 5), dtype=np.uint8))
    assert np.array_equal(run(toks, np.arange(6, 11), tri).view(np.uint32), causal.view(np.uint32))
    # nodes 0 -> 1 -> 2 and a node 3 that hangs off node 0 (rows = who each node may see among the batch).  Cache slots /
    # RoPE positions stay consecutive, as the reference's KV append requires (norm_attention.cpp:82-91).
    tree = np.zeros((4, 4), dtype=np.uint8)
    tree[0, 0] = 1
    tree[1, [0, 1]] = 1
    tree[2, [0, 1, 2]] = 1
    tree[3, [0, 3]] = 1
    lg = run(toks[:4], np.arange(6, 10), tree)
    chain4 = run(toks[:4], np.arange(6, 10), None)
    assert np.array_equal(lg[:3].view(np.uint32), chain4[:3].view(np.uint32))   # same visible sets
    assert not np.array_equal(lg[3], chain4[3])                                  # node 3 is blind to nodes 1 and 2
    gm.close()


@pytest.mark.parametrize("preset,wt,chunk,max_batch", [("small-llama-hs128", 12, 128, 512), ("small-llama-hs128", 1015, 64, 256), ("small-llama-hs128", 12, 96, 500),
                                                       ("tiny-llama", 8, 128, 384), ("tiny-qwen2", 2, 32, 128), ("small-llama-hs128", 12, 128, 128)])
def test_prefill_in_super_chunks_keeps_the_reference_chunking(ctx, oracle, tmp_path, preset, wt, chunk, max_batch):
    """ps_hip_model_prefill: several reference-sized chunks per launch sequence (mat-muls over all their columns, attention per chunk)
    must leave exactly the cache the reference's chunk-by-chunk prefill leaves: K rows and V columns bit-equal to the oracle run
    chunk by chunk, and so are the logits of the steps that follow (ragged last chunk, a prefix already in the cache, a chunk size that
    does not divide max_batch, max_batch == chunk)."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=1536, seed=11)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=16)
    gm = hip.Model(ctx, d, max_batch=max_batch, n_ctx=1536)
    rng = np.random.default_rng(3)
    pre = rng.integers(0, cfg.vocab_size, 5)  # a few tokens already in the cache
    om.forward(pre, np.arange(5), False); gm.forward(pre, np.arange(5), lm_head=False)
    P = 3 * max(chunk, max_batch // 2) + 2 * chunk + 17
    prompt = rng.integers(0, cfg.vocab_size, P)
    done = 0
    while done < P:
        bs = min(chunk, P - done)
        om.forward(prompt[done:done + bs], np.arange(5 + done, 5 + done + bs), False)
        done += bs
    gm.prefill(prompt, chunk)
    assert gm.position == om.position == 5 + P
    kv = cfg.kv_dim
    for L in (0, cfg.n_layers - 1):
        gk, gv = gm.k_cache(L), gm.v_cache(L)
        ok, ov = om.k_cache(L), om.v_cache(L)
        assert np.array_equal(gk[:5 + P].view(np.uint32), ok[:5 + P].view(np.uint32)), L
        assert np.array_equal(gv[:, :5 + P].view(np.uint32), ov[:, :5 + P].view(np.uint32)), L
    cur = int(prompt[-1])
    for s in range(3):
        want1 = om.forward([cur], [5 + P + s], True)
        got1, am1 = gm.forward([cur], [5 + P + s], True)
        assert np.array_equal(got1.view(np.uint32), want1.view(np.uint32)), (s, rel_err(got1, want1))
        cur = int(am1[0])
    gm.close()
    om.close()


@pytest.mark.parametrize("preset,wt,chunk", [("small-llama-hs128", 12, 256), ("small-llama-hs128", 12, 512), ("small-llama-hs128", 1015, 384), ("tiny-llama", 8, 256), ("tiny-qwen2", 2, 320)])
def test_wide_prefill_chunks_match_oracle(ctx, oracle, tmp_path, preset, wt, chunk):
    """hparams batch_size above the reference's default of 128 (bench.py reports a 512-token-chunk prefill next to the headline):
    chunks of 256 .. 512 columns, a ragged last one, logits of every column of the last chunk bit-equal to the oracle run with the
    same chunking (the soft-max row length n_kv = end of the chunk depends on it, in the reference too)."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=1024, seed=9)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=16)
    gm = hip.Model(ctx, d, max_batch=chunk, n_ctx=1024)
    P = 2 * chunk - 37 if chunk > 256 else 2 * chunk + 91
    prompt = np.random.default_rng(7).integers(0, cfg.vocab_size, P)
    done = 0
    while done < P:
        bs = min(chunk, P - done)
        last = done + bs == P
        want = om.forward(prompt[done:done + bs], np.arange(done, done + bs), last)
        got = gm.forward(prompt[done:done + bs], np.arange(done, done + bs), last)
        if last:
            assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32)), rel_err(got[0], want)
        done += bs
    cur = int(got[1][-1])
    for s in range(3):  # and single tokens behind it
        want1 = om.forward([cur], [P + s], True)
        got1, am1 = gm.forward([cur], [P + s], True)
        assert np.array_equal(got1.view(np.uint32), want1.view(np.uint32)), (s, rel_err(got1, want1))
        cur = int(am1[0])
    gm.close()
    om.close()


@pytest.mark.parametrize("wt", [12, 1015, 1017])  # pure Q4_K; the Q4_K_M mix (Q6_K attn_v / ffn_down / output: the prefill chunks take gemm6k); Q5_K_M (Q5_K producer + gemm6k)
def test_long_context_batches_match_oracle(ctx, oracle, tmp_path, wt):
    """Batches appended behind a long KV prefix (n_kv > 256: several 32-column chain rounds, leftovers, softmax tails):
    bit-exact against the oracle for decode steps and for batches of 3 and 12, and a tree whose root only sees itself
    gives the root the logits of the causal batch (the children are masked exactly like future tokens)."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "small-llama-hs128", wt, n_ctx=512, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=16)
    gm = hip.Model(ctx, d, max_batch=128, n_ctx=512)
    P = 250
    prompt = np.random.default_rng(42).integers(0, cfg.vocab_size, P)
    for mdl in (om, gm):
        done = 0
        while done < P - 1:
            bs = min(128, P - 1 - done)
            mdl.forward(prompt[done:done + bs], np.arange(done, done + bs), False)
            done += bs
    cur = int(prompt[-1])
    for s in range(14):  # pos0 249..262: single-token steps, then a batch of 3 and of 12 at the same position (rolled back)
        p0 = gm.position
        for bs in (3, 12):
            toks = np.array([cur] + [(7 * s + u) % cfg.vocab_size for u in range(1, bs)])
            want = om.forward(toks, np.arange(p0, p0 + bs), True)
            om_pos = om.position
            got, am = gm.forward(toks, np.arange(p0, p0 + bs), True)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (s, bs, rel_err(got, want))
            tree = np.eye(bs, dtype=np.uint8); tree[:, 0] = 1
            rp = np.array([p0] + [p0 + 1] * (bs - 1))
            gm.kv_rollback(bs)
            tl, _ = gm.forward_tree(toks, rp, tree, lm_head=True, want_logits=True, advance=False)
            assert np.array_equal(tl[0].view(np.uint32), want[0].view(np.uint32)), (s, bs)
            om.rollback(bs)
            assert om.position == om_pos - bs == gm.position
        want1 = om.forward([cur], [p0], True)
        got1, am1 = gm.forward([cur], [p0], True)
        assert np.array_equal(got1.view(np.uint32), want1.view(np.uint32)), (s, rel_err(got1, want1))
        cur = int(am1[0])
    gm.close()
    om.close()


def test_batched_forward_logits(ctx, oracle, tmp_path):
    """lm_head over a whole batch (LlamaModel::forward returns vocab x bs logits) + batch-size invariance."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 12, n_ctx=64)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, "llama", load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    gm = hip.Model(ctx, d, max_batch=16)
    toks = np.arange(3, 14) % cfg.vocab_size
    want = om.forward(toks, np.arange(toks.size), True)
    got, am = gm.forward(toks, np.arange(toks.size), True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rel_err(got, want)
    assert np.array_equal(am, want.argmax(1))
    gm.reset()
    got1 = np.concatenate([gm.forward([t], [i], True)[0] for i, t in enumerate(toks)])
    assert rel_err(got1, want) < 1e-3  # one-token-at-a-time changes n_kv per row (different vector/tail split in softmax): close, not identical
    gm.close()
    om.close()


@pytest.mark.parametrize("preset,n_ctx", [("small-llama-hs128", 6144), ("tiny-llama", 8192), ("small-llama", 4096)])
def test_batches_with_a_large_context_window(ctx, oracle, tmp_path, preset, n_ctx):
    """n_ctx beyond 4096 takes the workgroup-per-(kv head, column) soft-max (score rows in LDS) instead of the wave-per-row one,
    and head sizes other than 64 / 128 the direct-fetch V.p kernel: chunks of 40 and 9 tokens, then single tokens, bit-exact."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, 12, n_ctx=n_ctx, seed=11)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=8)
    gm = hip.Model(ctx, d, max_batch=64, n_ctx=n_ctx)
    toks = np.random.default_rng(3).integers(0, cfg.vocab_size, 52)
    for lo, hi in ((0, 40), (40, 49)):
        want = om.forward(toks[lo:hi], np.arange(lo, hi), True)
        got, _ = gm.forward(toks[lo:hi], np.arange(lo, hi), True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (lo, rel_err(got, want))
    for i in range(49, 52):
        want = om.forward(toks[i:i + 1], [i], True)
        got, _ = gm.forward(toks[i:i + 1], [i], True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (i, rel_err(got, want))
    gm.close()
    om.close()


@pytest.mark.parametrize("preset,wt,n_ctx", [("small-llama-hs128", 8, 96), ("small-llama-hs128", 12, 64), ("small-llama", 2, 96), ("tiny-qwen2", 8, 68)])
def test_context_windows_shorter_than_one_attention_step(ctx, oracle, tmp_path, preset, wt, n_ctx):
    """n_ctx below 128 slots (not a multiple of 32 either): the batch V.p kernel stages 128 positions per step, and a lane whose segment lies past the
    row's last whole block must not compute its address from that segment (round 5: with its own column index it read up to 124 bytes past the last
    row of a layer's V cache -- found by tools/gpu_fuzz.py as a memory fault when that allocation ended a mapping).  A chunk of 41, one of 9, then single
    tokens up to the last slot; logits and cache rows on bits."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=n_ctx, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    gm = hip.Model(ctx, d, max_batch=41)
    toks = np.random.default_rng(n_ctx).integers(0, cfg.vocab_size, n_ctx)
    for lo, hi in ((0, 41), (41, 50)):
        want = om.forward(toks[lo:hi], np.arange(lo, hi), True)
        got, _ = gm.forward(toks[lo:hi], np.arange(lo, hi), True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (lo, rel_err(got, want))
    for i in range(50, n_ctx):
        want = om.forward(toks[i:i + 1], [i], True)
        got, _ = gm.forward(toks[i:i + 1], [i], True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (i, rel_err(got, want))
    L = cfg.n_layers - 1
    assert np.array_equal(gm.k_cache(L), om.k_cache(L)) and np.array_equal(gm.v_cache(L), om.v_cache(L))
    with pytest.raises(hip.PSHipError):
        gm.forward([1], [n_ctx], True)  # the window is full
    gm.close()
    om.close()


def test_kv_full_and_bad_tokens_fail_loudly(ctx, tmp_path):
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "tiny-llama", 8, n_ctx=16)
    gm = hip.Model(ctx, d, max_batch=8)
    with pytest.raises(hip.PSHipError):
        gm.forward([1, 2, 3], [14, 15, 16], lm_head=False)      # runs past n_ctx
    with pytest.raises(hip.PSHipError):
        gm.forward([99999], [0], lm_head=False)                  # token id out of range
    with pytest.raises(hip.PSHipError):
        gm.forward([1, 2], [0, 2], lm_head=False)                # non-consecutive positions
    gm.close()


def attention_reference(m, n_kv):
    """softmax(q K^T / sqrt(hs)) V of the LAST layer in float64 from the model's own q and FP32 caches"""
    c = m.cfg
    hs, r2 = c.head_size, c.n_heads // c.n_kv_heads
    q = m.scratch(1, 1)[0].astype(np.float64).reshape(c.n_heads, hs)
    K = m.k_cache(c.n_layers - 1)[:n_kv].astype(np.float64).reshape(n_kv, c.n_kv_heads, hs)
    V = m.v_cache(c.n_layers - 1).astype(np.float64)[:, :n_kv].reshape(c.n_kv_heads, hs, n_kv)
    out = np.empty((c.n_heads, hs))
    for h in range(c.n_heads):
        s = K[:, h // r2, :] @ q[h] / np.sqrt(hs)
        p = np.exp(s - s.max()); p /= p.sum()
        out[h] = V[h // r2] @ p
    return out.reshape(-1)


@pytest.mark.parametrize("preset,wt,n_prompt,n_ctx", [("small-llama-hs128", 12, 150, 256), ("tiny-llama", 8, 37, 256), ("small-llama", 2, 90, 256),
                                                    ("small-llama-hs128", 12, 2101, 4096)])  # the bench's context: chunks of 66 positions per split
def test_fp16_kv_decode_mode_is_close_to_parity(ctx, tmp_path, preset, wt, n_prompt, n_ctx):
    """ps_hip_model_set_mode bit 3 — fp16 mirrors of K and V, split-KV online soft-max for the single-token
    attention.  Deliberately not bit-exact; the tolerances are stated here:
      * the attention output itself (last layer, against a float64 soft-max over the model's own q and FP32 caches) is
        within 2e-3 of the largest output, where the parity kernel is within 1e-6;
      * per-step logits (teacher-forced on the parity ids) stay within 8e-2 of the largest logit — the int8 activation
        quantizers downstream turn any perturbation of the attention output into rounding flips (DESIGN.md section 5 has
        the same effect between batch rows and single-token steps) — and the arg-max agrees wherever the parity margin
        exceeds that;
      * batches / prefill are identical bit for bit (they read the FP32 cache), and the mode cannot be entered on a used cache."""
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, preset, wt, n_ctx=n_ctx, seed=6)
    m = hip.Model(ctx, d, max_batch=64)
    rng = np.random.default_rng(2)
    prompt = rng.integers(0, m.cfg.vocab_size, n_prompt)

    def prefill():
        m.reset()
        for lo in range(0, n_prompt - 1, 64):
            hi = min(lo + 64, n_prompt - 1)
            m.forward(prompt[lo:hi], np.arange(lo, hi), lm_head=False)

    prefill()
    lb32, _ = m.forward(prompt[:9], np.arange(m.position, m.position + 9), lm_head=True)
    prefill()
    steps, cur, want, ids = 24, int(prompt[-1]), [], []
    for s in range(steps):
        lg, am = m.forward([cur], [m.position], lm_head=True)
        if s in (0, steps - 1):
            ref = attention_reference(m, m.position)
            assert np.abs(m.scratch(2, 1)[0] - ref).max() / np.abs(ref).max() < 1e-6
        want.append(lg[0].copy()); ids.append(int(am[0])); cur = int(am[0])
    assert m.ctx.L.ps_hip_model_set_mode(m.h, 8) != 0  # cache in use
    m.reset()
    assert m.ctx.L.ps_hip_model_set_mode(m.h, 8) == 0
    prefill()
    lb16, _ = m.forward(prompt[:9], np.arange(m.position, m.position + 9), lm_head=True)  # a batch: FP32 cache, FP32 kernels
    assert np.array_equal(lb16.view(np.uint32), lb32.view(np.uint32))
    prefill()
    cur, worst, worst_att = int(prompt[-1]), 0.0, 0.0
    for s in range(steps):
        lg, am = m.forward([cur], [m.position], lm_head=True)
        ref = attention_reference(m, m.position)
        worst_att = max(worst_att, float(np.abs(m.scratch(2, 1)[0] - ref).max() / np.abs(ref).max()))
        worst = max(worst, float(np.abs(lg[0] - want[s]).max() / np.abs(want[s]).max()))
        top2 = np.sort(want[s])[-2:]
        if (top2[1] - top2[0]) / np.abs(want[s]).max() > 8e-2:
            assert int(am[0]) == ids[s], s
        cur = ids[s]
    assert 1e-6 < worst_att < 2e-3, worst_att  # close, and really a different path
    assert worst < 8e-2, worst
    # the decode loop (hipGraph) takes the same path
    prefill()
    got = m.decode_greedy(int(prompt[-1]), 8)
    prefill()
    cur = int(prompt[-1])
    for s in range(8):
        _, am = m.forward([cur], [m.position], lm_head=True)
        assert int(am[0]) == int(got[s])
        cur = int(am[0])
    m.close()


@pytest.mark.parametrize("preset,wt", [("small-llama-hs128", 12), ("small-llama-hs128", 1015), ("tiny-qwen2", 2), ("tiny-llama", 8)])
def test_fp16_prefill_perf_mode_is_close_to_parity(ctx, oracle, tmp_path, preset, wt):
    """SURVEY 8 f4, second half: ps_hip_model_set_mode bit 5 -- the layer mat-muls of prefill batches as dense fp16 GEMMs on dequantized
    fp16 copies of the weights (csrc/perf16.hip's own v_mfma_f32_32x32x16_f16 GEMM), fp32 accumulation; no activation quantizer, no per-block fp32 chains.  Deliberately NOT
    bit-exact; the tolerances are stated here:
      * layer 0's V cache against a float64 evaluation of the same model (dequantized weights, exact RMSNorm, no activation rounding at
        all) is within 2e-3 of the largest |entry| -- fp16 rounding of the operands -- where the parity path, which rounds the activations
        to int8 as the reference does, sits near 1e-2: the mode is the more accurate evaluation of the quantized model, not a cheaper one;
      * the last layer's K rows / V columns are within 8e-2 of the largest |entry| of the parity prefill's and the logits of the parity
        steps that follow within 2e-1 of the largest |logit| (random synthetic weights: the int8 rounding the mode skips is an error of
        that size against the fp32 model, compounding over the layers);
      * single tokens and batches WITH logits are bit-identical to the parity path while the mode is on (they do not take it)."""
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, preset, wt, n_ctx=512, seed=6)
    ref, pm = hip.Model(ctx, d, max_batch=128, n_ctx=512), hip.Model(ctx, d, max_batch=128, n_ctx=512)
    pm.set_mode(32)
    rng = np.random.default_rng(4)
    n = 300
    prompt = rng.integers(0, ref.cfg.vocab_size, n)
    ref.prefill(prompt, 128); pm.prefill(prompt, 128)

    T = load_tensors(os.path.join(d, "ggml/weights.gguf"))
    dim, kvd = ref.cfg.dim, ref.cfg.kv_dim
    et, eb, _, _ = T["token_embd.weight"]
    xe = oracle.get_embedding(et, eb, dim, prompt).astype(np.float64)
    nw = T["blk.0.attn_norm.weight"][1].view(np.float32).astype(np.float64)
    xn = xe / np.sqrt((xe * xe).mean(axis=1, keepdims=True) + ref.cfg.norm_eps) * nw
    vt, vb, _, _ = T["blk.0.attn_v.weight"]
    Wv = oracle.dequantize(vt, vb, kvd * dim).astype(np.float64).reshape(kvd, dim)
    V64 = xn @ Wv.T
    if "blk.0.attn_v.bias" in T:
        V64 += T["blk.0.attn_v.bias"][1].view(np.float32).astype(np.float64)
    e_perf = np.abs(pm.v_cache(0)[:, :n].T - V64).max() / np.abs(V64).max()
    e_par = np.abs(ref.v_cache(0)[:, :n].T - V64).max() / np.abs(V64).max()
    print(f"layer-0 V against float64: perf mode {e_perf:.2e}, parity path {e_par:.2e}")
    assert e_perf <= 2e-3 and e_perf < e_par, (e_perf, e_par)

    L = ref.cfg.n_layers - 1
    for a, b in ((ref.k_cache(L)[:n], pm.k_cache(L)[:n]), (ref.v_cache(L)[:, :n], pm.v_cache(L)[:, :n])):
        assert np.isfinite(b).all()
        assert np.abs(a - b).max() <= 8e-2 * np.abs(a).max(), np.abs(a - b).max() / np.abs(a).max()
    cur, worst = int(prompt[-1]), 0.0
    for s in range(4):
        lr, ar = ref.forward([cur], [n + s], True)
        lp, ap = pm.forward([cur], [n + s], True)
        worst = max(worst, float(np.abs(lr - lp).max() / np.abs(lr).max()))
        cur = int(ar[0])
    assert worst <= 2e-1, worst
    # what does not take the mode is bit-identical: a batch with logits and a single token on a fresh cache
    ref.reset(); pm.reset()
    la, _ = ref.forward(prompt[:9], np.arange(9), True); lb, _ = pm.forward(prompt[:9], np.arange(9), True)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    la, _ = ref.forward([5], [9], True); lb, _ = pm.forward([5], [9], True)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    ref.close(); pm.close()


def test_bench_runs_under_rccl_world_of_one(tmp_path):
    """bench.py --force-dist: torch.distributed / RCCL is initialised for a single rank and the prompt broadcast, the id
    all-gather and the max-over-ranks reduction run on the GPU (the N > 1 code path, loaded under the driver
    even where only one GPU is there).  A small model keeps it to seconds; the JSON line must carry the contract's fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TMPDIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1", "--preset", "small-llama-hs128", "--wtype", "Q4_K",
                        "--prompt-len", "40", "--n-ctx", "128", "--batch", "32", "--no-cpu-baseline", "--no-kv-f16", "--no-graph-path"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]  # (libraries may still chat on stdout while they shut down)
    assert len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["replicas_agree"] is True and line["value"] > 0
    assert line["roofline"]["kernel"] and "prefill_roofline" in line
