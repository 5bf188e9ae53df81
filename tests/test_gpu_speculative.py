"""Speculative token-tree decoding (src/speculative/token_tree.cpp, spec_model.hpp) on the HIP backend.

The reference's CPU executor ignores mask objects (executor.cpp:210-224), so the reference cannot run a tree forward as
a whole; its compiled operators can (oracle/ref_ops_forward.py sequences them with a caller-supplied mask), and that is
what test_tree_forward_all_nodes_match_reference_operators pins every tree node's logits to, bit for bit.  The drivers
are checked through a size-independent property: with a greedy target sampler the speculative output IS the target
model's own greedy output, whatever the draft model proposes.  Checked with an unrelated draft model (nearly nothing
accepted: exercises catch-up forwards, branch switching with hidden cache slots, KV moves) and with the target as its own
draft (long accepted paths)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def teacher_forced_gaps(model, prompt, emitted):
    """Feed `emitted` back through SINGLE-token forwards: per step (top logit - logit of the emitted token, logit std)."""
    model.reset()
    model.forward(prompt[:-1], np.arange(len(prompt) - 1), lm_head=False)
    cur, gaps, stds = int(prompt[-1]), [], []
    for t in emitted:
        lg = model.forward([cur], [model.position], lm_head=True)[0]
        gaps.append(float(lg.max() - lg[int(t)]))
        stds.append(float(lg.std()))
        cur = int(t)
    return np.array(gaps), np.array(stds)


@pytest.mark.parametrize("draft,draft_seed,wt", [("small-llama-hs128", 77, 12), ("small-llama-hs128", 5, 12), ("small-llama-hs128", 5, 8),
                                               ("small-llama-draft", 9, 12), ("small-llama-draft", 9, 2)])
def test_speculative_output_follows_target_greedy(tmp_path, draft, draft_seed, wt):
    """BASELINE config #4 in miniature (target + smaller draft of a DIFFERENT shape, and same-shape pairs): the output of
    SpeculativeModel::generate (csrc/host/speculative.cpp) is the target's own greedy output.  Where the two differ the
    documented bound must hold (KV entries of accepted nodes come from the tree batch, whose softmax
    row split differs from single-token decode, as in the reference): every emitted token lies within 0.25 sigma of the
    single-token arg-max under teacher forcing — and on these models, whose margins are healthy, the ids are equal."""
    from powerserve_amd import host, synth
    td, dd = str(tmp_path / "t"), str(tmp_path / "d")
    synth.write_model_dir(td, "small-llama-hs128", wt if draft == "small-llama-hs128" else 12, n_ctx=160, seed=5)
    synth.write_model_dir(dd, draft, wt, n_ctx=160, seed=draft_seed)
    target, dm = host.HostModel(td, max_batch=16), host.HostModel(dd, max_batch=16)
    prompt = np.random.default_rng(11).integers(0, target.vocab, 13)
    steps = 40
    want = target.generate(prompt, 8, steps)
    got, st = host.spec_generate(target, dm, prompt, 8, steps)
    gaps, stds = teacher_forced_gaps(target, prompt, got)
    assert (gaps <= 0.25 * stds).all(), (gaps.max(), stds.mean())
    assert np.array_equal(got, want), (got, want)
    assert st["n_generated_tokens"] >= steps and st["n_iterations"] > 0
    if draft == "small-llama-hs128" and draft_seed == 5:  # the target drafts for itself: several tokens per iteration
        assert st["n_generated_tokens"] / st["n_iterations"] > 2.0, st
    else:                                                  # unrelated draft: catch-up forwards, branch switches, KV moves
        assert st["n_draft_times"] >= st["n_iterations"]
    # the plain path still works on the same objects afterwards (no hidden slots left behind in the visible prefix)
    assert np.array_equal(target.generate(prompt, 8, steps), want)
    # ... and so does the DRAFT model: slots it hid while drafting lie behind its position after the roll-back
    # (KVCache::rollback + advance/append un-hide them in the reference, core/kv_cache.hpp:249-272)
    fresh = host.HostModel(dd, max_batch=16)
    ref_d = fresh.generate(prompt, 8, 12)
    fresh.close()
    assert np.array_equal(dm.generate(prompt, 8, 12), ref_d)
    target.close()
    dm.close()


def test_speculative_verify_through_the_sampler_chain(tmp_path):
    """spec_model.hpp:105 hands the SAMPLER to TokenTree::verify (token_tree.cpp:214-216: sampler.apply on the node's logits,
    then the top candidate): a top_k = 1 chain reproduces the greedy text, a stop token ends the text where it is emitted, a
    cache too small for another tree ends it early instead of aborting, and a penalised chain is reproducible."""
    from powerserve_amd import host, synth
    td = str(tmp_path / "t")
    synth.write_model_dir(td, "small-llama-hs128", 12, n_ctx=96, seed=5)
    target, dm = host.HostModel(td, max_batch=16), host.HostModel(td, max_batch=16)
    prompt = np.random.default_rng(11).integers(0, target.vocab, 13)
    steps = 40
    want = target.generate(prompt, 8, steps)
    smp = host.Sampler(host.SamplerCfg.make(target.vocab, top_k=1))
    got, st = host.spec_generate_sampled(target, dm, prompt, 8, steps, smp)
    assert np.array_equal(got, want) and st["n_generated_tokens"] / st["n_iterations"] > 2.0, (got, want, st)
    # stop token: the first emitted occurrence ends the text (inclusive)
    eos = int(want[9])
    first = int(np.nonzero(want == eos)[0][0])
    got, _ = host.spec_generate_sampled(target, dm, prompt, 8, steps, smp, eos=eos)
    assert np.array_equal(got, want[:first + 1])
    # capacity: 96 slots hold the 13-token prompt and at most 96 - 12 - 12 further tokens before a 12-node tree no longer fits
    got, _ = host.spec_generate_sampled(target, dm, prompt, 8, 500, smp)
    assert 40 <= got.size <= 96 - 12 and np.array_equal(got[:steps], want)
    smp.close()
    # a chain with penalties: deterministic, and not the greedy text
    kw = dict(top_k=1, penalty_repeat=1.8, penalty_last_n=16, penalty_freq=0.3)
    a, b = host.Sampler(host.SamplerCfg.make(target.vocab, **kw)), host.Sampler(host.SamplerCfg.make(target.vocab, **kw))
    ga, _ = host.spec_generate_sampled(target, dm, prompt, 8, steps, a)
    gb, _ = host.spec_generate_sampled(target, dm, prompt, 8, steps, b)
    assert np.array_equal(ga, gb) and ga.size == steps
    a.close(); b.close()
    assert np.array_equal(target.generate(prompt, 8, steps), want)  # both models are left usable
    target.close(); dm.close()



def test_speculative_sampler_history_is_never_advanced(tmp_path):
    """The reference's speculative iterator never calls sampler.accept (only Model::decode does, llama_model.cpp:128;
    SpecTokenIterator::decode / TokenTree::verify, token_tree.cpp:181-234, do not): a penalised chain therefore sees its INITIAL history
    at every verified node.  The speculative text under penalty_repeat != 1 equals single-token steps whose logits go through a chain
    that has accepted nothing -- and differs from the text of a chain that accepts what it emits."""
    from powerserve_amd import host, synth
    td = str(tmp_path / "t")
    synth.write_model_dir(td, "small-llama-hs128", 12, n_ctx=96, seed=5)
    target, dm = host.HostModel(td, max_batch=16), host.HostModel(td, max_batch=16)
    prompt = np.random.default_rng(11).integers(0, target.vocab, 13)
    steps = 24
    kw = dict(top_k=1, penalty_repeat=1.8, penalty_last_n=16, penalty_freq=0.3)
    smp = host.Sampler(host.SamplerCfg.make(target.vocab, **kw))
    got, _ = host.spec_generate_sampled(target, dm, prompt, 8, steps, smp)
    smp.close()

    def plain(accepting):
        target.reset()
        target.forward(prompt[:-1], np.arange(prompt.size - 1), lm_head=False)
        cur, out = int(prompt[-1]), []
        chain = host.Sampler(host.SamplerCfg.make(target.vocab, **kw))
        for s in range(steps):
            lg = target.forward([cur], [prompt.size - 1 + s])[0]
            if not accepting:  # a chain that has accepted nothing: a fresh one per step
                chain.close()
                chain = host.Sampler(host.SamplerCfg.make(target.vocab, **kw))
            cur = chain.sample(lg)
            out.append(cur)
        chain.close()
        return np.array(out, dtype=np.int32)

    never, always = plain(False), plain(True)
    target.close(); dm.close()
    assert np.array_equal(got, never), (got, never)
    if np.array_equal(never, always):
        pytest.skip("the penalties never bite on this text: the two behaviours cannot be told apart here")


def test_tree_forward_positions_and_masks(ctx, tmp_path):
    """Column i of a tree forward is rotated with its own RoPE position and sees only its ancestors: a two-branch tree
    reproduces, per branch, the logits of that branch run as a plain causal chain (same cache slots in the same order
    inside the softmax for the first branch; the second branch differs only by the slot its tokens sit in)."""
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "small-llama-hs128", 12, n_ctx=64, seed=4)
    gm = hip.Model(ctx, d, max_batch=16)
    rng = np.random.default_rng(8)
    prompt = rng.integers(0, gm.cfg.vocab_size, 6)
    t = rng.integers(0, gm.cfg.vocab_size, 5)

    def prefill():
        gm.reset()
        gm.forward(prompt, np.arange(6), lm_head=False)

    # tree: 0 -> 1 -> 2 and 0 -> 3 -> 4   (positions 6,7,8 and 6,7,8)
    tree = np.zeros((5, 5), dtype=np.uint8)
    for u, anc in enumerate([[0], [0, 1], [0, 1, 2], [0, 3], [0, 3, 4]]):
        tree[u, anc] = 1
    prefill()
    lg, _ = gm.forward_tree(t, [6, 7, 8, 7, 8], tree, want_logits=True)
    prefill()
    a, _ = gm.forward(t[[0, 1, 2]], np.arange(6, 9), lm_head=True)
    prefill()
    b, _ = gm.forward(t[[0, 3, 4]], np.arange(6, 9), lm_head=True)
    assert np.array_equal(lg[:3].view(np.uint32), a.view(np.uint32))
    # second branch: identical visible set and RoPE positions, but its keys sit in slots 9,10 instead of 7,8: the
    # scores meet the same values in a different order inside the 8-wide softmax groups -> equal to rounding
    from conftest import rel_err
    assert rel_err(lg[3], b[1]) < 1e-5 and rel_err(lg[4], b[2]) < 1e-5
    assert np.argmax(lg[4]) == np.argmax(b[2])
    # hidden cache slots: hiding a prompt slot changes the result, showing it again restores it bit for bit
    prefill()
    base, _ = gm.forward_tree([int(t[0])], [6], None, want_logits=True)
    gm.kv_mask(2, False)
    hid, _ = gm.forward_tree([int(t[0])], [6], None, want_logits=True)
    gm.kv_mask(2, True)
    back, _ = gm.forward_tree([int(t[0])], [6], None, want_logits=True)
    assert not np.array_equal(base, hid)
    assert np.array_equal(base.view(np.uint32), back.view(np.uint32))
    gm.close()


def _sha(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", [0, 16])
def test_tree_forward_all_nodes_match_reference_operators(ctx, oracle, tmp_path, ci, mode):
    """SURVEY 8 f1 pinned: ps_hip_model_forward_tree for a BRANCHING 12-node tree behind a prefix with two hidden cache
    slots and RoPE positions = prefix + depth gives, for EVERY node, the logits that the reference's own operators give
    under that mask (tests/golden/tree_forward.npz = oracle/_ref via oracle/ref_ops_forward.py) and that the restatement
    gives live — bit for bit; then the accepted path is compacted (kv_move), the cache advanced, and one token decoded
    behind the hidden slots (the single-token attention kernels; mode 16 = the two-launch plan) — bit for bit again.
    Cases: Llama Q8_0, Qwen2 (bias, NEOX) Q4_0, head size 128 Q4_K, the Q4_K_M mix."""
    from oracle import binding as B
    from powerserve_amd import gguf, hip, synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tree_forward.npz"))
    k = f"c{ci}_"
    d = str(tmp_path / "m")
    n_ctx, prefix = int(g[k + "n_ctx"]), g[k + "prefix"]
    P = len(prefix)
    mj = synth.write_model_dir(d, str(g[k + "preset"]), int(g[k + "wt"]), n_ctx=n_ctx, seed=int(g[k + "seed"]))
    path = os.path.join(d, "ggml", "weights.gguf")
    assert _sha(path) == str(g[k + "gguf_sha256"])
    rd = gguf.GGUFReader(path)
    tensors = {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}
    om = oracle.model(B.make_config(mj["llm_config"]), mj["model_arch"], tensors, n_threads=8)
    gm = hip.Model(ctx, d, max_batch=32, n_ctx=n_ctx)
    gm.set_mode(mode)
    done = 0
    while done < P:
        bs = min(32, P - done)
        om.forward(prefix[done:done + bs], np.arange(done, done + bs), False)
        gm.forward(prefix[done:done + bs], np.arange(done, done + bs), lm_head=False)
        done += bs
    kv_vis = np.ones(n_ctx, dtype=np.uint8)
    kv_vis[g[k + "hidden"]] = 0
    for h in g[k + "hidden"]:
        gm.kv_mask(int(h), False)
    toks, rope, tree = g[k + "tokens"], g[k + "rope"], g[k + "tree"]
    want = om.forward_tree(toks, rope, tree, kv_vis, True, advance=False)
    got, am = gm.forward_tree(toks, rope, tree, lm_head=True, want_logits=True, advance=False)
    assert np.array_equal(want.view(np.uint32), g[k + "logits"].view(np.uint32))
    assert np.array_equal(got.view(np.uint32), g[k + "logits"].view(np.uint32)), [i for i in range(12) if not np.array_equal(got[i], g[k + "logits"][i])]
    assert np.array_equal(am, np.argmax(g[k + "logits"], axis=1))
    for L in range(gm.cfg.n_layers):  # the appended K rows / V columns
        assert np.array_equal(gm.k_cache(L)[:P + 12].view(np.uint32), om.k_cache(L)[:P + 12].view(np.uint32))
        assert np.array_equal(gm.v_cache(L)[:, :P + 12].view(np.uint32), om.v_cache(L)[:, :P + 12].view(np.uint32))
    acc = g[k + "accept"]
    for u, a in enumerate(acc):
        if a != u:
            gm.kv_move(P + u, P + int(a))
    gm.kv_advance(len(acc))
    step, _ = gm.forward_tree(g[k + "next"], [P + len(acc)], None, lm_head=True, want_logits=True, advance=False)
    assert np.array_equal(step.view(np.uint32), g[k + "step_logits"].view(np.uint32))
    gm.close()
    om.close()
