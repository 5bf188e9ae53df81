"""Parity at the HEADLINE's launch geometry ("close the two parity holes"):

(a) the Llama-3.1-8B layer shape (8 kv heads of 128) behind a cache of more than 2048 positions -- the exact
    attn_decode2 grid (256 workgroups) and the 17th+ prefill chunk the bench times, compared with the oracle bit for bit;
(b) BASELINE config #4 at real layer dimensions (8B-shaped target + 1B-shaped draft): the token tree of csrc/host/speculative.cpp is run
    twice through the SAME seven-call interface (CallbackSpecBackend) -- once over the oracle's models (pso_model_forward_tree + kv_move,
    i.e. the reference's operators in the reference's order), once over the HIP models -- and must emit the same ids, build the same
    trees and see the same logits at every tree node of every iteration (src/speculative/token_tree.cpp:181-234).
"""
import os

import numpy as np
import pytest

from conftest import load_tensors

pytestmark = pytest.mark.gpu


def test_headline_attention_geometry_matches_oracle(ctx, oracle, tmp_path):
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "llama-8b-dims-4l", 12, n_ctx=4096, seed=31)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=min(48, os.cpu_count() or 8))
    P, steps = 2101, 8  # 16 full chunks of 128 + a ragged 17th (52 columns); steps at n_kv 2101 .. 2108 (n_kv % 8 and % 32 leftovers)
    prompt = np.random.default_rng(23).integers(0, cfg.vocab_size, P)
    want_ids, want_logits, *_ = om.generate(prompt, 128, steps, want_logits=True)
    gm = hip.Model(ctx, d, max_batch=512, n_ctx=4096)
    assert np.array_equal(gm.generate(prompt, 128, steps), want_ids)  # prefill chunk by chunk + the captured step (hipGraph replay)
    n = P - 1
    for L in (0, 3):  # the cache the 17 chunks left behind: K rows and transposed V
        assert np.array_equal(gm.k_cache(L)[:n].view(np.uint32), om.k_cache(L)[:n].view(np.uint32)), L
        assert np.array_equal(gm.v_cache(L)[:, :n].view(np.uint32), om.v_cache(L)[:, :n].view(np.uint32)), L
    gm.kv_rollback(steps)
    cur = int(prompt[-1])
    for s in range(steps):  # eager single-token forwards (exact host hint for the K prefetch): every logit
        lg, am = gm.forward([cur], [gm.position], lm_head=True)
        assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want_logits[s]).view(np.uint32)), s
        cur = int(want_ids[s])
    # the bench's prefill entry (four reference chunks per launch sequence) leaves the same cache and the same next logits
    gm.reset()
    gm.prefill(prompt[:-1], 128)
    assert np.array_equal(gm.k_cache(3)[:n].view(np.uint32), om.k_cache(3)[:n].view(np.uint32))
    lg, _ = gm.forward([int(prompt[-1])], [gm.position], lm_head=True)
    assert np.array_equal(np.asarray(lg[0]).view(np.uint32), np.asarray(want_logits[0]).view(np.uint32))
    gm.close()
    om.close()


class _SpecSide:
    """The seven calls of SpecBackend (csrc/host/speculative.hpp) over a model object; records what every tree forward returned."""

    def __init__(self, vocab):
        self.vocab_size = vocab
        self.tree_logits, self.one_logits = [], []


class OracleSide(_SpecSide):
    def __init__(self, om):
        super().__init__(om.cfg.vocab_size)
        self.om = om
        self.vis = np.ones(om.cfg.seq_len, dtype=np.uint8)  # ps_hip_model_kv_mask's table, kept by hand: the oracle's forward takes it per call

    def _kv(self):
        return None if self.vis.all() else self.vis

    def kv_position(self):
        return self.om.position

    def forward_one(self, token, position, want):  # HIPSpecBackend::forward_one: a one-column tree forward that advances
        p = self.om.position
        lg = self.om.forward_tree([token], [position], None, self._kv(), lm_head=want, advance=True)
        self.vis[p] = 1
        if want:
            self.one_logits.append(lg[0].copy())
            return lg[0]
        return None

    def forward_tree(self, tokens, positions, mask):  # advance = 0: the accepted path is kept with kv_move + kv_advance
        lg = self.om.forward_tree(tokens, positions, mask, self._kv(), lm_head=True, advance=False)
        self.tree_logits.append(lg.copy())
        return np.argmax(lg, axis=1)

    def kv_mask(self, slot, visible):
        self.vis[slot] = 1 if visible else 0

    def kv_move(self, dst, src):
        self.om.kv_move(dst, src)

    def kv_advance(self, n):
        p = self.om.position
        self.vis[p:p + n] = 1
        self.om.kv_advance(n)

    def kv_rollback(self, n):
        self.om.rollback(n)
        self.vis[self.om.position:] = 1


class HipSide(_SpecSide):
    def __init__(self, gm):
        super().__init__(gm.cfg.vocab_size)
        self.gm = gm

    def kv_position(self):
        return self.gm.position

    def forward_one(self, token, position, want):
        lg, _ = self.gm.forward_tree([token], [position], None, lm_head=want, want_logits=want, advance=True)
        if want:
            self.one_logits.append(lg[0].copy())
            return lg[0]
        return None

    def forward_tree(self, tokens, positions, mask):
        lg, am = self.gm.forward_tree(tokens, positions, mask, lm_head=True, want_logits=True, advance=False)
        self.tree_logits.append(lg.copy())
        return am

    def kv_mask(self, slot, visible):
        self.gm.kv_mask(slot, visible)

    def kv_move(self, dst, src):
        self.gm.kv_move(dst, src)

    def kv_advance(self, n):
        self.gm.kv_advance(n)

    def kv_rollback(self, n):
        self.gm.kv_rollback(n)


@pytest.mark.parametrize("pair", ["8b+1b", "self"])
def test_speculative_loop_at_real_dimensions_equals_the_oracles(ctx, oracle, tmp_path, pair):
    from oracle import binding as B
    from powerserve_amd import hip, host, synth
    n_ctx, P, iters = 256, 70, 6
    td, dd = str(tmp_path / "t"), str(tmp_path / "d")
    mt = synth.write_model_dir(td, "llama-8b-dims-4l", 12, n_ctx=n_ctx, seed=41)
    if pair == "self":  # the target drafts for itself: long accepted paths, several kv_moves per iteration
        dd, md = td, mt
    else:               # an unrelated 1B-shaped Q4_0 draft: catch-up forwards, hidden slots, branch switches
        md = synth.write_model_dir(dd, "llama-1b-dims-2l", 2, n_ctx=n_ctx, seed=43)
    nt = min(48, os.cpu_count() or 8)
    ot = oracle.model(B.make_config(mt["llm_config"]), mt["model_arch"], load_tensors(os.path.join(td, "ggml/weights.gguf")), n_threads=nt)
    od = oracle.model(B.make_config(md["llm_config"]), md["model_arch"], load_tensors(os.path.join(dd, "ggml/weights.gguf")), n_threads=nt)
    gt, gd = hip.Model(ctx, td, max_batch=128, n_ctx=n_ctx), hip.Model(ctx, dd, max_batch=128, n_ctx=n_ctx)
    prompt = np.random.default_rng(3).integers(0, ot.cfg.vocab_size, P)
    pos = np.arange(P - 1)
    for m in (ot, od):
        m.forward(prompt[:-1], pos, lm_head=False)
    for m in (gt, gd):
        m.forward(prompt[:-1], pos, lm_head=False)
    cfg = host.SpecConfig.make()  # the reference's defaults (speculative_config.hpp:21-36): a 12-node tree
    o_t, o_d, h_t, h_d = OracleSide(ot), OracleSide(od), HipSide(gt), HipSide(gd)
    want, want_trees, want_st = host.token_tree_run(o_t, o_d, cfg, int(prompt[-1]), iters)
    got, got_trees, got_st = host.token_tree_run(h_t, h_d, cfg, int(prompt[-1]), iters)
    assert np.array_equal(got, want), (got, want)
    assert got_st == want_st, (got_st, want_st)
    assert len(got_trees) == len(want_trees) == iters
    for it in range(iters):
        assert np.array_equal(got_trees[it], want_trees[it]), it
        assert np.array_equal(h_t.tree_logits[it].view(np.uint32), o_t.tree_logits[it].view(np.uint32)), it  # EVERY node of the verify batch
    assert len(h_d.one_logits) == len(o_d.one_logits) and len(h_d.one_logits) > 0
    for a, b in zip(h_d.one_logits, o_d.one_logits):  # every draft step
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    n = ot.position
    assert gt.position == n and gd.position == od.position
    for L in (0, 3):  # the caches after the moves
        assert np.array_equal(gt.k_cache(L)[:n].view(np.uint32), ot.k_cache(L)[:n].view(np.uint32))
        assert np.array_equal(gt.v_cache(L)[:, :n].view(np.uint32), ot.v_cache(L)[:, :n].view(np.uint32))
    if pair == "self":
        assert want_st["n_generated_tokens"] / want_st["n_iterations"] > 1.5, want_st
    # ... and what the loop emits is NOT always plain greedy, in the oracle either ("lossless"): report how far the two agree
    ot.rollback(ot.position - (P - 1))
    plain, cur = [], int(prompt[-1])
    for s in range(len(want)):
        lg = ot.forward([cur], [P - 1 + s], True)[0]
        cur = int(np.argmax(lg))
        plain.append(cur)
    agree = int(np.argmin(np.append(np.array(plain) == want, False)))
    print(f"[{pair}] oracle speculative ids vs oracle plain greedy: first {agree} of {len(want)} equal")
    for m in (gt, gd):
        m.close()
    for m in (ot, od):
        m.close()


def test_thirty_two_layers_behind_a_long_cache(ctx, oracle, tmp_path):
    """The headline's DEPTH times a long cache (round-4 review, weak 3: the full-size tests above use 4- and 2-layer models, the bench's own parity object a
    4-token prompt): 32 layers of Q4_K at a width the CPU oracle finishes in seconds (dim 1024 / hidden 2048: the mat-vec and chunk mat-mul kernels of the 8B
    path, head size 128, 4 heads per kv head), a 700-token prompt prefilled in super-chunks of four 128-token reference chunks, then single-token steps
    behind 700 cached positions (n_kv % 32 and % 8 leftovers), eager and through the captured greedy loop -- every logit of every step and the last layer's cache rows, bit for bit."""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "deep-llama-hs128", 12, n_ctx=768, seed=3)
    cfg = B.make_config(mj["llm_config"])
    assert cfg.n_layers == 32
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=min(48, os.cpu_count() or 8))
    prompt = np.random.default_rng(1).integers(0, cfg.vocab_size, 700)
    steps = 6
    want_ids, want_logits, *_ = om.generate(prompt, 128, steps, want_logits=True)
    gm = hip.Model(ctx, d, max_batch=512, n_ctx=768)
    gm.prefill(prompt[:-1], 128)
    assert gm.position == 699
    n = 699
    assert np.array_equal(gm.k_cache(31)[:n].view(np.uint32), om.k_cache(31)[:n].view(np.uint32))
    assert np.array_equal(gm.v_cache(31)[:, :n].view(np.uint32), om.v_cache(31)[:, :n].view(np.uint32))
    cur = int(prompt[-1])
    for s in range(steps):
        lg, am = gm.forward([cur], [gm.position], lm_head=True)
        assert np.array_equal(lg[0].view(np.uint32), np.asarray(want_logits[s], dtype=np.float32).view(np.uint32)), s
        assert int(am[0]) == int(want_ids[s])
        cur = int(want_ids[s])
    # and the captured greedy loop from the same state
    gm.kv_rollback(steps)
    assert np.array_equal(gm.decode_greedy(int(prompt[-1]), steps), want_ids)
    gm.close()
    om.close()
