"""The product's token tree (powerserve_amd/csrc/host/speculative.cpp, TokenTree::draft / verify / iterate) against the
reference's own src/speculative/token_tree.cpp:

  * tests/golden/token_tree.npz — recorded from the real reference (compiled into oracle/_ref) driven by scripted models
    (oracle/ref_token_tree.cpp, generator oracle/gen_golden_spec.py).  For every case the product's tree, run over the
    Python twins of those scripted models (tests/spec_script.py), must create the same nodes in the same order (token,
    position, parent, attention mask), make the same model / KV-cache calls in the same order, and emit the same tokens.
  * the same comparison live against oracle/_ref on seeded random configurations, where the reference was built.

The scripted logits depend on the set of cache entries a token can see, so the emitted tokens also check that the
masks, moves and advances MEAN the same thing on both sides.  No GPU involved: the tree is host logic."""
import os

import numpy as np
import pytest

from spec_script import FORWARD_TREE, ScriptedModel, normalize_reference_events

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "token_tree.npz")
CFG_KEYS = ("draft_batch_size", "top_k", "max_fan_out", "early_stop", "temperature", "p_base", "min_prob")
SCR_KEYS = ("shared_seed", "target_seed", "draft_seed", "shared_w", "target_w", "draft_w", "vocab", "n_ctx")


def run_product(cfg, scr, prefix, iters, root_token=1):
    from powerserve_amd import host
    c = host.SpecConfig.make(int(cfg["draft_batch_size"]), int(cfg["top_k"]), int(cfg["max_fan_out"]), bool(cfg["early_stop"]),
                             float(cfg["temperature"]), float(cfg["p_base"]), float(cfg["min_prob"]))
    log = []
    mk = lambda i, own, w: ScriptedModel(i, int(scr["n_ctx"]), log, int(scr["shared_seed"]), int(own), scr["shared_w"], w, int(scr["vocab"]), prefix)
    target, draft = mk(0, scr["target_seed"], scr["target_w"]), mk(1, scr["draft_seed"], scr["draft_w"])
    tokens, trees, stats = host.token_tree_run(target, draft, c, root_token, iters)
    return tokens, trees, stats, log, (target, draft)


def compare(ref, tokens, trees, stats, log, bs):
    """ref: dict with tokens / tree / masks / events as recorded from the reference"""
    assert tokens.tolist() == ref["tokens"].tolist()
    for it, t in enumerate(trees):
        n = len(t)
        assert t[:, :3].tolist() == ref["tree"][it, :n].tolist(), f"iteration {it}: node order / token / position / parent"
        # the reference keeps batch_size nodes; the ones it never filled are blank (token 0, position 0, no parent)
        assert (ref["tree"][it, n:] == np.array([0, 0, -1])).all()
        mask = np.zeros((n, n), np.uint8)
        for u in range(n):
            x = u
            while x != -1:
                mask[u, x] = 1
                x = t[x, 2]
        assert (mask == ref["masks"][it, :n, :n]).all()
        # accepted flags form one root-to-leaf path
        acc = np.flatnonzero(t[:, 4])
        assert acc[0] == 0 and all(t[v, 2] in acc for v in acc[1:])
    want = normalize_reference_events(ref["events"])
    assert len(log) == len(want)
    it = 0
    for got, exp in zip(log, want):
        if exp[0] == 0 and exp[1] == FORWARD_TREE:
            assert exp[2] == bs and got[2] == len(trees[it]) and got[3] == exp[3]  # the product's batch is the nodes that exist
            it += 1
        else:
            assert tuple(got) == exp
    assert stats["n_iterations"] == len(trees) and stats["n_generated_tokens"] == len(tokens)
    assert stats["n_draft_tokens"] == sum(len(t) - 1 for t in trees)
    assert stats["n_accepted_tokens"] == len(tokens) - len(trees)


def golden_cases():
    z = np.load(GOLDEN)
    return sorted({k.split("/")[0] for k in z.files})


@pytest.mark.parametrize("name", golden_cases())
def test_token_tree_matches_reference_golden(name):
    z = np.load(GOLDEN)
    cfg, scr = dict(zip(CFG_KEYS, z[f"{name}/cfg"])), dict(zip(SCR_KEYS, z[f"{name}/script"]))
    tokens, trees, stats, log, _ = run_product(cfg, scr, z[f"{name}/prefix"], int(z[f"{name}/iters"][0]))
    compare({k: z[f"{name}/{k}"] for k in ("tokens", "tree", "masks", "events")}, tokens, trees, stats, log, int(cfg["draft_batch_size"]))


@pytest.mark.parametrize("seed", range(6))
def test_token_tree_matches_reference_live(ref, seed):
    from oracle import binding as B
    if not hasattr(ref.L, "ref_token_tree_run"):
        pytest.skip("oracle/_ref was built without the token tree")
    rng = np.random.default_rng(100 + seed)
    cfg = dict(draft_batch_size=int(rng.integers(2, 20)), top_k=int(rng.integers(1, 20)), max_fan_out=int(rng.integers(1, 5)), early_stop=int(rng.integers(0, 2)),
               temperature=float(np.float32(rng.uniform(0.5, 3.0))), p_base=float(np.float32(rng.uniform(0.3, 1.0))), min_prob=float(np.float32(rng.uniform(0.0, 0.4))))
    scr = dict(shared_seed=int(rng.integers(1, 2**40)), target_seed=int(rng.integers(1, 2**40)), draft_seed=int(rng.integers(1, 2**40)),
               shared_w=float(np.float32(rng.uniform(20, 120))), target_w=float(np.float32(rng.uniform(0, 20))), draft_w=float(np.float32(rng.uniform(0, 20))),
               vocab=int(rng.integers(20, 200)), n_ctx=512)
    prefix, iters = rng.integers(0, scr["vocab"], int(rng.integers(0, 12))), 10
    r = B.ref_token_tree_run(ref, B.SpecConfig(*[cfg[k] for k in CFG_KEYS]), B.Script(*[scr[k] for k in SCR_KEYS]), prefix, 1, iters)
    tokens, trees, stats, log, _ = run_product(cfg, scr, prefix, iters)
    compare(r, tokens, trees, stats, log, cfg["draft_batch_size"])


def test_caches_agree_after_generation():
    """after any number of rounds both models hold exactly the emitted sequence in slots 0..position-1, all visible"""
    z = np.load(GOLDEN)
    name = "default_close"
    cfg, scr = dict(zip(CFG_KEYS, z[f"{name}/cfg"])), dict(zip(SCR_KEYS, z[f"{name}/script"]))
    prefix = z[f"{name}/prefix"]
    tokens, _, _, _, (target, draft) = run_product(cfg, scr, prefix, 8)
    seq = list(prefix) + [1] + tokens.tolist()[:-1]  # the last emitted token is the next root, not yet in either cache
    for m in (target, draft):
        assert m.position == len(seq)
        assert m.tok[:m.position].tolist() == seq and m.pos[:m.position].tolist() == list(range(len(seq))) and m.vis[:m.position].all()
