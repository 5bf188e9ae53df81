"""CPU: the plain-C restatement (oracle/ps_oracle.c) against the committed golden vectors produced by the REAL
reference (oracle/gen_golden.py).  Runs anywhere (no GPU, no /root/reference).  The restatement reproduces the
reference's AVX2 lane structure, so everything here is BIT-EXACT."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
T = {"Q4_0": 2, "Q8_0": 8, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_act_quant_bit_exact(oracle):
    g = np.load(os.path.join(GOLD, "act_quant.npz"))
    for K in (256, 896, 4096):
        x = g[f"x_{K}"]
        for i in range(x.shape[0]):
            assert np.array_equal(oracle.from_float(8, x[i]), g[f"q8_0_{K}"][i])
            if K % 256 == 0:
                assert np.array_equal(oracle.from_float(15, x[i]), g[f"q8_K_{K}"][i])


@pytest.mark.parametrize("name,count", [("mul_mat", 12), ("mul_mat_wide", 16)])
def test_mul_mat_bit_exact(oracle, name, count):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = 0
    for key in g.files:
        if not key.startswith("y_"):
            continue
        _, tn1, tn2, K, N, bs = key.split("_")
        t = T[tn1 + "_" + tn2]
        y = oracle.mul_mat(t, g["w_" + key[2:]], int(K), int(N), g["x_" + key[2:]])
        assert np.array_equal(bits(y), bits(g[key])), key
        n += 1
    assert n == count


def test_ops_bit_exact(oracle):
    from oracle import binding as B
    g = np.load(os.path.join(GOLD, "ops.npz"))
    assert np.array_equal(bits(oracle.rms_norm(g["rms_x"], g["rms_w"], 1e-6)), bits(g["rms_y"]))
    for mode, hs, base in ((0, 64, 1e4), (2, 64, 1e6), (0, 128, 5e5)):
        rp = B.RopeParams(hs, 4096, base, 1.0, 0.0, 1.0, 32.0, 0.0, mode)
        y = oracle.rope(g[f"rope_x_{mode}_{hs}"], g[f"rope_pos_{mode}_{hs}"], rp)
        assert np.array_equal(bits(y), bits(g[f"rope_y_{mode}_{hs}"]))
    for n_kv in (1, 33, 300):
        y = oracle.softmax_ext(g[f"sm_x_{n_kv}"], g[f"sm_mask_{n_kv}"], 0.125)
        assert np.array_equal(bits(y), bits(g[f"sm_y_{n_kv}"]))
    assert np.array_equal(bits(oracle.silu_hadamard(g["silu_g"], g["silu_u"])), bits(g["silu_y"]))


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for c in iter(lambda: f.read(1 << 20), b""):
            h.update(c)
    return h.hexdigest()


@pytest.mark.parametrize("preset,tn", [("tiny-llama", "Q4_0"), ("tiny-llama", "Q8_0"), ("tiny-qwen2", "Q8_0"), ("tiny-qwen2", "Q4_0")])
def test_e2e_bit_exact(oracle, tmp_path, preset, tn):
    """whole-model forward of the restatement == real LlamaModel/Qwen2Model::forward, logits bit-for-bit"""
    from oracle import binding as B
    from powerserve_amd import gguf, synth
    g = np.load(os.path.join(GOLD, f"e2e_{preset}_{tn}.npz"))
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, T[tn], n_ctx=int(g["n_ctx"]), seed=int(g["seed"]))
    path = os.path.join(d, "ggml", "weights.gguf")
    assert _sha(path) == str(g["gguf_sha256"]), "synthetic model generator drifted: regenerate tests/golden (oracle/gen_golden.py)"
    rd = gguf.GGUFReader(path)
    tensors = {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}
    m = oracle.model(B.make_config(mj["llm_config"]), mj["model_arch"], tensors, n_threads=4)
    ids, logits, *_ = m.generate(g["prompt"], 8, 24, want_logits=True)
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(bits(logits), bits(g["logits"]))
    m.reset()
    assert np.array_equal(bits(m.forward(g["prompt"][:9], np.arange(9), True)), bits(g["batch_logits"]))
    m.close()


@pytest.mark.parametrize("preset,tn", [("tiny-llama", "Q4_0"), ("tiny-llama", "Q8_0"), ("tiny-qwen2", "Q8_0"), ("tiny-qwen2", "Q4_0")])
def test_e2e_both_reference_builds(oracle, tmp_path, preset, tn):
    """tests/golden/e2e_builds_*.npz (oracle/gen_golden_fast.py): the real forward of the reference built with -ffp-contract=off AND as its own CMake
    builds it (GCC's default -ffp-contract=fast).  The oracle's default mode is the first, pso_set_contract(1) the second -- every logit, bit for bit;
    and the distance between the two builds is what the fixture says (2.8e-3 of the largest logit on tiny-llama Q4_0, nothing on the other three)."""
    from oracle import binding as B
    from powerserve_amd import gguf, synth
    g = np.load(os.path.join(GOLD, f"e2e_builds_{preset}_{tn}.npz"))
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, T[tn], n_ctx=int(g["n_ctx"]), seed=int(g["seed"]))
    path = os.path.join(d, "ggml", "weights.gguf")
    assert _sha(path) == str(g["gguf_sha256"]), "synthetic model generator drifted: regenerate tests/golden (oracle/gen_golden_fast.py)"
    rd = gguf.GGUFReader(path)
    tensors = {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}
    try:
        for mode, name in ((0, "off"), (1, "fast")):
            oracle.L.pso_set_contract(mode)
            m = oracle.model(B.make_config(mj["llm_config"]), mj["model_arch"], tensors, n_threads=4)
            ids, logits, *_ = m.generate(g["prompt"], int(g["batch"]), len(g["ids_" + name]), want_logits=True)
            m.close()
            assert np.array_equal(ids, g["ids_" + name]), name
            assert np.array_equal(bits(logits), bits(g["logits_" + name])), name
    finally:
        oracle.L.pso_set_contract(0)
    dev = float(np.abs(g["logits_fast"] - g["logits_off"]).max() / np.abs(g["logits_off"]).max())
    assert np.array_equal(g["ids_off"], g["ids_fast"])
    assert (dev > 1e-3) == (preset == "tiny-llama" and tn == "Q4_0") and dev < 1e-2, dev


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_tree_forward_golden(oracle, tmp_path, ci):
    """Token-tree forward of the restatement == the reference's own operators given the tree mask
    (tests/golden/tree_forward.npz, oracle/gen_golden_tree.py): every node's logits, then the compacted cache + one decoded
    token behind hidden slots — bit for bit."""
    from oracle import binding as B
    from powerserve_amd import gguf, synth
    g = np.load(os.path.join(GOLD, "tree_forward.npz"))
    k = f"c{ci}_"
    d = str(tmp_path / "m")
    n_ctx, P = int(g[k + "n_ctx"]), len(g[k + "prefix"])
    mj = synth.write_model_dir(d, str(g[k + "preset"]), int(g[k + "wt"]), n_ctx=n_ctx, seed=int(g[k + "seed"]))
    path = os.path.join(d, "ggml", "weights.gguf")
    assert _sha(path) == str(g[k + "gguf_sha256"]), "synthetic model generator drifted: regenerate (oracle/gen_golden_tree.py)"
    rd = gguf.GGUFReader(path)
    tensors = {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}
    m = oracle.model(B.make_config(mj["llm_config"]), mj["model_arch"], tensors, n_threads=4)
    done = 0
    while done < P:
        bs = min(32, P - done)
        m.forward(g[k + "prefix"][done:done + bs], np.arange(done, done + bs), False)
        done += bs
    kv_vis = np.ones(n_ctx, dtype=np.uint8)
    kv_vis[g[k + "hidden"]] = 0
    got = m.forward_tree(g[k + "tokens"], g[k + "rope"], g[k + "tree"], kv_vis, True, advance=False)
    assert np.array_equal(bits(got), bits(g[k + "logits"]))
    acc = g[k + "accept"]
    for u, a in enumerate(acc):
        if a != u:
            m.kv_move(P + u, P + int(a))
    m.kv_advance(len(acc))
    step = m.forward_tree(g[k + "next"], [P + len(acc)], None, kv_vis, True, advance=False)
    assert np.array_equal(bits(step), bits(g[k + "step_logits"]))
    m.close()


def test_scaled_rope_golden(oracle, tmp_path):
    """rope_freq_scale / rope_attn_factor off 1.0 (src/core/config.cpp:96,98 -> ggml.c:15344-15358; tests/golden/rope_scaled.npz made by the REAL reference,
    oracle/gen_golden_rope_scaled.py): the operator in both rotation modes and whole-model generations, bit for bit."""
    import importlib.util
    from oracle import binding as B
    from powerserve_amd import gguf, synth
    spec = importlib.util.spec_from_file_location("gen_rs", os.path.join(os.path.dirname(GOLD), "..", "oracle", "gen_golden_rope_scaled.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = np.load(os.path.join(GOLD, "rope_scaled.npz"))
    for i, (mode, hs, base, fs, af) in enumerate(gen.OPS):
        y = oracle.rope(g[f"op{i}_x"], g[f"op{i}_pos"], B.RopeParams(hs, 4096, base, fs, 0.0, af, 32.0, 0.0, mode))
        assert np.array_equal(bits(y), bits(g[f"op{i}_y"])), i
        plain = oracle.rope(g[f"op{i}_x"], g[f"op{i}_pos"], B.RopeParams(hs, 4096, base, 1.0, 0.0, 1.0, 32.0, 0.0, mode))
        assert not np.array_equal(bits(y), bits(plain))  # (the parameters do something)
    for i, (preset, t, fs, af) in enumerate(gen.E2E):
        d = str(tmp_path / f"m{i}")
        mj = synth.write_model_dir(d, preset, t, n_ctx=128, seed=777 + i, rope_freq_scale=fs, rope_attn_factor=af)
        path = os.path.join(d, "ggml", "weights.gguf")
        assert _sha(path) == str(g[f"e{i}_gguf_sha256"]), "synthetic model generator drifted: regenerate (oracle/gen_golden_rope_scaled.py)"
        rd = gguf.GGUFReader(path)
        tensors = {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}
        m = oracle.model(B.make_config(mj["llm_config"]), mj["model_arch"], tensors, n_threads=4)
        ids, logits, *_ = m.generate(g[f"e{i}_prompt"], 8, 20, want_logits=True)
        m.close()
        assert np.array_equal(ids, g[f"e{i}_ids"]), i
        assert np.array_equal(bits(logits), bits(g[f"e{i}_logits"])), i
