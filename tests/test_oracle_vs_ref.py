"""CPU, dev container only: restatement vs the live reference library (oracle/_ref), randomized, incl. Q4_K/Q5_K/Q6_K
which the reference can only run at op level (its loader aborts on K-quants).  Skipped where _ref is absent."""
import os

import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("t", [2, 8, 12, 13, 14])
@pytest.mark.parametrize("K,N,bs", [(256, 40, 1), (1024, 24, 5), (4096, 16, 2)])
def test_mul_mat(oracle, ref, t, K, N, bs):
    from powerserve_amd import synth
    rng = np.random.default_rng(t + K + bs)
    w = synth.random_blocks(rng, t, N, K)
    x = (rng.standard_normal((bs, K)) * rng.choice([0.01, 1, 30])).astype(np.float32)
    yr, ar = ref.mul_mat(t, w, K, N, x, want_act=True)
    yo, ao = oracle.mul_mat(t, w, K, N, x, want_act=True)
    assert np.array_equal(ar, ao)          # activation blocks
    assert np.array_equal(bits(yr), bits(yo))
    # weights produced by the reference's own quantizer (realistic scale statistics)
    wf = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    wq = ref.quantize(t, wf)
    assert np.array_equal(bits(ref.mul_mat(t, wq, K, N, x)), bits(oracle.mul_mat(t, wq, K, N, x)))
    assert np.array_equal(ref.dequantize(t, wq[: ref.row_size(t, K)], K), oracle.dequantize(t, wq[: oracle.row_size(t, K)], K))


def test_thread_count_invariance(ref):
    from oracle import binding as B
    from powerserve_amd import synth
    rng = np.random.default_rng(5)
    w = synth.random_blocks(rng, 12, 64, 1024)
    x = rng.standard_normal((3, 1024)).astype(np.float32)
    r1 = B.Ref(1)
    assert np.array_equal(bits(r1.mul_mat(12, w, 1024, 64, x)), bits(ref.mul_mat(12, w, 1024, 64, x)))
    r1.close()


def test_embedding(oracle, ref):
    from powerserve_amd import synth
    rng = np.random.default_rng(9)
    for t in (0, 2, 8):
        tab = synth.random_blocks(rng, t, 50, 256)
        toks = [0, 49, 7]
        assert np.array_equal(ref.get_embedding(t, tab, 256, 50, toks), oracle.get_embedding(t, tab, 256, toks))


# ---------------------------------------------------------------- token-tree forward
def _load_tensors(path):
    from powerserve_amd import gguf
    rd = gguf.GGUFReader(path)
    return {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}


TREE = np.array([[1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],   # 0 root
                 [1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],   # 1 <- 0
                 [1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0],   # 2 <- 0
                 [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0],   # 3 <- 0
                 [1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0],   # 4 <- 1
                 [1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0],   # 5 <- 1
                 [1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0],   # 6 <- 2
                 [1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0],   # 7 <- 4
                 [1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0],   # 8 <- 7
                 [1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0],   # 9 <- 6
                 [1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0],   # 10 <- 8
                 [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1]],  # 11 <- 3
                dtype=np.uint8)
TREE_DEPTH = np.array([0, 1, 1, 1, 2, 2, 2, 3, 4, 3, 5, 2])


@pytest.mark.parametrize("preset,t", [("tiny-llama", 8), ("tiny-qwen2", 2)])
def test_ref_ops_forward_reproduces_the_real_model_forward(ref, tmp_path, preset, t):
    """The plumbing of oracle/ref_ops_forward.py (the reference's compiled operators sequenced by this repository) is right:
    with the executor's own mask it gives the logits of the real LlamaModel / Qwen2Model::forward bit for bit — prefill
    batches behind each other and single-token steps."""
    from oracle import binding as B
    from oracle.ref_ops_forward import RefOpsModel
    from powerserve_amd import synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, t, n_ctx=96, seed=21)
    cfg = B.make_config(mj["llm_config"])
    path = os.path.join(d, "ggml", "weights.gguf")
    real = ref.model(path, mj["model_arch"], cfg, 2)
    mine = RefOpsModel(ref, cfg, mj["model_arch"], _load_tensors(path))
    toks = np.random.default_rng(3).integers(0, cfg.vocab_size, 40)
    done = 0
    for bs in (7, 12, 1, 1, 5, 1):
        pos = np.arange(done, done + bs)
        want = real.forward(toks[done:done + bs], pos, True)
        got = mine.forward_causal(toks[done:done + bs], pos, True)
        assert np.array_equal(bits(got), bits(want)), (preset, done, bs)
        done += bs
    real.close()


@pytest.mark.parametrize("preset,t", [("tiny-llama", 8), ("tiny-qwen2", 2), ("small-llama-hs128", 12), ("tiny-llama", 1015)])
def test_tree_forward_restatement_vs_reference_operators(oracle, ref, tmp_path, preset, t):
    """pso_model_forward_tree (oracle/ps_oracle.c) against the reference's own operators given the tree mask: a branching
    12-node tree behind a 37-token prefix with two hidden cache slots, RoPE positions = prefix + depth (not the slots) —
    the logits of EVERY node bit for bit, and the K / V rows that were appended."""
    from oracle import binding as B
    from oracle.ref_ops_forward import RefOpsModel
    from powerserve_amd import synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, t, n_ctx=96, seed=33)
    cfg = B.make_config(mj["llm_config"])
    tensors = _load_tensors(os.path.join(d, "ggml", "weights.gguf"))
    om = oracle.model(cfg, mj["model_arch"], tensors, n_threads=4)
    rm = RefOpsModel(ref, cfg, mj["model_arch"], tensors)
    rng = np.random.default_rng(8)
    P = 37
    prefix = rng.integers(0, cfg.vocab_size, P)
    for lo in (0, 20):
        hi = min(P, lo + 20)
        a = om.forward(prefix[lo:hi], np.arange(lo, hi), True)
        b = rm.forward_causal(prefix[lo:hi], np.arange(lo, hi), True)
        assert np.array_equal(bits(a), bits(b))
    kv_vis = np.ones(cfg.seq_len, dtype=np.uint8)
    kv_vis[[5, 33]] = 0
    toks = rng.integers(0, cfg.vocab_size, 12)
    rope = P + TREE_DEPTH
    want = rm.forward_tree(toks, rope, TREE, kv_vis, True, advance=False)
    got = om.forward_tree(toks, rope, TREE, kv_vis, True, advance=False)
    assert np.array_equal(bits(got), bits(want))
    assert om.position == P == rm.position
    for L in range(cfg.n_layers):
        assert np.array_equal(bits(om.k_cache(L)[:P + 12]), bits(rm.k_cache[L][:P + 12]))
        assert np.array_equal(bits(om.v_cache(L)[:, :P + 12]), bits(rm.v_cache[L][:, :P + 12]))
    # the mask matters: the causal reading of the same batch differs for every node but the root
    causal = om.forward_tree(toks, rope, None, None, True, advance=False)
    assert np.array_equal(bits(causal[0]), bits(om.forward_tree(toks, rope, TREE, None, True, advance=False)[0]))
    assert not any(np.array_equal(causal[i], got[i]) for i in range(12))
    om.close()


def test_q4k_corner_case_blocks(oracle, ref):
    """scale 0 / min 63, d = 0, dmin = 0, every field at its maximum, -0, subnormal d (tests/test_gpu_ops.py plants them for the
    GPU kernels): the restatement agrees with the reference on them, too"""
    from test_gpu_ops import q4k_extreme_blocks
    for K, N, bs in [(4096, 48, 1), (2048, 32, 5), (1024, 32, 3)]:
        rng = np.random.default_rng(K + N + bs + 44)
        w = q4k_extreme_blocks(rng, N, K)
        x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
        x[0, 0:512] = np.where(rng.random(512) < 0.5, 1.0, -1.0) * 7.0
        assert np.array_equal(bits(ref.mul_mat(12, w, K, N, x)), bits(oracle.mul_mat(12, w, K, N, x)))


@pytest.mark.parametrize("preset,t", [("odd-llama-hs96", 8), ("odd-llama-hs32", 8), ("odd-qwen2-r3", 2), ("odd-llama-r5", 8), ("odd-llama-r6", 8), ("odd-llama-r8", 2)])
def test_whole_forwards_at_head_sizes_and_gqa_ratios_no_public_config_has(oracle, ref, tmp_path, preset, t):
    """Head sizes 32 / 96 and 1, 3, 5, 6, 8 query heads per kv head (synth.PRESETS "odd-*"): the restatement's generate() against the real
    LlamaModel / Qwen2Model::forward, ids and every step's logits on bits (the shapes tools/gpu_fuzz.py --odd-share draws on the GPU)."""
    from conftest import load_tensors
    from oracle import binding as B
    from powerserve_amd import synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, t, n_ctx=64, seed=3)
    cfg = B.make_config(mj["llm_config"])
    path = os.path.join(d, "ggml/weights.gguf")
    om = oracle.model(cfg, mj["model_arch"], load_tensors(path), n_threads=4)
    rm = ref.model(path, mj["model_arch"], cfg, 2)
    prompt = np.random.default_rng(1).integers(0, cfg.vocab_size, 19)
    ids, lg, *_ = om.generate(prompt, 8, 6, want_logits=True)
    rids, rlg, *_ = rm.generate(prompt, 8, 6, want_logits=True)
    assert np.array_equal(ids, rids)
    assert np.array_equal(bits(lg), bits(rlg))
    om.close()
    rm.close()


@pytest.mark.parametrize("fast", [False, True])
def test_seeded_random_sweep_against_the_real_reference(ref, fast):
    """A fixed stretch of tools/cpu_fuzz_oracle.py: whole-model generate() of the restatement against the real forward at random shapes, chunkings and lengths,
    ids and logits on bits; fast: the stock-flags build against the restatement's contract mode.  One process per draw (the reference leaks a ggml context
    and its spinning pool threads per model)."""
    import importlib.util
    import sys
    from oracle import binding as B
    if fast and not B.have_ref_fast():
        pytest.skip("oracle/_ref/libps_ref_fast.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cpu_fuzz_oracle", os.path.join(root, "tools", "cpu_fuzz_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, fails = mod.run(seconds=240, seed=21 + int(fast), max_draws=10, fast=fast)
    assert n == 10 and not fails, fails


@pytest.mark.parametrize("fs,af", [(0.5, 1.0), (0.25, 1.25), (1.0, 0.8), (0.3, 0.7)])
def test_scaled_rope_against_the_real_reference(oracle, ref, tmp_path, fs, af):
    """rope_freq_scale / rope_attn_factor off 1.0 (src/core/config.cpp:96,98 -> rope_compute_params -> ggml_rope_cache_init, ggml.c:15344-15358): no
    preset, fixture or fuzz draw had moved them before round 6.  The operator in both rotation modes and whole-model generations, restatement against the live reference."""
    from conftest import load_tensors
    from oracle import binding as B
    from powerserve_amd import synth
    rng = np.random.default_rng(int(fs * 100 + af * 10))
    pos = np.array([0, 1, 17, 777, 2047, 4095], dtype=np.int32)
    for mode, hs, base in ((0, 64, 1e4), (2, 64, 1e6), (0, 128, 5e5), (2, 128, 1e4)):
        x = rng.standard_normal((pos.size, 3, hs)).astype(np.float32)
        rp = B.RopeParams(hs, 4096, base, fs, 0.0, af, 32.0, 0.0, mode)
        assert np.array_equal(bits(ref.rope(x, pos, rp)), bits(oracle.rope(x, pos, rp))), (mode, hs)
    for preset, t in (("tiny-llama", 8), ("tiny-qwen2", 2)):
        d = str(tmp_path / preset)
        mj = synth.write_model_dir(d, preset, t, n_ctx=64, seed=5, rope_freq_scale=fs, rope_attn_factor=af)
        cfg = B.make_config(mj["llm_config"])
        path = os.path.join(d, "ggml/weights.gguf")
        om = oracle.model(cfg, mj["model_arch"], load_tensors(path), n_threads=4)
        rm = ref.model(path, mj["model_arch"], cfg, 2)
        prompt = np.random.default_rng(2).integers(0, cfg.vocab_size, 19)
        ids, lg, *_ = om.generate(prompt, 8, 8, want_logits=True)
        rids, rlg, *_ = rm.generate(prompt, 8, 8, want_logits=True)
        om.close()
        rm.close()
        assert np.array_equal(ids, rids)
        assert np.array_equal(bits(lg), bits(rlg))
