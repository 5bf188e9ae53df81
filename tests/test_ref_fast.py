"""CPU, dev container (needs oracle/_ref/libps_ref.so AND libps_ref_fast.so): the two legitimate builds of the reference against each other.

`libps_ref.so` is compiled with -ffp-contract=off (what the oracle, the golden vectors and the HIP library's "bit-exact" refer to); `libps_ref_fast.so` is
the same sources under GCC's default -ffp-contract=fast, which is what the reference's own CMake produces on an FMA machine (it sets no contraction flag:
CMakeLists.txt:24-33, libs/ggml/src/CMakeLists.txt:1173).  These tests pin (1) that on the hot path exactly three places differ between the two -- the RoPE
rotation, the n % 32 leftovers of ggml_vec_dot_f32, Q5_K's summs -- and nothing else does, and (2) that the oracle's contract mode (pso_set_contract(1)) IS
the contracted build, bit for bit, op by op and through whole forwards (the fixtures tests/golden/e2e_builds_*.npz carry the same statement to boxes
without the reference)."""
import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ref_fast():
    from oracle import binding as B
    if not (B.have_ref() and B.have_ref_fast()):
        pytest.skip("oracle/_ref/libps_ref_fast.so not built (make -C oracle ref_fast; needs /root/reference)")
    return B.Ref(2, so=B.REF_FAST_SO)


@pytest.fixture()
def contracted(oracle):
    oracle.L.pso_set_contract(1)
    yield oracle
    oracle.L.pso_set_contract(0)


def f32_dot(r, w, x):
    from oracle import binding as B
    N, K = w.shape
    y = np.empty((x.shape[0], N), np.float32)
    r.mul_mat_t(B.ref_tensor(y, B.F32, [N, x.shape[0]]), B.ref_tensor(w, B.F32, [K, N]), B.ref_tensor(x, B.F32, [K, x.shape[0]]))
    return y


def test_exactly_three_sites_differ_between_the_builds(ref, ref_fast):
    from oracle import binding as B
    from powerserve_amd import synth
    rng = np.random.default_rng(0)
    differs, same = {}, {}
    q = rng.standard_normal((64, 4, 64)).astype(np.float32)
    pos = rng.integers(0, 4096, 64).astype(np.int32)
    for mode in (0, 2):
        rp = B.RopeParams(64, 4096, 5e5, 1.0, 0.0, 1.0, 32.0, 0.0, mode)
        differs[f"rope mode {mode}"] = float((bits(ref.rope(q, pos, rp)) != bits(ref_fast.rope(q, pos, rp))).mean())
    w, x = rng.standard_normal((64, 63)).astype(np.float32), rng.standard_normal((3, 63)).astype(np.float32)
    differs["f32 dot, 31 leftovers"] = float((bits(f32_dot(ref, w, x)) != bits(f32_dot(ref_fast, w, x))).mean())
    for K in (64, 128, 40, 100):  # no leftovers / 8 leftovers / 4 leftovers: the vectorised part of the leftover loop is not fused
        w, x = rng.standard_normal((64, K)).astype(np.float32), rng.standard_normal((3, K)).astype(np.float32)
        same[f"f32 dot K {K}"] = float((bits(f32_dot(ref, w, x)) != bits(f32_dot(ref_fast, w, x))).mean())
    for t in (B.Q4_0, B.Q8_0, B.Q4_K, B.Q5_K, B.Q6_K):
        wq = synth.random_blocks(rng, t, 64, 1024)
        xx = rng.standard_normal((5, 1024)).astype(np.float32)
        (differs if t == B.Q5_K else same)["mul_mat " + B.TYPE_NAMES[t]] = float((bits(ref.mul_mat(t, wq, 1024, 64, xx)) != bits(ref_fast.mul_mat(t, wq, 1024, 64, xx))).mean())
    x = (rng.standard_normal((8, 896)) * 2).astype(np.float32)
    wn = (1 + 0.1 * rng.standard_normal(896)).astype(np.float32)
    same["rms_norm"] = float((bits(ref.rms_norm(x, wn, 1e-6)) != bits(ref_fast.rms_norm(x, wn, 1e-6))).mean())
    for n_kv in (1, 33, 300, 2301):
        s = (rng.standard_normal((4, 2, n_kv)) * 4).astype(np.float32)
        mask = np.where(np.arange(n_kv)[None, :] <= np.array([max(0, n_kv - 2), n_kv - 1])[:, None], 0.0, -np.inf).astype(np.float32)
        same[f"softmax {n_kv}"] = float((bits(ref.softmax_ext(s, mask, 0.125)) != bits(ref_fast.softmax_ext(s, mask, 0.125))).mean())
    g, u = (rng.standard_normal(7777) * 4).astype(np.float32), rng.standard_normal(7777).astype(np.float32)
    same["silu_hadamard"] = float((bits(ref.silu_hadamard(g, u)) != bits(ref_fast.silu_hadamard(g, u))).mean())
    for K in (256, 896, 4096):
        xx = (rng.standard_normal(K) * 3).astype(np.float32)
        same[f"q8_0 {K}"] = float((ref.from_float(B.Q8_0, xx) != ref_fast.from_float(B.Q8_0, xx)).mean())
        if K % 256 == 0:
            same[f"q8_K {K}"] = float((ref.from_float(B.Q8_K, xx) != ref_fast.from_float(B.Q8_K, xx)).mean())
    print("differ:", differs)
    assert all(v > 0.02 for v in differs.values()), differs
    assert all(v == 0.0 for v in same.values()), same


def test_oracle_contract_mode_is_the_contracted_build(contracted, ref_fast):
    from oracle import binding as B
    from powerserve_amd import synth
    import ctypes as C
    o = contracted
    rng = np.random.default_rng(1)
    for mode, hs in ((0, 64), (2, 64), (0, 128)):
        q = rng.standard_normal((64, 4, hs)).astype(np.float32)
        pos = rng.integers(0, 4096, 64).astype(np.int32)
        rp = B.RopeParams(hs, 4096, 5e5, 1.0, 0.0, 1.0, 32.0, 0.0, mode)
        assert np.array_equal(bits(o.rope(q, pos, rp)), bits(ref_fast.rope(q, pos, rp)))
    o.L.pso_vec_dot_f32.restype = C.c_float
    o.L.pso_vec_dot_f32.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
    for K in list(range(1, 70)) + [95, 97, 127, 2049, 2051, 2079]:
        w, x = rng.standard_normal((32, K)).astype(np.float32), rng.standard_normal((2, K)).astype(np.float32)
        got = np.array([[o.L.pso_vec_dot_f32(K, w[n].ctypes.data, x[b].ctypes.data) for n in range(32)] for b in range(2)], dtype=np.float32)
        assert np.array_equal(bits(got), bits(f32_dot(ref_fast, w, x))), K
    for K, N, bs in ((256, 48, 3), (1024, 64, 5), (2048, 32, 2)):
        wq = synth.random_blocks(rng, B.Q5_K, N, K)
        xx = rng.standard_normal((bs, K)).astype(np.float32)
        assert np.array_equal(bits(o.mul_mat(B.Q5_K, wq, K, N, xx)), bits(ref_fast.mul_mat(B.Q5_K, wq, K, N, xx)))


@pytest.mark.parametrize("preset,t", [("tiny-llama", 2), ("tiny-qwen2", 8)])
def test_whole_forwards_of_both_builds_and_both_oracle_modes(oracle, ref, ref_fast, tmp_path, preset, t):
    """the real LlamaModel / Qwen2Model::forward of each build == the oracle in the matching mode, every logit of 41 + 42 tokens (another seed than the fixtures)"""
    import os
    from conftest import load_tensors
    from oracle import binding as B
    from powerserve_amd import synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, t, n_ctx=128, seed=77)
    path = os.path.join(d, "ggml", "weights.gguf")
    cfg = B.make_config(mj["llm_config"])
    prompt = np.random.default_rng(77).integers(0, cfg.vocab_size, 41).astype(np.int32)
    try:
        for mode, r in ((0, ref), (1, ref_fast)):
            m = r.model(path, mj["model_arch"], cfg, 2)
            ids, logits, *_ = m.generate(prompt, 8, 42, want_logits=True)
            m.close()
            oracle.L.pso_set_contract(mode)
            om = oracle.model(cfg, mj["model_arch"], load_tensors(path), n_threads=4)
            oids, ologits, *_ = om.generate(prompt, 8, 42, want_logits=True)
            om.close()
            assert np.array_equal(oids, ids) and np.array_equal(bits(ologits), bits(logits)), mode
    finally:
        oracle.L.pso_set_contract(0)
