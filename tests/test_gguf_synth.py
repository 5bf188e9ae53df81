"""CPU: GGUF writer/reader round trip and the synthetic quantized-domain weight generator."""
import os

import numpy as np


def test_gguf_roundtrip(tmp_path):
    from powerserve_amd import gguf
    p = str(tmp_path / "t.gguf")
    w = gguf.GGUFWriter(p)
    w.add_kv("general.architecture", "llama")
    w.add_kv("x.int", 7)
    w.add_kv("x.float", 0.5)
    w.add_tensor("a", gguf.F32, (5,))
    w.add_tensor("b", gguf.Q4_0, (64, 3))
    a = np.arange(5, dtype=np.float32)
    b = np.arange(3 * 2 * 18, dtype=np.uint8)
    w.write(lambda ti: a if ti.name == "a" else b)
    r = gguf.GGUFReader(p)
    assert r.kv["general.architecture"] == "llama" and r.kv["x.int"] == 7
    assert np.array_equal(r.data("a"), a) and np.array_equal(r.data("b"), b)
    assert r.data_off % 32 == 0 and r.tensors["b"].offset % 32 == 0


def test_synth_blocks_are_sane(oracle):
    from powerserve_amd import synth
    rng = np.random.default_rng(0)
    for t in (2, 8, 12, 14):
        blocks = synth.random_blocks(rng, t, 8, 2048, std=0.02)
        w = np.stack([oracle.dequantize(t, blocks.reshape(8, -1)[i], 2048) for i in range(8)])
        assert np.isfinite(w).all()
        assert 0.012 < w.std() < 0.035 and abs(w.mean()) < 0.004, (t, w.std(), w.mean())


def test_model_dir_layout(tmp_path):
    from powerserve_amd import gguf, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-qwen2", 8, n_ctx=64)
    assert os.path.exists(os.path.join(d, "model.json")) and os.path.exists(os.path.join(d, "ggml", "weights.gguf"))
    assert mj["llm_config"]["kv_dim"] == 128 and mj["model_arch"] == "qwen2"
    r = gguf.GGUFReader(os.path.join(d, "ggml", "weights.gguf"))
    assert "blk.1.attn_q.bias" in r.tensors and "output.weight" not in r.tensors  # tied lm_head
    assert r.tensors["blk.0.ffn_down.weight"].ne == (512, 256)


def test_q4_k_m_recipe(tmp_path):
    """synth.Q4_K_M follows llama.cpp's per-tensor recipe: Q4_K everywhere, Q6_K for attn_v / ffn_down in the "more bits"
    layers (first and last eighth, every third in between) and for output.weight; the file reads back with those types."""
    from powerserve_amd import gguf, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "small-llama", synth.Q4_K_M, n_ctx=32, seed=1)  # 3 layers, tied embeddings
    rd = gguf.GGUFReader(d + "/ggml/weights.gguf")
    t = {n: ti.type for n, ti in rd.tensors.items()}
    more = [synth._more_bits(i, 3) for i in range(3)]
    assert any(more) and not all(more)
    for i in range(3):
        want = gguf.Q6_K if more[i] else gguf.Q4_K
        assert t[f"blk.{i}.attn_v.weight"] == want and t[f"blk.{i}.ffn_down.weight"] == want
        for n in ("attn_q", "attn_k", "attn_output", "ffn_gate", "ffn_up"):
            assert t[f"blk.{i}.{n}.weight"] == gguf.Q4_K
    assert t["token_embd.weight"] == gguf.Q6_K and "output.weight" not in t  # tied: the table takes the output's type
    d2 = str(tmp_path / "m2")
    synth.write_model_dir(d2, "small-llama-hs128", synth.Q4_K_M, n_ctx=32, seed=1)  # untied
    t2 = {n: ti.type for n, ti in gguf.GGUFReader(d2 + "/ggml/weights.gguf").tensors.items()}
    assert t2["output.weight"] == gguf.Q6_K and t2["token_embd.weight"] == gguf.Q4_K
