"""A GGUF file this repository did not write: tests/golden/ref_written_model/ggml/weights.gguf comes out of the
reference's own gguf_write_to_file (generator oracle/gen_golden_gguf.py, writer harness oracle/ref_gguf.cpp) and looks
like a stock Q4_K_M Llama-3 checkpoint in miniature — Q4_K / Q6_K mix, a `rope_freqs.weight` tensor, tokenizer arrays,
metadata keys of every value type, general.alignment 64.  Both product readers (csrc/host/json_gguf.cpp, gguf.py) must
find every tensor where the writer put it; the bytes must be the seeded synthetic tensors the generator fed in."""
import os

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_written_model")
PATH = os.path.join(FIX, "ggml", "weights.gguf")


def fnv1a64(b: bytes) -> str:
    h = 0xcbf29ce484222325
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def expected_tensors(tmp_path):
    from powerserve_amd import gguf, synth
    mj = synth.load_model_json(FIX)
    assert mj["model_id"] == "tiny-llama-Q4_K_M-refwriter"
    d = str(tmp_path / "mine")
    synth.write_model_dir(d, "tiny-llama", synth.Q4_K_M, n_ctx=mj["llm_config"]["n_ctx"], seed=4321, model_id=mj["model_id"])
    rd = gguf.GGUFReader(os.path.join(d, "ggml", "weights.gguf"))
    return {t.name: (t.type, tuple(t.ne), np.array(rd.data(t.name)).tobytes()) for t in rd.tensors.values()}


def test_cpp_reader_on_reference_written_gguf(tmp_path):
    from powerserve_amd import host
    want = expected_tensors(tmp_path)
    tensors, strings, numbers = host.gguf_summary(PATH)
    assert strings["general.architecture"] == "llama" and strings["tokenizer.ggml.model"] == "llama"
    assert numbers["general.alignment"] == 64 and numbers["general.file_type"] == 15
    # one key of every scalar type was stepped over with the right width
    assert (numbers["test.u8"], numbers["test.i8"], numbers["test.u16"], numbers["test.i16"]) == (200, -100, 60000, -30000)
    assert (numbers["test.i32"], numbers["test.f32"], numbers["test.u64"], numbers["test.i64"]) == (-2000000000, 0.15625, 2.0**40, -2.0**40)
    assert numbers["test.f64"] == 1.0 / 3.0 and numbers["test.bool"] == 1
    names = [t[0] for t in tensors]
    assert names[1] == "rope_freqs.weight" and set(names) - {"rope_freqs.weight"} == set(want)
    assert {t[1] for t in tensors} == {0, 12, 14}  # F32, Q4_K, Q6_K
    for name, typ, nbytes, h, ne in tensors:
        if name == "rope_freqs.weight":
            assert (typ, ne, nbytes) == (0, (32,), 128)
            continue
        wt, wne, wb = want[name]
        assert (typ, ne, nbytes) == (wt, wne, len(wb)), name
        assert h == fnv1a64(wb), name


def test_python_reader_on_reference_written_gguf(tmp_path):
    from powerserve_amd import gguf
    want = expected_tensors(tmp_path)
    rd = gguf.GGUFReader(PATH)
    assert rd.kv["tokenizer.ggml.tokens"] == ["<s>", "</s>", "hello", "", "世界"] and rd.kv["tokenizer.ggml.token_type"] == [3, 3, 1, 1, 1]
    assert rd.data_off % 64 == 0
    for name, (wt, wne, wb) in want.items():
        ti = rd.tensors[name]
        assert (ti.type, tuple(ti.ne)) == (wt, wne) and ti.offset % 64 == 0
        assert np.array(rd.data(name)).tobytes() == wb, name
    f = rd.data("rope_freqs.weight")
    assert f.dtype == np.float32 and f.shape == (32,) and f[0] == 1.0 and f[-1] == 8.0


@pytest.mark.gpu
def test_reference_written_model_runs_and_ignores_rope_freqs(ctx, oracle, tmp_path):
    """Loaded through json_gguf.cpp (HostModel) and through gguf.py (hip.Model): both give, bit for bit, the logits of
    the same tensors in a file of this repository's writer, which are the CPU oracle's.  `rope_freqs.weight` changes
    nothing: the reference's rope() is called without frequency factors."""
    from powerserve_amd import hip, host, synth
    mj = synth.load_model_json(FIX)
    mine = str(tmp_path / "mine")
    synth.write_model_dir(mine, "tiny-llama", synth.Q4_K_M, n_ctx=mj["llm_config"]["n_ctx"], seed=4321, model_id=mj["model_id"])
    prompt = np.random.default_rng(3).integers(0, mj["llm_config"]["vocab_size"], 11)
    pos = np.arange(prompt.size)
    a = host.HostModel(FIX, max_batch=16)
    la = a.forward(prompt, pos, lm_head=True)
    ids_a = a.generate(prompt, 8, 16)
    a.close()
    b = hip.Model(ctx, FIX, max_batch=16)
    lb, _ = b.forward(prompt, pos, lm_head=True)
    b.close()
    c = hip.Model(ctx, mine, max_batch=16)
    lc, _ = c.forward(prompt, pos, lm_head=True)
    ids_c = c.generate(prompt, 8, 16)
    c.close()
    assert np.array_equal(la.view(np.uint32), lc.view(np.uint32)) and np.array_equal(lb.view(np.uint32), lc.view(np.uint32))
    assert np.array_equal(ids_a, ids_c)
    from oracle import binding as B
    from test_gpu_host import load_tensors
    om = oracle.model(B.make_config(mj["llm_config"]), mj["model_arch"], load_tensors(os.path.join(mine, "ggml", "weights.gguf")), n_threads=4)
    lo = om.forward(prompt, pos, True)
    om.close()
    assert np.array_equal(lo.view(np.uint32), lc.view(np.uint32))
