"""GPU: the C++ host facade (Graph -> Executor -> HIPBackend, mirror of the reference's host side).  Every forward builds
the reference's graph with the NormAttention / FFN builders and goes through Executor::run; HIPBackend::plan recognises
the canonical op sequence and lowers it to the fused launch plan.  The lowered path and the op-by-op path (every
reference op through its own ps_hip_* entry point; plan() lowering switched off) must give the SAME bits, and both must
equal the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def load_tensors(path):
    from powerserve_amd import gguf
    rd = gguf.GGUFReader(path)
    return {n: (ti.type, np.array(rd.data(n)), ti.ne[0], (list(ti.ne) + [1])[1]) for n, ti in rd.tensors.items()}


@pytest.mark.parametrize("preset,wt", [("tiny-llama", 2), ("tiny-llama", 12), ("tiny-qwen2", 8), ("small-llama-hs128", 12)])
def test_graph_path_equals_fused_equals_oracle(oracle, tmp_path, preset, wt):
    from oracle import binding as B
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=96, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    hm = host.HostModel(d, 0, max_batch=16)
    prompt = np.random.default_rng(1).integers(0, cfg.vocab_size, 14)
    want_ids, want_logits, *_ = om.generate(prompt, 8, 10, want_logits=True)
    # Graph -> Executor with plan() lowering (the default), through the C++ Model::generate
    assert np.array_equal(hm.generate(prompt, 8, 10), want_ids)
    n_plans, n_low = hm.plan_stats()
    assert n_plans > 0 and n_low == n_plans          # every graph the builders emitted was recognised
    # op-by-op graph path: same ids, and bit-identical logits on a batched forward
    hm.set_fused(False)
    assert np.array_equal(hm.generate(prompt, 8, 10), want_ids)
    assert hm.plan_stats()[1] == n_low               # nothing lowered with the switch off
    hm.reset(); om.reset()
    lg_graph = hm.forward(prompt[:9], np.arange(9), True)
    lg_oracle = om.forward(prompt[:9], np.arange(9), True)
    assert np.array_equal(lg_graph.view(np.uint32), lg_oracle.view(np.uint32))
    assert hm.position == 9
    hm.set_fused(True)
    hm.reset()
    lg_fused = hm.forward(prompt[:9], np.arange(9), True)  # lowered: one graph, fused launches
    assert hm.plan_stats()[1] == n_low + 1
    assert np.array_equal(lg_fused.view(np.uint32), lg_graph.view(np.uint32))
    assert hm.position == 9
    # single-token steps through Graph -> Executor -> plan() (5 launches per layer): teacher-forced logits == oracle
    om.reset(); om.forward(prompt[:9], np.arange(9), False)
    for s in range(3):
        a = hm.forward([int(prompt[9 + s])], [9 + s], True)
        b = om.forward([int(prompt[9 + s])], [9 + s], True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), s
    hm.close(); om.close()


@pytest.mark.parametrize("preset,wt", [("tiny-llama", 12), ("tiny-qwen2", 8)])
def test_decode_hands_back_the_device_argmax_and_the_prefill_loop_is_lowered(oracle, tmp_path, preset, wt):
    """Model::decode (greedy) over the op API: a lowered graph returns the ids of the device arg-max kernel (4 bytes per token, the logits stay on the
    GPU -- SURVEY a21, src/model/model.hpp:170-183), an op-by-op graph arg-maxes the copied logits on the host like the reference: the same ids, the
    oracle's.  Model::prefill (ModelTokenIterator's loop) plans its first chunk's graph and, lowered, hands the whole loop to ps_hip_model_prefill:
    the cache rows are the chunk-by-chunk ones."""
    from oracle import binding as B
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, preset, wt, n_ctx=96, seed=6)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    prompt = np.random.default_rng(2).integers(0, cfg.vocab_size, 30)
    want_ids, *_ = om.generate(prompt, 8, 12)
    hm = host.HostModel(d, 0, max_batch=16)
    for fused in (True, False):
        hm.set_fused(fused)
        hm.reset()
        p0 = hm.plan_stats()
        hm.prefill(prompt[:-1], 8)  # 29 tokens: 8 + 8 + 8 + 5
        assert hm.position == 29
        if fused:
            assert hm.plan_stats()[1] == p0[1] + 1  # ONE graph planned and lowered for the whole loop
        cur, got = int(prompt[-1]), []
        for s in range(12):
            cur = int(hm.decode([cur], [29 + s])[0])
            got.append(cur)
        assert np.array_equal(got, want_ids), fused
        if fused:  # eleven of the twelve single-token graphs were never built: the (1, lm_head) shape had been lowered once (the plan cache)
            assert hm.plan_cache_hits() == 11, hm.plan_cache_hits()
            hm.set_plan_cache(False)  # ... and without the cache every step builds and plans its graph: the same ids
            hm.kv("rollback_tokens", 12)
            p1 = hm.plan_stats()
            cur, again = int(prompt[-1]), []
            for s in range(12):
                cur = int(hm.decode([cur], [29 + s])[0])
                again.append(cur)
            assert np.array_equal(again, want_ids) and hm.plan_stats()[0] == p1[0] + 12 and hm.plan_cache_hits() == 11
            hm.set_plan_cache(True)
        for L in range(2):
            om_k, om_v = om.k_cache(L)[:29], om.v_cache(L)[:, :29]
            for slot in (0, 7, 8, 28):
                k, v = hm.kv_read(L, slot, cfg.kv_dim)
                assert np.array_equal(k.view(np.uint32), om_k[slot].view(np.uint32)) and np.array_equal(v.view(np.uint32), om_v[:, slot].view(np.uint32)), (fused, L, slot)
    # a batch through decode(): one id per column
    hm.set_fused(True)
    hm.reset(); om.reset()
    ids = hm.decode(prompt[:9], np.arange(9))
    lg = om.forward(prompt[:9], np.arange(9), True)
    assert np.array_equal(ids, np.argmax(lg, axis=1))
    hm.close(); om.close()


def test_plan_does_not_lower_what_the_fused_forward_cannot_do(oracle, tmp_path):
    """HIPBackend::plan lowers a graph only when the fused forward does exactly what the graph says (advisor, round 2): a
    forward whose first position is NOT the cache position (an earlier position run again) keeps the canonical op order
    but appends elsewhere than HIPKV::advance assumes — it must run op by op, write where it is told, and still equal the
    oracle; a batch wider than the model's buffers must run op by op instead of aborting."""
    from oracle import binding as B
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 12, n_ctx=96, seed=6)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    hm = host.HostModel(d, 0, max_batch=8)
    toks = np.random.default_rng(2).integers(0, cfg.vocab_size, 24)
    a = hm.forward(toks[:8], np.arange(8), True)
    b = om.forward(toks[:8], np.arange(8), True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    low0 = hm.plan_stats()[1]
    assert low0 == 1
    # positions 2..4 once more while the cache stands at 8: same ops, different append slot
    a = hm.forward(toks[8:11], np.arange(2, 5), True)
    b = om.forward(toks[8:11], np.arange(2, 5), True)
    assert hm.plan_stats()[1] == low0, "a graph for another cache position was lowered"
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # wider than max_batch = 8
    hm.reset(); om.reset()
    a = hm.forward(toks[:12], np.arange(12), True)
    b = om.forward(toks[:12], np.arange(12), True)
    assert hm.plan_stats()[1] == low0, "a batch wider than the model's buffers was lowered"
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    hm.close(); om.close()


def test_host_errors_surface(tmp_path):
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "tiny-llama", 8, n_ctx=16)
    hm = host.HostModel(d, 0, max_batch=8)
    with pytest.raises(host.HostError):
        hm.forward([1, 2, 3], [14, 15, 16], lm_head=False)   # KV full
    hm.set_fused(False)
    with pytest.raises(host.HostError):
        hm.forward([1, 2, 3], [14, 15, 16], lm_head=False)
    with pytest.raises(host.HostError):
        host.HostModel(str(tmp_path / "missing"), 0)
    hm.close()


@pytest.mark.parametrize("kw", [dict(seed=7), dict(seed=3, temperature=1.3, top_k=50, penalty_repeat=1.2, penalty_last_n=8, penalize_nl=True)])
def test_sampled_generation_matches_reference_pipeline(oracle, ref, tmp_path, kw):
    """Model::generate with the sampler chain on the GPU path == CPU oracle logits fed through the reference's own sampler
    classes (oracle/_ref): the logits are bit-identical, so the sampled token streams must be too."""
    import ctypes as C
    from oracle import binding as B
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "small-llama-hs128", 12, n_ctx=96, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=8)
    hm = host.HostModel(d, 0, max_batch=16)
    prompt = np.random.default_rng(2).integers(0, cfg.vocab_size, 11)
    steps = 24
    scfg = host.SamplerCfg.make(cfg.vocab_size, **kw)
    got = host.generate_sampled(hm, prompt, 8, steps, scfg)
    assert np.array_equal(got, host.generate_sampled(hm, prompt, 8, steps, scfg))  # same seed, same stream
    # reference pipeline: oracle forward + the reference's SamplerChain
    rc = B.SamplerCfg.from_buffer_copy(bytes(scfg))
    rh = ref.L.ref_sampler_create(C.byref(rc))
    om.forward(prompt[:8], np.arange(8), False)
    om.forward(prompt[8:10], np.arange(8, 10), False)
    cur, want = int(prompt[-1]), []
    for s in range(steps):
        lg = om.forward([cur], [om.position], True)
        cur = ref.L.ref_sampler_sample(rh, lg.ctypes.data, cfg.vocab_size)
        want.append(cur)
    ref.L.ref_sampler_free(rh)
    assert list(got) == want
    # top_k = 1 through the chain is the greedy path
    g1 = host.generate_sampled(hm, prompt, 8, steps, host.SamplerCfg.make(cfg.vocab_size, top_k=1))
    assert np.array_equal(g1, hm.generate(prompt, 8, steps))
    hm.close(); om.close()


def test_workspace_folder(tmp_path):
    """workspace.json -> hparams + main / draft model directories (src/core/config.cpp:121-152), opened on the HIP backend:
    greedy hparams (top_k 1) reproduce Model::generate; with a draft model configured the output is still the target's."""
    import json
    from powerserve_amd import host, synth
    wf = tmp_path / "work"
    synth.write_model_dir(str(wf / "main"), "small-llama-hs128", 12, n_ctx=128, seed=3)
    synth.write_model_dir(str(wf / "draft"), "small-llama-draft", 2, n_ctx=128, seed=4)
    (wf / "hparams.json").write_text(json.dumps({"batch_size": 16, "sampler": {"top_k": 1, "seed": 5}}))
    (wf / "workspace.json").write_text(json.dumps({"hparams_config": "hparams.json", "model_main": "main"}))
    prompt = np.random.default_rng(0).integers(0, 1024, 19)
    ws = host.Workspace(str(wf))
    assert ws.batch_size == 16 and ws.draft is None
    want = ws.main.generate(prompt, 16, 12)
    assert np.array_equal(ws.generate(prompt, 12), want)
    ws.close()
    (wf / "workspace.json").write_text(json.dumps({"hparams_config": "hparams.json", "model_main": "main", "model_draft": "draft"}))
    ws = host.Workspace(str(wf))
    assert np.array_equal(ws.generate(prompt, 12), want)
    ws.close()
