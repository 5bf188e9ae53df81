"""Round 5: the retry paths behind a time-out of the one-launch attention (every entry point, forced through
ps_hip_debug_set(5, n)), and the boundary members no model graph uses -- ps_hip_soft_max / Graph::softmax, get_n_tasks,
add_cache, and the KVCacheInterface members copy / save_tokens / unmask_tokens / append_tokens
(reference: libs/ggml/include/ggml.h:781, src/graph/graph.cpp:118, src/backend/ggml/ggml.hpp:227,233,
src/core/kv_cache.hpp:120-162)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_tensors

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture()
def pair(ctx, oracle, tmp_path):
    """tiny Q4_K llama: (hip model on the ONE-launch attention, oracle model, config, prompt)"""
    from oracle import binding as B
    from powerserve_amd import hip, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 12, n_ctx=128, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    gm = hip.Model(ctx, d, max_batch=16)
    gm.set_mode(64)  # re-arm: whatever an earlier test left behind, this model starts on the one-launch form
    prompt = np.random.default_rng(3).integers(0, cfg.vocab_size, 22)
    yield gm, om, cfg, prompt, d
    ctx.L.ps_hip_debug_set(5, 0)
    gm.close()
    om.close()


def force(ctx, n=1):
    assert ctx.L.ps_hip_debug_set(5, n) == 0


def prefill_both(gm, om, prompt, n):
    gm.reset()
    om.reset()
    gm.forward(prompt[:n], np.arange(n), lm_head=False)
    om.forward(prompt[:n], np.arange(n), False)


def test_timeout_retry_forward(ctx, pair):
    """ps_hip_model_forward, one token: the forced time-out is answered by ONE re-run on the two launches -- same logits, the position moved once,
    and the switch is sticky (a later set_mode(0) keeps bit 4 until bit 6 re-arms)."""
    gm, om, cfg, prompt, _ = pair
    prefill_both(gm, om, prompt, 9)
    force(ctx)
    lg, am = gm.forward([int(prompt[9])], [9], lm_head=True)
    want = om.forward([int(prompt[9])], [9], True)
    assert np.array_equal(bits(lg[0]), bits(want[0]))
    assert gm.position == 10
    # sticky: set_mode(0) must not bring the one-launch form back ...
    gm.set_mode(0)
    force(ctx)  # ... so this armed time-out is NOT consumed by the next forward (note_single_token arms nothing under bit 4)
    lg2, _ = gm.forward([int(prompt[10])], [10], lm_head=True)
    assert np.array_equal(bits(lg2[0]), bits(om.forward([int(prompt[10])], [10], True)[0]))
    gm.set_mode(64)  # re-armed: now the pending forced time-out fires (and is recovered from) on the next single token
    lg3, _ = gm.forward([int(prompt[11])], [11], lm_head=True)
    assert np.array_equal(bits(lg3[0]), bits(om.forward([int(prompt[11])], [11], True)[0]))
    assert gm.position == 12


def test_timeout_retry_decode_greedy_and_graph(ctx, pair):
    """decode_greedy (captured step): a time-out anywhere in the run re-runs the whole run on the two launches; ids equal the oracle's."""
    gm, om, cfg, prompt, _ = pair
    want, *_ = om.generate(prompt, 8, 12)
    gm.reset()
    done = 0
    while done < prompt.size - 1:
        bs = min(8, prompt.size - 1 - done)
        gm.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
        done += bs
    force(ctx)
    got = gm.decode_greedy(int(prompt[-1]), 12)
    assert np.array_equal(got, want), (got, want)
    assert gm.position == prompt.size - 1 + 12


def test_timeout_retry_prefill_tail_and_tree(ctx, pair):
    """ps_hip_model_prefill with a one-token tail chunk, and ps_hip_model_forward_tree with n = 1 (no tree): both single-token forwards."""
    gm, om, cfg, prompt, _ = pair
    gm.reset()
    om.reset()
    force(ctx)
    gm.prefill(prompt[:17], 16)  # 16 + a one-token tail
    om.forward(prompt[:16], np.arange(16), False)
    om.forward(prompt[16:17], [16], False)
    assert gm.position == 17
    n = 17
    assert np.array_equal(bits(gm.k_cache(1)[:n]), bits(om.k_cache(1)[:n]))
    assert np.array_equal(bits(gm.v_cache(1)[:, :n]), bits(om.v_cache(1)[:, :n]))
    gm.set_mode(64)
    force(ctx)
    lg, am = gm.forward_tree([int(prompt[17])], [17], None, lm_head=True, want_logits=True, advance=True)
    want = om.forward([int(prompt[17])], [17], True)
    assert np.array_equal(bits(lg[0]), bits(want[0]))
    assert gm.position == 18


def test_timeout_lowered_forward_is_reported_not_inherited(ctx, pair):
    """The lowered (enqueue-only) forward: sync_check / kv_advance report PS_HIP_ATTN_TIMEOUT, nothing is advanced, the same forward again is valid.
    And an UNCONSUMED lowered forward's time-out is handed back by the next entry point instead of being pinned on that forward."""
    from powerserve_amd import hip
    gm, om, cfg, prompt, _ = pair
    prefill_both(gm, om, prompt, 9)
    want = om.forward([int(prompt[9])], [9], True)
    # (a) kv_advance looks at the flag
    force(ctx)
    gm.forward_lowered([int(prompt[9])], [9], lm_head=True)
    rc = ctx.L.ps_hip_model_kv_advance(gm.h, 1)
    assert rc == hip.ATTN_TIMEOUT, rc
    assert gm.position == 9
    assert b"timed out" in ctx.L.ps_hip_last_error(ctx.h)
    gm.forward_lowered([int(prompt[9])], [9], lm_head=True)  # once more: the model is on the two launches now
    assert gm.sync_check() == 0
    assert np.array_equal(bits(gm.logits(1)[0]), bits(want[0]))
    gm.kv_advance(1)
    assert gm.position == 10
    # (b) an unconsumed lowered forward followed by an unrelated forward: the unrelated one refuses with the time-out code, runs nothing
    gm.set_mode(64)
    force(ctx)
    gm.forward_lowered([int(prompt[10])], [10], lm_head=True)
    tok = np.asarray(prompt[10:13], dtype=np.int32)
    pos = np.arange(10, 13, dtype=np.int32)
    am = np.empty(3, dtype=np.int32)
    rc = ctx.L.ps_hip_model_forward(gm.h, tok.ctypes.data_as(C.c_void_p), 3, pos.ctypes.data_as(C.c_void_p), None, 0, am.ctypes.data_as(C.c_void_p))
    assert rc == hip.ATTN_TIMEOUT, rc
    assert gm.position == 10
    lg, _ = gm.forward([int(prompt[10])], [10], lm_head=True)  # the caller re-runs what it still holds the inputs of
    assert np.array_equal(bits(lg[0]), bits(om.forward([int(prompt[10])], [10], True)[0]))


def test_timeout_retry_host_graph_path(ctx, oracle, tmp_path):
    """Model::forward over the reference's op API (Graph -> Executor::run -> HIPBackend::plan lowers it): the façade's loop runs the graph once
    more behind a time-out, and only then advances the cache."""
    from oracle import binding as B
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 12, n_ctx=128, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    hm = host.HostModel(d, max_batch=16)
    prompt = np.random.default_rng(3).integers(0, cfg.vocab_size, 12)
    hm.forward(prompt[:9], np.arange(9), lm_head=False)
    om.forward(prompt[:9], np.arange(9), False)
    force(ctx)
    lg = hm.forward([int(prompt[9])], [9], lm_head=True)
    assert ctx.L.ps_hip_debug_set(5, 0) == 0
    assert np.array_equal(bits(lg[0]), bits(om.forward([int(prompt[9])], [9], True)[0]))
    assert hm.position == 10
    hm.close()
    om.close()


def test_timeout_retry_on_a_plan_cache_hit(ctx, oracle, tmp_path):
    """The façade's OTHER branch (round-5 advice): once the (1 token, lm_head) shape has been lowered, Model::forward / Model::decode run the cached launch plan without
    a graph; a time-out there is answered by the same one re-run (both branches share Model::forward_graph's run_checked / finish), the cache advances once,
    and the device arg-max of the greedy caller is the re-run's."""
    from oracle import binding as B
    from powerserve_amd import host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 12, n_ctx=128, seed=5)
    cfg = B.make_config(mj["llm_config"])
    om = oracle.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=4)
    hm = host.HostModel(d, max_batch=16)
    prompt = np.random.default_rng(3).integers(0, cfg.vocab_size, 14)
    hm.forward(prompt[:9], np.arange(9), lm_head=False)
    om.forward(prompt[:9], np.arange(9), False)
    lg = hm.forward([int(prompt[9])], [9], lm_head=True)  # builds, plans and lowers the (1, lm_head) graph: the shape is cached from here on
    assert np.array_equal(bits(lg[0]), bits(om.forward([int(prompt[9])], [9], True)[0]))
    hits = hm.plan_cache_hits()
    force(ctx)
    lg = hm.forward([int(prompt[10])], [10], lm_head=True)  # cache hit + forced time-out
    assert ctx.L.ps_hip_debug_set(5, 0) == 0
    assert hm.plan_cache_hits() == hits + 1 and hm.position == 11
    want = om.forward([int(prompt[10])], [10], True)[0]
    assert np.array_equal(bits(lg[0]), bits(want))
    ids = hm.decode([int(prompt[11])], [11])  # (the model is on the two launches now; the greedy caller's 4-byte path)
    assert int(ids[0]) == int(np.argmax(om.forward([int(prompt[11])], [11], True)[0])) and hm.position == 12
    hm.close()
    om.close()


# ---------------------------------------------------------------------------------------------- boundary leftovers
@pytest.mark.parametrize("n", [1, 33, 300])
def test_soft_max_entry_and_graph_softmax(ctx, oracle, tmp_path, n):
    """powerserve_compute_forward_soft_max (ggml.h:781): scale 1, no mask -- through the C entry and through Graph::softmax + Executor."""
    from powerserve_amd import hip, host, synth
    s = (np.random.default_rng(n).standard_normal((6, n)) * 4).astype(np.float32)
    want = oracle.softmax_ext(s.reshape(3, 2, n), np.zeros((2, n), dtype=np.float32), 1.0).reshape(6, n)
    ds, do = ctx.to_device(s), ctx.empty(s.shape)
    ctx.check(ctx.L.ps_hip_soft_max(ctx.h, C.byref(do.tensor()), C.byref(ds.tensor())))
    assert np.array_equal(bits(do.numpy()), bits(want))
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "tiny-llama", 8, n_ctx=64, seed=1)
    hm = host.HostModel(d, max_batch=8)
    assert np.array_equal(bits(hm.graph_softmax(s)), bits(want))
    assert hm.get_n_tasks() == 1
    hm.close()


def test_kv_interface_members(ctx, oracle, tmp_path):
    """copy / move / mask / unmask / save_tokens / unmask_tokens / advance / rollback / truncate / append_tokens of KVCacheInterface and the
    deprecated add_cache on the device cache, through the C++ façade (HIPKV) and through the C entries."""
    from oracle import binding as B
    from powerserve_amd import hip, host, synth
    d = str(tmp_path / "m")
    mj = synth.write_model_dir(d, "tiny-llama", 8, n_ctx=64, seed=2)
    cfg = B.make_config(mj["llm_config"])
    kvd = cfg.kv_dim
    hm = host.HostModel(d, max_batch=8)
    prompt = np.random.default_rng(9).integers(0, cfg.vocab_size, 8)
    hm.forward(prompt[:6], np.arange(6), lm_head=False)  # slots 0..5, position 6
    assert hm.position == 6
    # add_cache: two rows behind the position (slots 6, 7), position unchanged
    rng = np.random.default_rng(1)
    k, v = rng.standard_normal((2, kvd)).astype(np.float32), rng.standard_normal((2, kvd)).astype(np.float32)
    hm.add_cache(1, k, v)
    assert hm.position == 6
    for i in range(2):
        kr, vr = hm.kv_read(1, 6 + i, kvd)
        assert np.array_equal(bits(kr), bits(k[i])) and np.array_equal(bits(vr), bits(v[i]))
    # copy(dst cache index, src TOKEN index) = move(dst, position + token): token 1 of the batch behind the position -> slot 3
    hm.kv("copy", 3, 1)
    kr, vr = hm.kv_read(1, 3, kvd)
    assert np.array_equal(bits(kr), bits(k[1])) and np.array_equal(bits(vr), bits(v[1]))
    hm.kv("move", 2, 6)
    kr, _ = hm.kv_read(1, 2, kvd)
    assert np.array_equal(bits(kr), bits(k[0]))
    hm.kv("save_tokens", 2)
    hm.kv("unmask_tokens", 2)
    assert hm.position == 6
    assert hm.kv("append_tokens", 2) == 6 and hm.position == 8
    assert hm.kv("rollback_tokens", 3) == 8 and hm.position == 5
    assert hm.kv("advance_tokens", 1) == 5 and hm.position == 6
    hm.kv("mask", 4)
    hm.kv("unmask", 4)
    with pytest.raises(host.HostError):
        hm.kv("mask", 60)  # POWERSERVE_ASSERT_KVCACHE(cache_index < position)
    with pytest.raises(host.HostError):
        hm.kv("save_tokens", 1000)  # beyond n_ctx
    assert hm.kv("truncate_tokens", 2) == 6 and hm.position == 2
    hm.close()
    # the same members as C entries on a bare device model
    gm = hip.Model(ctx, d, max_batch=8)
    gm.forward(prompt[:4], np.arange(4), lm_head=False)
    gm.forward_tree(prompt[4:7], [4, 5, 5], np.array([[1, 0, 0], [1, 1, 0], [1, 0, 1]], np.uint8), lm_head=False, advance=False)
    assert gm.position == 4
    k_before = gm.k_cache(0).copy()
    gm.kv_copy(4, 2)  # token 2 of the tree batch (slot 6) becomes slot 4
    assert np.array_equal(bits(gm.k_cache(0)[4]), bits(k_before[6]))
    gm.kv_save_tokens(1)
    gm.kv_unmask_tokens(1)
    assert gm.kv_append_tokens(1) == 4 and gm.position == 5
    assert ctx.L.ps_hip_model_kv_save_tokens(gm.h, 10_000) == 2
    gm.close()
