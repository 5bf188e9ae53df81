"""Run by tests/test_gpu_golden.py::test_contract_build_is_the_stock_reference_build in a process of its own with PS_HIP_LIB = lib/libps_hip_contract.so
(the -DPS_CONTRACT build).  Not collected by pytest (no test_ prefix): exits non-zero on the first failed check."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from conftest import load_tensors  # noqa: E402
from oracle import binding as B  # noqa: E402
from powerserve_amd import hip, synth  # noqa: E402
import test_gpu_golden as G  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main():
    import pathlib
    tmp = pathlib.Path(os.environ.get("PS_CONTRACT_TMP") or tempfile.mkdtemp())
    ctx = hip.Ctx(0)
    assert ctx.L.ps_hip_build_contract() == 1, "PS_HIP_LIB must point at libps_hip_contract.so"
    o = B.Oracle()
    o.L.pso_set_contract(1)
    # 1. the fixtures of the real reference's stock build
    for preset, tn in (("tiny-llama", "Q4_0"), ("tiny-llama", "Q8_0"), ("tiny-qwen2", "Q8_0"), ("tiny-qwen2", "Q4_0")):
        sub = tmp / f"{preset}_{tn}"
        sub.mkdir(exist_ok=True)
        g, ids, logits = G._run_builds_fixture(ctx, sub, preset, tn)
        assert np.array_equal(ids, g["ids_fast"]), (preset, tn)
        assert np.array_equal(bits(logits), bits(g["logits_fast"])), (preset, tn)
        off = int((bits(logits) != bits(g["logits_off"])).sum())
        print(f"[contract] {preset} {tn}: == the stock (-ffp-contract=fast) reference build bit for bit; {off} logits away from the -ffp-contract=off build")
    # 2. op level against the oracle's contract mode: RoPE ...
    rng = np.random.default_rng(3)
    for mode, hs, base in ((0, 64, 1e4), (2, 64, 1e6), (0, 128, 5e5)):
        pos = np.array([0, 1, 17, 2047, 4095], dtype=np.int32)
        x = rng.standard_normal((pos.size, 8, hs)).astype(np.float32)
        rp = hip.RopeParams(hs, 4096, base, 1.0, 0.0, 1.0, 32.0, 0.0, mode)
        dx, dy = ctx.to_device(x), ctx.empty(x.shape)
        ctx.check(ctx.L.ps_hip_rope(ctx.h, C.byref(dy.tensor()), C.byref(dx.tensor()), pos.ctypes.data_as(C.c_void_p), pos.size, C.byref(rp)))
        want = o.rope(x, pos, B.RopeParams(hs, 4096, base, 1.0, 0.0, 1.0, 32.0, 0.0, mode))
        assert np.array_equal(bits(dy.numpy()), bits(want)), ("rope", mode, hs)
    # ... and F32 mat-mul rows with every leftover count (ggml_vec_dot_f32)
    o.L.pso_vec_dot_f32.restype = C.c_float
    o.L.pso_vec_dot_f32.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
    for K in (33, 34, 35, 36, 39, 47, 63, 97, 2079):
        w, x = rng.standard_normal((24, K)).astype(np.float32), rng.standard_normal((3, K)).astype(np.float32)
        dw, dx, dy = ctx.to_device(w), ctx.to_device(x), ctx.empty((3, 24))
        ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(dw.tensor()), C.byref(dx.tensor())))
        want = np.array([[o.L.pso_vec_dot_f32(K, w[n].ctypes.data, x[b].ctypes.data) for n in range(24)] for b in range(3)], dtype=np.float32)
        assert np.array_equal(bits(dy.numpy()), bits(want)), ("f32 dot", K)
    # 3. whole models against the oracle's contract mode: Q4_K (the headline's kernels), wide prefill chunks (matrix-core attention with leftovers),
    #    one-launch and two-launch single-token attention, head size 128
    for preset, wt, chunk in (("tiny-llama", 12, 32), ("small-llama-hs128", 12, 32), ("tiny-qwen2", 8, 8)):
        d = str(tmp / f"m_{preset}_{wt}")
        mj = synth.write_model_dir(d, preset, wt, n_ctx=256, seed=21)
        cfg = B.make_config(mj["llm_config"])
        om = o.model(cfg, mj["model_arch"], load_tensors(os.path.join(d, "ggml/weights.gguf")), n_threads=8)
        prompt = np.random.default_rng(5).integers(0, cfg.vocab_size, 75)
        want_ids, want_logits, *_ = om.generate(prompt, chunk, 40, want_logits=True)
        for mode in (0, 16):
            gm = hip.Model(ctx, d, max_batch=32)
            gm.set_mode(mode)
            assert np.array_equal(gm.generate(prompt, chunk, 40), want_ids), (preset, mode)
            gm.reset()
            done = 0
            while done < prompt.size - 1:
                bs = min(chunk, prompt.size - 1 - done)
                gm.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
                done += bs
            cur = int(prompt[-1])
            for s in range(40):
                lg, _ = gm.forward([cur], [gm.position], lm_head=True)
                assert np.array_equal(bits(lg[0]), bits(want_logits[s])), (preset, mode, s)
                cur = int(want_ids[s])
            gm.close()
        om.close()
        print(f"[contract] {preset} wt {wt}: == oracle (contract mode) bit for bit, one- and two-launch attention")
    # 4. Q5_K is refused by this build
    d = str(tmp / "m_q5k")
    synth.write_model_dir(d, "tiny-llama", 13, n_ctx=64, seed=2)
    try:
        hip.Model(ctx, d, max_batch=8)
    except hip.PSHipError as e:
        assert "PS_CONTRACT" in str(e), e
    else:
        raise AssertionError("the contract build took Q5_K weights")
    print("contract build: all checks passed")


if __name__ == "__main__":
    main()
