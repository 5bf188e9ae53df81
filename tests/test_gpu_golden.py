"""GPU: the HIP path against the committed golden vectors that the REAL reference produced (tests/golden/,
generator oracle/gen_golden.py).  No oracle library involved here: data only.  Everything is bit-exact."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
T = {"Q4_0": 2, "Q8_0": 8, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14}


def test_act_quant_golden(ctx):
    g = np.load(os.path.join(GOLD, "act_quant.npz"))
    for K in (256, 896, 4096):
        x = g[f"x_{K}"]
        for vdt, key in ((8, "q8_0"), (15, "q8_K")):
            if f"{key}_{K}" not in g.files:
                continue
            rs = ctx.L.ps_hip_row_size(vdt, K)
            out = ctx.empty((x.shape[0], rs), np.uint8)
            ctx.check(ctx.L.ps_hip_quantize_act(ctx.h, vdt, ctx.to_device(x).ptr, K, x.shape[0], out.ptr))
            assert np.array_equal(out.numpy(), g[f"{key}_{K}"])


@pytest.mark.parametrize("name,count", [("mul_mat", 12), ("mul_mat_wide", 16)])
def test_mul_mat_golden(ctx, name, count):
    """ps_hip_mul_mat against the real reference's outputs, Q6_K included.  `mul_mat_wide` holds the shapes that reach
    gemv4 (one column, K % 1024 == 0), gemm4k (MFMA, full / ragged / small batches) and the Q6_K kernels beyond a few rows."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = 0
    for key in g.files:
        if not key.startswith("y_"):
            continue
        _, a, b, K, N, bs = key.split("_")
        t, K, N = T[a + "_" + b], int(K), int(N)
        W = ctx.upload_weight(t, g["w_" + key[2:]], K, N)
        x = g["x_" + key[2:]]
        dx, dy = ctx.to_device(x), ctx.empty((x.shape[0], N))
        ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
        assert np.array_equal(dy.numpy().view(np.uint32), g[key].view(np.uint32)), key
        W.free()
        n += 1
    assert n == count


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for c in iter(lambda: f.read(1 << 20), b""):
            h.update(c)
    return h.hexdigest()


@pytest.mark.parametrize("preset,tn", [("tiny-llama", "Q4_0"), ("tiny-llama", "Q8_0"), ("tiny-qwen2", "Q8_0"), ("tiny-qwen2", "Q4_0")])
def test_e2e_golden(ctx, tmp_path, preset, tn):
    """logits and ids of the real LlamaModel/Qwen2Model::forward (reference binary), reproduced bit for bit"""
    from powerserve_amd import hip, synth
    g = np.load(os.path.join(GOLD, f"e2e_{preset}_{tn}.npz"))
    d = str(tmp_path / "m")
    synth.write_model_dir(d, preset, T[tn], n_ctx=int(g["n_ctx"]), seed=int(g["seed"]))
    assert _sha(os.path.join(d, "ggml", "weights.gguf")) == str(g["gguf_sha256"])
    m = hip.Model(ctx, d, max_batch=16)
    prompt = g["prompt"]
    assert np.array_equal(m.generate(prompt, 8, 24), g["ids"])
    # per-step logits with the reference's own prefill chunking (8, 8, 4)
    m.reset()
    for lo in range(0, 20, 8):
        hi = min(lo + 8, 20)
        m.forward(prompt[lo:hi], np.arange(lo, hi), lm_head=False)
    cur = int(prompt[-1])
    for s in range(24):
        lg, am = m.forward([cur], [m.position], lm_head=True)
        assert np.array_equal(lg[0].view(np.uint32), g["logits"][s].view(np.uint32)), s
        cur = int(g["ids"][s])
    m.reset()
    lg, _ = m.forward(prompt[:9], np.arange(9), lm_head=True)
    assert np.array_equal(lg.view(np.uint32), g["batch_logits"].view(np.uint32))
    m.close()


def _run_builds_fixture(ctx, tmp_path, preset, tn):
    """the e2e_builds fixture through the HIP library loaded in THIS process: (ids, per-step logits)"""
    from powerserve_amd import hip, synth
    g = np.load(os.path.join(GOLD, f"e2e_builds_{preset}_{tn}.npz"))
    d = str(tmp_path / "m")
    synth.write_model_dir(d, preset, T[tn], n_ctx=int(g["n_ctx"]), seed=int(g["seed"]))
    assert _sha(os.path.join(d, "ggml", "weights.gguf")) == str(g["gguf_sha256"])
    m = hip.Model(ctx, d, max_batch=16)
    prompt, bs, steps = g["prompt"], int(g["batch"]), len(g["ids_off"])
    ids = m.generate(prompt, bs, steps)
    m.reset()
    for lo in range(0, len(prompt) - 1, bs):
        hi = min(lo + bs, len(prompt) - 1)
        m.forward(prompt[lo:hi], np.arange(lo, hi), lm_head=False)
    cur, logits = int(prompt[-1]), []
    for s in range(steps):
        lg, _ = m.forward([cur], [m.position], lm_head=True)
        logits.append(lg[0].copy())
        cur = int(ids[s])
    m.close()
    return g, ids, np.stack(logits)


@pytest.mark.parametrize("preset,tn", [("tiny-llama", "Q4_0"), ("tiny-llama", "Q8_0"), ("tiny-qwen2", "Q8_0"), ("tiny-qwen2", "Q4_0")])
def test_e2e_against_both_reference_builds(ctx, tmp_path, preset, tn):
    """tests/golden/e2e_builds_*.npz: the real reference built with -ffp-contract=off ("off") and as its own CMake builds it, GCC's default
    -ffp-contract=fast ("fast").  The default library is the first build bit for bit; against the second it is as far away as the two builds
    are from each other (ids equal; 2.8e-3 of the largest logit on tiny-llama Q4_0, zero on the other three) -- reported, and bounded."""
    assert ctx.L.ps_hip_build_contract() == 0
    g, ids, logits = _run_builds_fixture(ctx, tmp_path, preset, tn)
    assert np.array_equal(ids, g["ids_off"]) and np.array_equal(logits.view(np.uint32), g["logits_off"].view(np.uint32))
    dev = float(np.abs(logits - g["logits_fast"]).max() / np.abs(g["logits_fast"]).max())
    print(f"[builds] {preset} {tn}: HIP (default) vs -ffp-contract=off: bit-exact; vs the stock -ffp-contract=fast build: ids equal "
          f"{np.array_equal(ids, g['ids_fast'])}, max logit deviation {dev:.3e} of the largest logit")
    assert np.array_equal(ids, g["ids_fast"]) and dev < 1e-2


def test_contract_build_is_the_stock_reference_build(tmp_path):
    """lib/libps_hip_contract.so (-DPS_CONTRACT: fused RoPE rotation and fused last n % 4 dot-product leftovers, powerserve_amd/build.py) in a process
    of its own (PS_HIP_LIB): every logit of every fixture model equals the stock -ffp-contract=fast build of the reference, bit for bit; the op level
    (RoPE modes 0 / 2, V.p behind 1..31 leftovers) against the oracle's contract mode; Q5_K is refused."""
    import subprocess
    import sys
    from powerserve_amd import hip
    lib = os.path.join(os.path.dirname(hip.LIB_PATH), "libps_hip_contract.so")
    assert os.path.exists(lib), "run __graft_entry__.build()"
    env = dict(os.environ, PS_HIP_LIB=lib, PS_CONTRACT_TMP=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "contract_check.py")], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "contract build: all checks passed" in r.stdout


@pytest.mark.parametrize("preset,tn", [("tiny-llama", "Q8_0"), ("tiny-qwen2", "Q4_0")])
@pytest.mark.parametrize("eager", [False, True])
def test_cache_longer_than_4096_tokens_against_the_real_reference(ctx, tmp_path, preset, tn, eager):
    """tests/golden/long_cache_*.npz (oracle/gen_golden_long.py): the REAL reference's forward of a 4 300-token prompt in chunks of 128 behind a window of
    4 608 slots, then 6 greedy steps.  Beyond 4 096 slots the single-token attention is the two-launch form, the batch soft-max keeps its rows in LDS and V.p
    walks more than one tile; the oracle takes minutes for such a prompt, which is why this case is a fixture.  Ids and every step's logits on bits."""
    from powerserve_amd import hip, synth
    g = np.load(os.path.join(GOLD, f"long_cache_{preset}_{tn}.npz"))
    d = str(tmp_path / "m")
    synth.write_model_dir(d, preset, T[tn], n_ctx=int(g["n_ctx"]), seed=int(g["seed"]))
    assert _sha(os.path.join(d, "ggml", "weights.gguf")) == str(g["gguf_sha256"])
    m = hip.Model(ctx, d, max_batch=256)
    if eager:
        m.set_mode(1)
    prompt, bs, steps = g["prompt"], int(g["batch"]), len(g["ids"])
    ids = m.generate(prompt, bs, steps)
    assert np.array_equal(ids, g["ids"]), (ids, g["ids"])
    m.reset()
    m.prefill(prompt[:-1], bs)  # (two reference chunks per launch sequence)
    cur = int(prompt[-1])
    for s in range(steps):
        lg, am = m.forward([cur], [m.position], lm_head=True)
        assert np.array_equal(lg[0].view(np.uint32), g["logits"][s].view(np.uint32)), s
        cur = int(g["ids"][s])
    m.close()


def test_scaled_rope_golden(ctx, tmp_path):
    """rope_freq_scale / rope_attn_factor off 1.0 (src/core/config.cpp:96,98 -> ggml.c:15344-15358): ps_hip_rope and whole-model generations against
    what the REAL reference produced (tests/golden/rope_scaled.npz, oracle/gen_golden_rope_scaled.py), bit for bit."""
    from powerserve_amd import hip, synth
    ops = ((0, 64, 1e4, 0.25, 1.0), (0, 128, 5e5, 0.5, 1.25), (2, 64, 1e6, 0.5, 0.75), (0, 64, 1e4, 1.0, 1.3), (2, 128, 5e5, 0.25, 0.8660254))
    e2e = (("tiny-llama", 8, 0.5, 1.25), ("tiny-qwen2", 2, 0.25, 0.8), ("tiny-llama", 2, 0.25, 1.0))  # (== the generator's tables: it imports the oracle binding, this test must not)
    g = np.load(os.path.join(GOLD, "rope_scaled.npz"))
    for i, (mode, hs, base, fs, af) in enumerate(ops):
        x, pos = g[f"op{i}_x"], g[f"op{i}_pos"]
        rp = hip.RopeParams(hs, 4096, base, fs, 0.0, af, 32.0, 0.0, mode)
        dx, dy = ctx.to_device(x), ctx.empty(x.shape)
        ctx.check(ctx.L.ps_hip_rope(ctx.h, C.byref(dy.tensor()), C.byref(dx.tensor()), pos.ctypes.data_as(C.c_void_p), pos.size, C.byref(rp)))
        assert np.array_equal(dy.numpy().view(np.uint32), g[f"op{i}_y"].view(np.uint32)), i
    for i, (preset, t, fs, af) in enumerate(e2e):
        d = str(tmp_path / f"m{i}")
        synth.write_model_dir(d, preset, t, n_ctx=128, seed=777 + i, rope_freq_scale=fs, rope_attn_factor=af)
        assert _sha(os.path.join(d, "ggml", "weights.gguf")) == str(g[f"e{i}_gguf_sha256"])
        m = hip.Model(ctx, d, max_batch=16)
        prompt = g[f"e{i}_prompt"]
        assert np.array_equal(m.generate(prompt, 8, 20), g[f"e{i}_ids"]), i
        m.reset()
        for lo in range(0, 22, 8):
            hi = min(lo + 8, 22)
            m.forward(prompt[lo:hi], np.arange(lo, hi), lm_head=False)
        cur = int(prompt[-1])
        for s in range(20):  # the fused single-token path (RoPE in the QKV launch's epilogue) step by step
            lg, _ = m.forward([cur], [m.position], lm_head=True)
            assert np.array_equal(lg[0].view(np.uint32), g[f"e{i}_logits"][s].view(np.uint32)), (i, s)
            cur = int(g[f"e{i}_ids"][s])
        m.close()
