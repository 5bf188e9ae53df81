"""Synthetic PowerServe model directories (model.json + ggml/weights.gguf) with random-init weights.

There is no network for checkpoints, so benchmark/test weights are generated directly in the *quantized
domain*: every GGUF block gets random quants and a random fp16 scale chosen so that the dequantized
weights are ~zero-mean with std ≈ `std`.  No weight quantizer is involved (quantizing weights is an offline
tool step in the reference — tools/convert_hf_to_gguf — and is not on the hot path).

Directory layout and model.json schema are the reference's (tools/gguf_export.py:112-175,
src/core/config.cpp:68-120); tensor names are src/model/common/weights.hpp:26-69.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import gguf
from .gguf import F32, Q4_0, Q4_K, Q5_K, Q6_K, Q8_0

# SURVEY.md §8 shape table (public HF configs)
PRESETS = {
    # name: (arch, dim, hidden, L, n_heads, n_kv, head, vocab, rope_base, rope_type, tied, eps)
    "qwen2-0.5b": ("qwen2", 896, 4864, 24, 14, 2, 64, 151936, 1e6, 2, True, 1e-6),
    "llama-3.2-1b": ("llama", 2048, 8192, 16, 32, 8, 64, 128256, 5e5, 0, True, 1e-5),
    "llama-3.1-8b": ("llama", 4096, 14336, 32, 32, 8, 128, 128256, 5e5, 0, False, 1e-5),
    "llama-8b-dims-4l": ("llama", 4096, 14336, 4, 32, 8, 128, 4096, 5e5, 0, False, 1e-5),  # kernel diagnostics
    "llama-1b-dims-2l": ("llama", 2048, 8192, 2, 32, 8, 64, 4096, 5e5, 0, True, 1e-5),      # config 2's layer shape (Q4_0), small vocabulary
    "qwen2-0.5b-dims-2l": ("qwen2", 896, 4864, 2, 14, 2, 64, 4096, 1e6, 2, True, 1e-6),     # config 1's layer shape (Q8_0: 896 and 4864 are not multiples of 256)
    # small shapes for parity tests (finish in seconds on the CPU oracle)
    "tiny-llama": ("llama", 256, 512, 2, 4, 2, 64, 512, 1e4, 0, False, 1e-5),
    "tiny-qwen2": ("qwen2", 256, 512, 2, 4, 2, 64, 512, 1e6, 2, True, 1e-6),
    "small-llama": ("llama", 512, 1536, 3, 8, 2, 64, 2048, 5e5, 0, True, 1e-5),
    "small-llama-hs128": ("llama", 1024, 2048, 2, 8, 2, 128, 1024, 5e5, 0, False, 1e-5),
    # a draft model for small-llama-hs128: same vocabulary, different width / depth / head size
    "small-llama-draft": ("llama", 512, 1024, 3, 8, 4, 64, 1024, 5e5, 0, True, 1e-5),
    # the headline's depth behind a long cache at a width the CPU oracle finishes in seconds (tests/test_gpu_fullsize.py)
    "deep-llama-hs128": ("llama", 1024, 2048, 32, 8, 2, 128, 1024, 5e5, 0, False, 1e-5),
    # >= 256 row groups in every mat-vec of a layer (one workgroup per CU)
    "wide-llama": ("llama", 2048, 4096, 2, 16, 4, 128, 2048, 5e5, 0, False, 1e-5),
    # head sizes and query-heads-per-kv-head ratios no public config of the survey has but the backend accepts (32 / 96; 1, 3, 5, 6, 8): tools/gpu_fuzz.py
    "odd-llama-hs96": ("llama", 768, 1024, 2, 8, 2, 96, 512, 1e4, 0, False, 1e-5),
    "odd-llama-hs32": ("llama", 256, 512, 2, 8, 8, 32, 512, 1e4, 0, True, 1e-5),
    "odd-qwen2-r3": ("qwen2", 384, 768, 2, 6, 2, 64, 512, 1e6, 2, True, 1e-6),
    "odd-llama-r5": ("llama", 320, 640, 2, 5, 1, 64, 512, 5e5, 0, False, 1e-5),
    "odd-llama-r6": ("llama", 768, 1024, 2, 6, 1, 128, 512, 5e5, 0, False, 1e-5),
    "odd-llama-r8": ("llama", 512, 1024, 2, 8, 1, 64, 512, 5e5, 0, True, 1e-5),
}


# (rope_freq_scale, rope_attn_factor) draws of the random sweeps (tools/cpu_fuzz_oracle.py, tools/gpu_fuzz.py): picked by the model seed, so a replayed draw gets the same pair
ROPE_DRAWS = [(1.0, 1.0), (0.5, 1.0), (0.25, 1.25), (1.0, 0.8)]


def llm_config(preset: str, n_ctx: int, rope_freq_scale: float = 1.0, rope_attn_factor: float = 1.0) -> dict:
    """model.json's llm_config of a preset.  rope_freq_scale / rope_attn_factor: the two RoPE parameters the reference reads from the model file
    (src/core/config.cpp:96,98 -> ggml_rope_cache_init, libs/ggml/src/ggml.c:15344-15358) that no public preset of the survey moves off 1.0."""
    arch, dim, hidden, L, nh, nkv, hs, vocab, base, rtype, tied, eps = PRESETS[preset]
    return {
        "embed_dim": dim, "ffn_dim": hidden, "n_layers": L, "n_attn_heads": nh, "n_attn_kv_heads": nkv,
        "n_ctx": n_ctx, "vocab_size": vocab, "kv_dim": nkv * hs, "head_size": hs, "norm_eps": eps,
        "rope_config": {"rope_dim": hs, "n_rope_ctx_orig": n_ctx, "rope_freq_base": base, "rope_freq_scale": float(rope_freq_scale),
                        "rope_attn_factor": float(rope_attn_factor), "rope_type": rtype},
    }


def _f16(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float16).view(np.uint16)


def random_blocks(rng: np.random.Generator, t: int, n_rows: int, k: int, std: float = 0.02) -> np.ndarray:
    """Random valid GGUF blocks for an [k, n_rows] weight; returns uint8 [n_rows * row_size]."""
    if t == F32:
        return (rng.standard_normal((n_rows, k), dtype=np.float32) * std).view(np.uint8).reshape(-1)
    blk, ts = gguf.BLOCK[t]
    nb = n_rows * (k // blk)
    out = np.empty((nb, ts), dtype=np.uint8)
    jit = np.exp(rng.standard_normal(nb, dtype=np.float32) * 0.25)  # log-normal scale jitter
    if t == Q4_0:  # w = d*(q-8), q~U[0,15]: std 4.61*d
        out[:, 0:2] = _f16(std / 4.61 * jit).reshape(-1, 1).view(np.uint8)
        out[:, 2:] = rng.integers(0, 256, (nb, 16), dtype=np.uint8)
    elif t == Q8_0:  # w = d*q, q~U[-127,127]: std 73.3*d
        out[:, 0:2] = _f16(std / 73.3 * jit).reshape(-1, 1).view(np.uint8)
        out[:, 2:] = rng.integers(-127, 128, (nb, 32), dtype=np.int8).view(np.uint8)
    elif t == Q4_K:  # w = d*sc*q - dmin*m; dmin = 8d, m = round(7.5*sc/8) -> ~zero-mean
        sc = rng.integers(8, 64, (nb, 8), dtype=np.uint8)
        m = np.minimum(63, np.rint(7.5 * sc / 8.0)).astype(np.uint8)
        d = std / (36.0 * 4.61) * jit
        out[:, 0:2] = _f16(d).reshape(-1, 1).view(np.uint8)
        out[:, 2:4] = _f16(8.0 * d).reshape(-1, 1).view(np.uint8)
        s = out[:, 4:16]
        s[:, 0:4] = sc[:, 0:4] | ((sc[:, 4:8] >> 4) << 6)  # inverse of get_scale_min_k4 (ggml-quants.c:1912-1920)
        s[:, 4:8] = m[:, 0:4] | ((m[:, 4:8] >> 4) << 6)
        s[:, 8:12] = (sc[:, 4:8] & 0xF) | ((m[:, 4:8] & 0xF) << 4)
        out[:, 16:] = rng.integers(0, 256, (nb, 128), dtype=np.uint8)
    elif t == Q5_K:  # w = d*sc*q - dmin*m, q~U[0,31]: std 9.23; dmin = 16d, m = round(15.5*sc/16) -> ~zero-mean
        sc = rng.integers(8, 64, (nb, 8), dtype=np.uint8)
        m = np.minimum(63, np.rint(15.5 * sc / 16.0)).astype(np.uint8)
        d = std / (36.0 * 9.23) * jit
        out[:, 0:2] = _f16(d).reshape(-1, 1).view(np.uint8)
        out[:, 2:4] = _f16(16.0 * d).reshape(-1, 1).view(np.uint8)
        s = out[:, 4:16]
        s[:, 0:4] = sc[:, 0:4] | ((sc[:, 4:8] >> 4) << 6)
        s[:, 4:8] = m[:, 0:4] | ((m[:, 4:8] >> 4) << 6)
        s[:, 8:12] = (sc[:, 4:8] & 0xF) | ((m[:, 4:8] & 0xF) << 4)
        out[:, 16:] = rng.integers(0, 256, (nb, 160), dtype=np.uint8)  # qh[32] + qs[128]
    elif t == Q6_K:  # w = d*sc*(q-32), q~U[0,63]: std 18.5
        out[:, 0:192] = rng.integers(0, 256, (nb, 192), dtype=np.uint8)
        out[:, 192:208] = rng.integers(-64, 64, (nb, 16), dtype=np.int8).view(np.uint8)
        out[:, 208:210] = _f16(std / (37.0 * 18.5) * jit).reshape(-1, 1).view(np.uint8)
    else:
        raise ValueError(t)
    return out.reshape(-1)


Q5_K_M = 1017  # pseudo type: llama.cpp's "Q5_K_M" recipe, Q5_K with Q6_K for the same sensitive tensors
Q4_K_M = 1015  # pseudo type: the per-tensor mix llama.cpp's "Q4_K_M" file type uses (Q4_K with Q6_K for the sensitive tensors)


def _more_bits(i: int, n: int) -> bool:  # llama.cpp use_more_bits(i_layer, n_layers)
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def tensor_plan(cfg: dict, arch: str, wtype: int, tied: bool, embd_type: int | None = None):
    """[(name, type, ne)] in file order."""
    dim, hid, L, kvd, vocab = cfg["embed_dim"], cfg["ffn_dim"], cfg["n_layers"], cfg["kv_dim"], cfg["vocab_size"]
    mix = wtype in (Q4_K_M, Q5_K_M)
    if mix:
        wtype = Q4_K if wtype == Q4_K_M else Q5_K
    et = (Q6_K if (mix and tied) else wtype) if embd_type is None else embd_type
    plan = [("token_embd.weight", et, (dim, vocab))]
    for i in range(L):
        b = f"blk.{i}."
        hi = Q6_K if (mix and _more_bits(i, L)) else wtype
        plan += [(b + "attn_norm.weight", F32, (dim,)), (b + "attn_q.weight", wtype, (dim, dim)),
                 (b + "attn_k.weight", wtype, (dim, kvd)), (b + "attn_v.weight", hi, (dim, kvd)),
                 (b + "attn_output.weight", wtype, (dim, dim)), (b + "ffn_norm.weight", F32, (dim,)),
                 (b + "ffn_gate.weight", wtype, (dim, hid)), (b + "ffn_up.weight", wtype, (dim, hid)),
                 (b + "ffn_down.weight", hi, (hid, dim))]
        if arch == "qwen2":
            plan += [(b + "attn_q.bias", F32, (dim,)), (b + "attn_k.bias", F32, (kvd,)), (b + "attn_v.bias", F32, (kvd,))]
    plan.append(("output_norm.weight", F32, (dim,)))
    if not tied:
        plan.append(("output.weight", Q6_K if mix else wtype, (dim, vocab)))
    return plan


def write_model_dir(out_dir: str, preset: str, wtype: int, n_ctx: int, seed: int = 1234, std: float = 0.02,
                    embd_type: int | None = None, model_id: str | None = None, rope_freq_scale: float = 1.0, rope_attn_factor: float = 1.0,
                    late_layers: tuple | None = None) -> dict:
    """Create <out_dir>/{model.json, ggml/weights.gguf}; returns the model.json dict.
    late_layers = (first_layer, factor): the output projections (attn_output, ffn_down) of layers >= first_layer are drawn `factor` times smaller -- a target whose
    late layers only refine what its early layers decide, so that a prefix of its own layers is a draft with measurable acceptance (tools/bench_speculative.py).
    The random stream is the one of the plain model (only scales differ)."""
    arch, *_rest = PRESETS[preset]
    tied = PRESETS[preset][10]
    cfg = llm_config(preset, n_ctx, rope_freq_scale, rope_attn_factor)
    os.makedirs(os.path.join(out_dir, "ggml"), exist_ok=True)
    mj = {"version": 1, "model_arch": arch, "model_id": model_id or f"{preset}-{'Q4_K_M' if wtype == Q4_K_M else ('Q5_K_M' if wtype == Q5_K_M else gguf.TYPE_NAME[wtype])}",
          "llm_config": cfg}
    with open(os.path.join(out_dir, "model.json"), "w") as f:
        json.dump(mj, f, indent=1)
    w = gguf.GGUFWriter(os.path.join(out_dir, "ggml", "weights.gguf"))
    w.add_kv("general.architecture", arch)
    w.add_kv("general.name", mj["model_id"])
    w.add_kv("general.alignment", gguf.ALIGNMENT)
    plan = tensor_plan(cfg, arch, wtype, tied, embd_type)
    for name, t, ne in plan:
        w.add_tensor(name, t, ne)
    rng = np.random.default_rng(seed)

    def produce(ti):
        if ti.type == F32:
            n = int(np.prod(ti.ne))
            if ti.name.endswith("norm.weight"):
                return (1.0 + 0.02 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
            return (0.02 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)  # biases
        k = ti.ne[0]
        rows = int(np.prod(ti.ne[1:]))
        # std scaled so that a K-long dot with unit-rms input stays O(1): 0.02 at K=4096
        s = std * (4096.0 / k) ** 0.5 if ti.name != "token_embd.weight" or not tied else std * (4096.0 / k) ** 0.5
        if late_layers and ti.name.startswith("blk.") and int(ti.name.split(".")[1]) >= late_layers[0] and (ti.name.endswith("attn_output.weight") or ti.name.endswith("ffn_down.weight")):
            s *= late_layers[1]
        return random_blocks(rng, ti.type, rows, k, s)

    w.write(produce)
    return mj


def truncate_model_dir(src_dir: str, out_dir: str, n_layers: int, model_id: str | None = None) -> dict:
    """A model that is the first `n_layers` layers of the one in `src_dir` plus ITS output norm / lm_head / embeddings (a "truncated self-draft" for speculative
    decoding: on synthetic weights an unrelated small model agrees with the target on nothing, a prefix of the target's own layers on a measurable share)."""
    mj = load_model_json(src_dir)
    L = int(mj["llm_config"]["n_layers"])
    assert 0 < n_layers <= L
    mj = json.loads(json.dumps(mj))
    mj["llm_config"]["n_layers"] = int(n_layers)
    mj["model_id"] = model_id or f"{mj['model_id']}-first{n_layers}"
    os.makedirs(os.path.join(out_dir, "ggml"), exist_ok=True)
    with open(os.path.join(out_dir, "model.json"), "w") as f:
        json.dump(mj, f, indent=1)
    rd = gguf.GGUFReader(os.path.join(src_dir, "ggml", "weights.gguf"))
    w = gguf.GGUFWriter(os.path.join(out_dir, "ggml", "weights.gguf"))
    w.add_kv("general.architecture", mj["model_arch"])
    w.add_kv("general.name", mj["model_id"])
    w.add_kv("general.alignment", gguf.ALIGNMENT)
    for name, ti in rd.tensors.items():
        if name.startswith("blk.") and int(name.split(".")[1]) >= n_layers:
            continue
        w.add_tensor(name, ti.type, ti.ne)
    w.write(lambda ti: np.array(rd.data(ti.name)))
    return mj


def load_model_json(model_dir: str) -> dict:
    with open(os.path.join(model_dir, "model.json")) as f:
        return json.load(f)
