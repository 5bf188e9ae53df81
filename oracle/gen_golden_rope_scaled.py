"""oracle/gen_golden_rope_scaled.py — TEST INFRASTRUCTURE ONLY.  Runs in the dev container (needs oracle/_ref).

tests/golden/rope_scaled.npz: the two RoPE parameters the reference reads from model.json but no public preset moves off 1.0 —
`rope_freq_scale` (linear position interpolation) and `rope_attn_factor` (the magnitude the cos / sin table is multiplied with)
(/root/reference/src/core/config.cpp:96,98 -> rope_compute_params -> ggml_rope_cache_init / rope_yarn,
libs/ggml/src/ggml.c:15319-15358).  Inputs + the outputs of the REAL reference (oracle/_ref/libps_ref.so): the operator
(both rotation modes) and whole-model generations through LlamaModel / Qwen2Model::forward.  Data only.

    python oracle/gen_golden_rope_scaled.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402
from powerserve_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "rope_scaled.npz")
# (mode, head size, base, freq_scale, attn_factor)
OPS = ((0, 64, 1e4, 0.25, 1.0), (0, 128, 5e5, 0.5, 1.25), (2, 64, 1e6, 0.5, 0.75), (0, 64, 1e4, 1.0, 1.3), (2, 128, 5e5, 0.25, 0.8660254))
# (preset, weight type, freq_scale, attn_factor)
E2E = (("tiny-llama", B.Q8_0, 0.5, 1.25), ("tiny-qwen2", B.Q4_0, 0.25, 0.8), ("tiny-llama", B.Q4_0, 0.25, 1.0))


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    r = B.Ref(2)
    d = {}
    rng = np.random.default_rng(606)
    for i, (mode, hs, base, fs, af) in enumerate(OPS):
        x = rng.standard_normal((6, 4, hs)).astype(np.float32)
        pos = np.array([0, 1, 17, 1000, 2047, 4095], dtype=np.int32)
        d[f"op{i}_x"], d[f"op{i}_pos"] = x, pos
        d[f"op{i}_y"] = r.rope(x, pos, B.RopeParams(hs, 4096, base, fs, 0.0, af, 32.0, 0.0, mode))
    for i, (preset, t, fs, af) in enumerate(E2E):
        with tempfile.TemporaryDirectory() as td:
            mj = synth.write_model_dir(td, preset, t, n_ctx=128, seed=777 + i, rope_freq_scale=fs, rope_attn_factor=af)
            path = os.path.join(td, "ggml", "weights.gguf")
            cfg = B.make_config(mj["llm_config"])
            m = r.model(path, mj["model_arch"], cfg, 2)
            prompt = np.random.default_rng(43 + i).integers(0, cfg.vocab_size, 23).astype(np.int32)
            ids, logits, *_ = m.generate(prompt, 8, 20, want_logits=True)
            m.close()
            d[f"e{i}_gguf_sha256"], d[f"e{i}_prompt"], d[f"e{i}_ids"], d[f"e{i}_logits"] = sha(path), prompt, ids, logits
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
