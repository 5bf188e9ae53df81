"""oracle/gen_golden_long.py — TEST INFRASTRUCTURE ONLY.  Runs in the dev container (needs oracle/_ref/libps_ref.so).

Caches longer than 4096 tokens: the single-token attention leaves its one-launch form (n_ctx > 4096), a score row no longer fits the wave-per-row soft-max
and V.p walks more than one LDS tile.  The CPU oracle needs minutes for such a prompt, so the expected values are made HERE, by the real reference
(LlamaModel / Qwen2Model::forward, -ffp-contract=off build) and checked against the oracle before they are written: a 4 300-token prompt in chunks of 128
behind a window of 4 608 slots, then 6 greedy steps -> tests/golden/long_cache_*.npz (data only: prompt, ids, logits, the GGUF's hash).

    python oracle/gen_golden_long.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tensors  # noqa: E402
from oracle import binding as B  # noqa: E402
from oracle.gen_golden import sha  # noqa: E402
from powerserve_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MODELS = (("tiny-llama", B.Q8_0), ("tiny-qwen2", B.Q4_0))
SEED, N_CTX, N_PROMPT, STEPS, BATCH = 4321, 4608, 4300, 6, 128


def main():
    r, o = B.Ref(4), B.Oracle()
    for preset, t in MODELS:
        with tempfile.TemporaryDirectory() as td:
            mj = synth.write_model_dir(td, preset, t, n_ctx=N_CTX, seed=SEED)
            path = os.path.join(td, "ggml", "weights.gguf")
            cfg = B.make_config(mj["llm_config"])
            prompt = np.random.default_rng(SEED).integers(0, cfg.vocab_size, N_PROMPT).astype(np.int32)
            m = r.model(path, mj["model_arch"], cfg, 4)
            ids, logits, *_ = m.generate(prompt, BATCH, STEPS, want_logits=True)
            m.close()
            om = o.model(cfg, mj["model_arch"], load_tensors(path), n_threads=8)
            oids, ologits, *_ = om.generate(prompt, BATCH, STEPS, want_logits=True)
            om.close()
            same = np.array_equal(ids, oids) and np.array_equal(logits.view(np.uint32), ologits.view(np.uint32))
            print(f"{preset} {B.TYPE_NAMES[t]}: ids {ids.tolist()}, oracle bit-equal {same}", flush=True)
            assert same
            np.savez_compressed(os.path.join(OUT, f"long_cache_{preset}_{B.TYPE_NAMES[t]}.npz"), gguf_sha256=sha(path), prompt=prompt, seed=SEED, n_ctx=N_CTX,
                                batch=BATCH, ids=ids, logits=logits)


if __name__ == "__main__":
    main()
