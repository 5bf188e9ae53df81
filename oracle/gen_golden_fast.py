"""oracle/gen_golden_fast.py — TEST INFRASTRUCTURE ONLY.  Runs in the dev container (needs both oracle/_ref/libps_ref.so and libps_ref_fast.so:
`make -C oracle ref ref_fast`).

The reference's own CMake sets no floating-point contraction flag (CMakeLists.txt:24-33, libs/ggml/src/CMakeLists.txt:1173), so a stock build on an
FMA machine is GCC's default -ffp-contract=fast: a*b+c in scalar C code is fused.  On the hot path that changes exactly three places (the RoPE rotation
ggml.c:15455-15475, the n % 32 leftovers of ggml_vec_dot_f32 ggml.c:2123-2125, Q5_K's summs ggml-quants.c:8411; tests/test_ref_fast.py).  The oracle, the
golden vectors of gen_golden.py and the default HIP library follow the build WITHOUT contraction (oracle/Makefile CONTRACT=off); pso_set_contract(1) and
lib/libps_hip_contract.so follow the contracted one.  This script writes the fixtures that pin BOTH: the real LlamaModel / Qwen2Model::forward of the two
builds on the same GGUF bytes and prompt -- 41 prompt tokens + 42 greedy steps, long enough that the two builds part on one of the four models (tiny-llama
Q4_0: 512 of 21 504 logits, 2.8e-3 of the largest; ids equal) -> tests/golden/e2e_builds_*.npz (data only).

    python oracle/gen_golden_fast.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402
from oracle.gen_golden import sha  # noqa: E402
from powerserve_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MODELS = (("tiny-llama", B.Q4_0), ("tiny-llama", B.Q8_0), ("tiny-qwen2", B.Q8_0), ("tiny-qwen2", B.Q4_0))
SEED, N_PROMPT, STEPS, BATCH = 1234, 41, 42, 8


def main():
    builds = {"off": B.Ref(2), "fast": B.Ref(2, so=B.REF_FAST_SO)}
    for preset, t in MODELS:
        with tempfile.TemporaryDirectory() as td:
            mj = synth.write_model_dir(td, preset, t, n_ctx=128, seed=SEED)
            path = os.path.join(td, "ggml", "weights.gguf")
            cfg = B.make_config(mj["llm_config"])
            prompt = np.random.default_rng(SEED).integers(0, cfg.vocab_size, N_PROMPT).astype(np.int32)
            d = dict(gguf_sha256=sha(path), prompt=prompt, seed=SEED, n_ctx=128, batch=BATCH)
            for name, r in builds.items():
                m = r.model(path, mj["model_arch"], cfg, 2)
                ids, logits, *_ = m.generate(prompt, BATCH, STEPS, want_logits=True)
                m.close()
                d["ids_" + name], d["logits_" + name] = ids, logits
        lo, lf = d["logits_off"], d["logits_fast"]
        print(f"{preset} {B.TYPE_NAMES[t]}: ids equal {np.array_equal(d['ids_off'], d['ids_fast'])}, {int((lo.view(np.uint32) != lf.view(np.uint32)).sum())} of {lo.size} "
              f"logits differ, max |fast - off| / max |off| = {float(np.abs(lf - lo).max() / np.abs(lo).max()):.3e}")
        np.savez_compressed(os.path.join(OUT, f"e2e_builds_{preset}_{B.TYPE_NAMES[t]}.npz"), **d)


if __name__ == "__main__":
    main()
