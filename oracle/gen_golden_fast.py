"""oracle/gen_golden_fast.py — TEST INFRASTRUCTURE ONLY.  Runs in the dev container (needs oracle/_ref/libps_ref_fast.so: `make -C oracle ref_fast`).

The reference's own CMake sets no floating-point contraction flag (CMakeLists.txt:24-33, libs/ggml/src/CMakeLists.txt:1173), so a stock build on an
FMA machine is GCC's default -ffp-contract=fast: a*b+c in the scalar C code (RoPE rotation ggml.c:15368-15491, the soft-max / dot-product tails,
Q5_K summs, ...) is fused.  The oracle, the golden vectors of gen_golden.py and the HIP path follow the build WITHOUT contraction
(oracle/Makefile CONTRACT=off: every operation rounds where the C source rounds).  This script writes the SECOND pin: the e2e fixtures of gen_golden.py
(same seeds, same GGUF bytes, same prompts) through the contracted build -> tests/golden/e2e_fast_*.npz (data only: ids + logits), so that the
distance between the two legitimate builds of the reference -- and of the HIP path to each -- is a tested, stated number (tests/test_ref_fast.py,
tests/test_gpu_golden.py::test_e2e_against_both_reference_builds).

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


def run(r, preset, t):
    """what gen_golden.py's F5 section runs, through the given build of the reference"""
    with tempfile.TemporaryDirectory() as td:
        mj = synth.write_model_dir(td, preset, t, n_ctx=128, seed=1234)
        path = os.path.join(td, "ggml", "weights.gguf")
        cfg = B.make_config(mj["llm_config"])
        m = r.model(path, mj["model_arch"], cfg, 2)
        prompt = np.random.default_rng(42).integers(0, cfg.vocab_size, 21).astype(np.int32)
        ids, logits, *_ = m.generate(prompt, 8, 24, want_logits=True)
        m.reset()
        batch_logits = m.forward(prompt[:9], np.arange(9), True)
        m.close()
        return dict(gguf_sha256=sha(path), prompt=prompt, ids=ids, logits=logits, batch_logits=batch_logits)


def main():
    fast = B.Ref(2, so=B.REF_FAST_SO)
    for preset, t in MODELS:
        d = run(fast, preset, t)
        base = np.load(os.path.join(OUT, f"e2e_{preset}_{B.TYPE_NAMES[t]}.npz"))
        assert str(base["gguf_sha256"]) == d["gguf_sha256"] and np.array_equal(base["prompt"], d["prompt"])
        dev = float(np.abs(d["logits"] - base["logits"]).max() / np.abs(base["logits"]).max())
        ndiff = int((d["logits"].view(np.uint32) != base["logits"].view(np.uint32)).sum())
        print(f"{preset} {B.TYPE_NAMES[t]}: ids equal {np.array_equal(d['ids'], base['ids'])}, {ndiff} of {d['logits'].size} decode logits differ, "
              f"max |fast - off| / max |off| = {dev:.3e}")
        np.savez_compressed(os.path.join(OUT, f"e2e_fast_{preset}_{B.TYPE_NAMES[t]}.npz"), gguf_sha256=d["gguf_sha256"], ids=d["ids"], logits=d["logits"],
                            batch_logits=d["batch_logits"], contract="fast")


if __name__ == "__main__":
    main()
