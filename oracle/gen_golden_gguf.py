"""Generates tests/golden/ref_written_model/ — a model directory whose ggml/weights.gguf was written by the REFERENCE's
own GGUF writer (libs/ggml/src/ggml.c gguf_write_to_file, through oracle/ref_gguf.cpp), not by this repository's
powerserve_amd/gguf.py.  It is what a stock llama.cpp "Q4_K_M" Llama-3 file looks like in miniature:
  * Q4_K tensors with Q6_K for attn_v / ffn_down of the "more bits" layers and for output.weight;
  * a `rope_freqs.weight` F32 tensor (Llama-3 long-context factors) that the reference never loads or applies
     and that this backend must therefore ignore as well;
  * tokenizer arrays and one metadata key of every GGUF value type in front of the tensor table;
  * general.alignment = 64 instead of the default 32.
The weights are the seeded synthetic ones of powerserve_amd/synth.py (tiny-llama, seed 4321), so the expected logits can
be produced by any path from the same tensors.  Run in the dev container:  python oracle/gen_golden_gguf.py"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402
from powerserve_amd import gguf, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_written_model")


def main():
    ref = B.Ref()
    with tempfile.TemporaryDirectory() as td:
        mj = synth.write_model_dir(td, "tiny-llama", synth.Q4_K_M, n_ctx=128, seed=4321, model_id="tiny-llama-Q4_K_M-refwriter")
        rd = gguf.GGUFReader(os.path.join(td, "ggml", "weights.gguf"))
        tensors = [(t.name, t.type, tuple(t.ne), np.array(rd.data(t.name))) for t in rd.tensors.values()]
    hs = mj["llm_config"]["head_size"]
    # Llama-3.1 style factors: 1 for the high-frequency half, growing towards 8 for the low frequencies
    freqs = np.concatenate([np.ones(hs // 4), np.linspace(1.0, 8.0, hs // 2 - hs // 4)]).astype(np.float32)
    tensors.insert(1, ("rope_freqs.weight", gguf.F32, (hs // 2,), freqs))
    os.makedirs(os.path.join(OUT, "ggml"), exist_ok=True)
    path = os.path.join(OUT, "ggml", "weights.gguf")
    B.ref_gguf_write(ref, path, mj["model_arch"], mj["model_id"], 64, tensors)
    with open(os.path.join(OUT, "model.json"), "w") as f:
        json.dump(mj, f, indent=1)
    types = sorted({gguf.TYPE_NAME[t[1]] for t in tensors})
    print("wrote", path, os.path.getsize(path), "bytes;", len(tensors), "tensors of types", types)


if __name__ == "__main__":
    main()
