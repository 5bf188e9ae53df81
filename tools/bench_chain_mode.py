import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, hip, synth
tmp = os.environ.get("TMPDIR", "/tmp")
d = os.path.join(tmp, "ps_bench_llama-3.1-8b_Q4_K_1234")
if not os.path.exists(d + "/.done"):
    synth.write_model_dir(d, "llama-3.1-8b", 12, n_ctx=4096, seed=1234); open(d + "/.done", "w").write("ok")
ctx = hip.Ctx(0)
m = hip.Model(ctx, d, max_batch=128, n_ctx=4096)
prompt = np.random.default_rng(42).integers(0, m.cfg.vocab_size, 2048).astype(np.int32)
for mode in (0, 2, 0, 2):
    m.set_mode(mode)
    ids0 = m.generate(prompt, 128, 8)
    m.decode_greedy(int(ids0[-1]), 8); ctx.sync()
    t0 = time.perf_counter(); ids = m.decode_greedy(5, 128); ctx.sync(); dt = time.perf_counter() - t0
    print("mode", mode, "decode tok/s", round(128 / dt, 1), "first ids", ids0[:4])
