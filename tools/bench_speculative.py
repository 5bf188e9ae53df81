#!/usr/bin/env python3
"""BASELINE.json config 4: Llama-3.1-8B target + Llama-3.2-1B draft, speculative decode on one MI355X (synthetic weights).
Random-init draft and target are unrelated, so almost nothing is accepted: the number that means something here is the
cost of one iteration (draft forwards + one 12-node tree verify + KV bookkeeping) and that the output follows the target's
greedy output (the KV entries of accepted nodes were computed inside the tree batch, i.e. with a different summation order
in the softmax than single-token decode: with flat random-init logits a near-tie can flip after a while, exactly as in the
reference; tests/test_gpu_speculative.py checks equality on models with healthy margins).  --self-draft uses the target as its own draft to exercise long accepted paths.
usage: bench_speculative.py [--steps N] [--prompt-len P] [--self-draft] [--wtype Q4_K]"""
import argparse, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import gguf, host, synth

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--prompt-len", type=int, default=256)
ap.add_argument("--n-ctx", type=int, default=1024)
ap.add_argument("--target", default="llama-3.1-8b")
ap.add_argument("--draft", default="llama-3.2-1b")
ap.add_argument("--wtype", default="Q4_K")
ap.add_argument("--draft-wtype", default="Q4_0")
ap.add_argument("--self-draft", action="store_true")
ap.add_argument("--truncated-draft", type=int, default=0, help="draft = the target's first K layers + its own output norm / lm_head (a truncated self-draft: the acceptance an unrelated "
                                                                 "random-init draft cannot have; /root/reference/README.md:16,18 reports 1.7-1.8x from this path on trained pairs)")
ap.add_argument("--late-scale", type=float, default=1.0, help="with --truncated-draft K: the target's layers >= K have their output projections drawn this much smaller "
                                                             "(synth.write_model_dir late_layers): the knob that sets how predictive the K-layer draft is")
a = ap.parse_args()

tmp = os.environ.get("TMPDIR", "/tmp")
def model_dir(preset, wt, seed):
    late = (a.truncated_draft, a.late_scale) if (a.truncated_draft and a.late_scale != 1.0 and preset == a.target) else None
    d = os.path.join(tmp, f"ps_spec_{preset}_{wt}_{seed}_{a.n_ctx}" + (f"_late{late[0]}x{late[1]}" if late else ""))
    if not os.path.exists(os.path.join(d, ".done")):
        synth.write_model_dir(d, preset, gguf.NAME_TYPE[wt], n_ctx=a.n_ctx, seed=seed, late_layers=late)
        open(os.path.join(d, ".done"), "w").write("ok")
    return d

target = host.HostModel(model_dir(a.target, a.wtype, 1234), max_batch=128, n_ctx=a.n_ctx)
# --self-draft: a second instance of the same weights (the draft needs its own KV cache)
if a.truncated_draft:
    dd = os.path.join(tmp, f"ps_spec_{a.target}_{a.wtype}_1234_{a.n_ctx}_first{a.truncated_draft}_late{a.late_scale}")
    if not os.path.exists(os.path.join(dd, ".done")):
        synth.truncate_model_dir(model_dir(a.target, a.wtype, 1234), dd, a.truncated_draft)
        open(os.path.join(dd, ".done"), "w").write("ok")
    draft = host.HostModel(dd, max_batch=128, n_ctx=a.n_ctx)
else:
    draft = host.HostModel(model_dir(a.target, a.wtype, 1234) if a.self_draft else model_dir(a.draft, a.draft_wtype, 99), max_batch=128, n_ctx=a.n_ctx)
prompt = np.random.default_rng(42).integers(0, target.vocab, a.prompt_len).astype(np.int32)

t0 = time.perf_counter(); want = target.generate(prompt, 128, a.steps); t_plain = time.perf_counter() - t0
host.spec_generate(target, draft, prompt, 128, 8)  # warm-up (one-time kernel attribute calls)
t0 = time.perf_counter(); got, st = host.spec_generate(target, draft, prompt, 128, a.steps); t_spec = time.perf_counter() - t0
# the prefill share of both timings (the same prompt on both sides; the speculative run prefills the draft as well): one-token runs
t0 = time.perf_counter(); target.generate(prompt, 128, 1); t_plain1 = time.perf_counter() - t0
t0 = time.perf_counter(); _, st1 = host.spec_generate(target, draft, prompt, 128, 1); t_spec1 = time.perf_counter() - t0
# How far from the plain path's greedy choice is every emitted token?  Teacher-force the speculative output through
# single-token forwards: gap = (top logit) - (logit of the token the speculative run emitted next); 0 where they agree.
target.reset()
done = 0
while done < a.prompt_len - 1:
    bs = min(128, a.prompt_len - 1 - done); target.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False); done += bs
cur, gaps, stds = int(prompt[-1]), [], []
for i in range(a.steps):
    lg = target.forward([cur], [target.position], lm_head=True)
    gaps.append(float(lg[0].max() - lg[0][int(got[i])])); stds.append(float(lg[0].std()))
    cur = int(got[i])
it = max(st["n_iterations"], 1)
dname = f"its own first {a.truncated_draft} layers + lm_head (layers >= {a.truncated_draft}: output projections x {a.late_scale})" if a.truncated_draft else ("itself" if a.self_draft else a.draft + " " + a.draft_wtype)
ms_plain_tok = 1e3 * (t_plain - t_plain1) / max(a.steps - 1, 1)  # decode only
ms_spec_it = 1e3 * (t_spec - t_spec1) / max(it - st1["n_iterations"], 1)
print(json.dumps({"config": f"{a.target} {a.wtype} target + {dname} draft, tree of 12, prompt {a.prompt_len}, {a.steps} tokens",
                  "plain_greedy_ms_per_token": ms_plain_tok, "speculative_ms_per_iteration_decode_only": ms_spec_it,
                  "break_even_tokens_per_iteration": ms_spec_it / ms_plain_tok, "decode_speedup_vs_plain_greedy": ms_plain_tok * (st["n_generated_tokens"] / it) / ms_spec_it,
                  "matching_prefix": int(np.argmax(np.append(got != want, True))), "max_logit_gap_vs_single_token_greedy": max(gaps),
                  "max_gap_in_sigma": float(max(g / s for g, s in zip(gaps, stds))), "n_tokens_with_gap": int(sum(g > 0 for g in gaps)), "logit_std": float(np.mean(stds)),
                  "tokens": int(a.steps), "speculative_s_incl_prefill": t_spec, "plain_greedy_s_incl_prefill": t_plain,
                  "tokens_per_iteration": st["n_generated_tokens"] / it, "draft_forwards_per_iteration": st["n_draft_times"] / it,
                  "accept_ratio": st["n_accepted_tokens"] / max(st["n_draft_tokens"], 1), "iterations": st["n_iterations"],
                  "ms_per_iteration": 1e3 * t_spec / it}))
