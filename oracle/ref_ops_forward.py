"""TEST INFRASTRUCTURE (dev container only): a model forward made of the REAL reference's compiled operators
(oracle/_ref: powerserve_compute_forward_{mul_mat, rms_norm, rope, softmax_ext, add}, GGMLBackend::silu_hadamard /
get_embedding) sequenced like NormAttention::build (src/model/module/norm_attention.cpp:26-160), FFN::build
(src/model/module/ffn.cpp:22-42) and LlamaModel::forward (src/model/llama/llama_model.cpp:52-117) — but with the mask
tensor, the RoPE positions and the cache slots supplied by the caller instead of derived from `pos` by the executor's
GET_MASK (src/executor/executor.cpp:210-224), which cannot express a token tree.

Every arithmetic operation below is the reference's own binary; only the plumbing (views, strides, KV copies — byte
copies in the reference too, ggml.c:9341) is this file's.  The plumbing is pinned by the causal case:
tests/test_oracle_vs_ref.py requires forward() with the executor's mask to reproduce the real LlamaModel::forward /
Qwen2Model::forward logits bit for bit.  With a tree mask it is the pin for oracle/ps_oracle.c's
pso_model_forward_tree and the source of tests/golden/tree_forward.npz (oracle/gen_golden_tree.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding as B

F32 = 0


class RefOpsModel:
    def __init__(self, ref: "B.Ref", cfg, arch: str, tensors: dict):
        """tensors: name -> (ggml type, raw uint8 / float32 data, ne0, ne1) as tests' load_tensors() returns them."""
        self.r, self.cfg, self.arch, self.t = ref, cfg, arch, tensors
        c = cfg
        self.k_cache = [np.zeros((c.seq_len, c.kv_dim), dtype=np.float32) for _ in range(c.n_layers)]
        self.v_cache = [np.zeros((c.kv_dim, c.seq_len), dtype=np.float32) for _ in range(c.n_layers)]
        self.position = 0
        self.rp = B.RopeParams(c.rope.n_dims, c.rope.n_ctx_orig, c.rope.freq_base, c.rope.freq_scale, c.rope.ext_factor,
                               c.rope.attn_factor, c.rope.beta_fast, c.rope.beta_slow, c.rope.mode)

    # ---- weight helpers
    def _mm(self, name, x):
        t, data, k, n = self.t[name]
        return self.r.mul_mat(t, data, k, n, x)

    def _f32(self, name):
        return np.ascontiguousarray(self.t[name][1]).view(np.float32)

    def forward(self, tokens, slot0, rope_pos, mask, lm_head=True, advance=True):
        """tokens [n]; cache slots slot0 .. slot0 + n - 1; rope_pos [n]; mask float32 [n][n_kv] (0 / -inf), n_kv = mask.shape[1]."""
        r, c = self.r, self.cfg
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        rope_pos = np.ascontiguousarray(rope_pos, dtype=np.int32)
        mask = np.ascontiguousarray(mask, dtype=np.float32)
        bs, n_kv = tokens.size, mask.shape[1]
        dim, kvd, hs, nh, nkvh, nctx = c.dim, c.kv_dim, c.head_size, c.n_heads, c.n_kv_heads, c.seq_len
        assert mask.shape == (bs, n_kv) and slot0 + bs <= nctx and n_kv <= nctx
        te = self.t["token_embd.weight"]
        if te[0] in (0, 2, 8):
            x = r.get_embedding(te[0], te[1], dim, c.vocab_size, tokens)                   # [bs, dim]
        else:  # K-quant tables: GGMLBackend::get_embedding aborts (ggml_wrapper.cpp:199-205); the row through the reference's own dequantizer
            rs = r.row_size(te[0], dim)
            raw = np.ascontiguousarray(te[1]).view(np.uint8).reshape(-1)
            x = np.stack([r.dequantize(te[0], raw[int(tk) * rs:(int(tk) + 1) * rs], dim) for tk in tokens])
        scale = np.float32(1.0) / np.sqrt(np.float32(hs))
        for L in range(c.n_layers):
            p = f"blk.{L}."
            nrm = r.rms_norm(x, self._f32(p + "attn_norm.weight"), c.norm_eps)
            q, k, v = self._mm(p + "attn_q.weight", nrm), self._mm(p + "attn_k.weight", nrm), self._mm(p + "attn_v.weight", nrm)
            if self.arch == "qwen2":
                q = r.add(q, self._f32(p + "attn_q.bias"))
                k = r.add(k, self._f32(p + "attn_k.bias"))
                v = r.add(v, self._f32(p + "attn_v.bias"))
            qr = r.rope(q.reshape(bs, nh, hs), rope_pos, self.rp)                          # [bs, nh, hs]
            kr = r.rope(k.reshape(bs, nkvh, hs), rope_pos, self.rp)
            # store kv (norm_attention.cpp:78-105): same-type copies are byte copies (ggml.c:9341)
            self.k_cache[L][slot0:slot0 + bs] = kr.reshape(bs, kvd)
            self.v_cache[L][:, slot0:slot0 + bs] = v.T
            # kq = mat_mul(K view {hs, n_kv, nkvh}, q permuted {hs, bs, nh})  (norm_attention.cpp:115-129)
            K, V = self.k_cache[L], self.v_cache[L]
            qc = np.ascontiguousarray(qr)
            kq = np.empty((nh, bs, n_kv), dtype=np.float32)
            tk = B.ref_tensor(K, F32, [hs, n_kv, nkvh], nb=[4, 4 * kvd, 4 * hs, 4 * hs * nkvh])
            tq = B.ref_tensor(qc, F32, [hs, bs, nh], nb=[4, 4 * dim, 4 * hs, 4 * dim * bs])
            r.mul_mat_t(B.ref_tensor(kq, F32, [n_kv, bs, nh]), tk, tq)
            sm = r.softmax_ext(kq, mask, float(scale))                                     # [nh, bs, n_kv]
            # kqv = mat_mul(V view {n_kv, hs, nkvh}, sm {n_kv, bs, nh}) -> {hs, bs, nh}  (norm_attention.cpp:138-147)
            kqv = np.empty((nh, bs, hs), dtype=np.float32)
            tv = B.ref_tensor(V, F32, [n_kv, hs, nkvh], nb=[4, 4 * nctx, 4 * nctx * hs, 4 * nctx * hs * nkvh])
            r.mul_mat_t(B.ref_tensor(kqv, F32, [hs, bs, nh]), tv, B.ref_tensor(sm, F32, [n_kv, bs, nh]))
            att = np.ascontiguousarray(kqv.transpose(1, 0, 2)).reshape(bs, dim)            # permute {0,2,1,3} + cont
            x = r.add(x, self._mm(p + "attn_output.weight", att))
            nrm = r.rms_norm(x, self._f32(p + "ffn_norm.weight"), c.norm_eps)
            hb = r.silu_hadamard(self._mm(p + "ffn_gate.weight", nrm), self._mm(p + "ffn_up.weight", nrm))
            x = r.add(x, self._mm(p + "ffn_down.weight", hb))
        out = None
        if lm_head:
            nrm = r.rms_norm(x, self._f32("output_norm.weight"), c.norm_eps)
            out = self._mm("output.weight" if "output.weight" in self.t else "token_embd.weight", nrm)
        if advance:
            self.position = slot0 + bs
        return out

    # ---- the two ways of calling it
    def forward_causal(self, tokens, pos, lm_head=True):
        """exactly what LlamaModel::forward does with its `pos`: n_kv = pos.back() + 1, mask j <= pos[i]"""
        pos = np.asarray(pos, dtype=np.int64)
        n_kv = int(pos[-1]) + 1
        mask = np.where(np.arange(n_kv)[None, :] <= pos[:, None], np.float32(0), np.float32(-np.inf)).astype(np.float32)
        return self.forward(tokens, int(pos[0]), pos, mask, lm_head, advance=True)

    def forward_tree(self, tokens, rope_pos, tree=None, kv_vis=None, lm_head=True, advance=False):
        """the token-tree forward of ps_hip_model_forward_tree / pso_model_forward_tree"""
        n, p0 = len(tokens), self.position
        n_kv = p0 + n
        vis = np.ones((n, n_kv), dtype=bool)
        if kv_vis is not None:
            vis[:, :p0] = np.asarray(kv_vis, dtype=np.uint8)[:p0][None, :] != 0
        vis[:, p0:] = (np.asarray(tree, dtype=np.uint8) != 0) if tree is not None else np.tril(np.ones((n, n), dtype=bool))
        mask = np.where(vis, np.float32(0), np.float32(-np.inf)).astype(np.float32)
        return self.forward(tokens, p0, rope_pos, mask, lm_head, advance=advance)
