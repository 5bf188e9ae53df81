"""oracle/binding.py — TEST INFRASTRUCTURE ONLY.

ctypes bindings for
  * oracle/libps_oracle.so      (my plain-C restatement, oracle/ps_oracle.c)            -> class Oracle
  * oracle/_ref/libps_ref.so    (the real reference compiled in place, oracle/Makefile) -> class Ref
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (powerserve_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libps_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libps_ref.so")
# the same sources compiled WITHOUT -ffp-contract=off: what the reference's own CMake produces on an FMA machine (oracle/Makefile, CONTRACT=fast)
REF_FAST_SO = os.path.join(HERE, "_ref", "libps_ref_fast.so")

# ggml_type values (libs/ggml/include/ggml.h:361-398)
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K, I32 = 0, 1, 2, 8, 12, 13, 14, 15, 26
TYPE_NAMES = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q8_0: "Q8_0", Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K", Q8_K: "Q8_K"}


def build(ref: bool = True) -> None:
    """(Re)build the oracle; the real reference only when /root/reference is present."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref and os.path.isdir("/root/reference/libs/ggml/src"):
        subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref"], check=True)
        subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref_fast"], check=True)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def have_ref_fast() -> bool:
    return os.path.exists(REF_FAST_SO)


class RopeParams(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("n_ctx_orig", C.c_int32), ("freq_base", C.c_float),
                ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float),
                ("beta_fast", C.c_float), ("beta_slow", C.c_float), ("mode", C.c_int32)]


class LLMConfig(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("hidden_dim", C.c_uint32), ("n_layers", C.c_uint32),
                ("n_heads", C.c_uint32), ("n_kv_heads", C.c_uint32), ("seq_len", C.c_uint32),
                ("vocab_size", C.c_uint32), ("kv_dim", C.c_uint32), ("head_size", C.c_uint32),
                ("norm_eps", C.c_float), ("rope", RopeParams)]


def make_config(d: dict) -> LLMConfig:
    """d: the llm_config dict of a PowerServe model.json (src/core/config.cpp:68-104)."""
    r = d["rope_config"]
    rp = RopeParams(int(r["rope_dim"]), int(r["n_rope_ctx_orig"]), float(r["rope_freq_base"]),
                    float(r["rope_freq_scale"]), 0.0, float(r["rope_attn_factor"]), 32.0, 0.0, int(r["rope_type"]))
    return LLMConfig(int(d["embed_dim"]), int(d["ffn_dim"]), int(d["n_layers"]), int(d["n_attn_heads"]),
                     int(d["n_attn_kv_heads"]), int(d["n_ctx"]), int(d["vocab_size"]), int(d["kv_dim"]),
                     int(d["head_size"]), float(d["norm_eps"]), rp)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------- restatement
class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = self.L = C.CDLL(ORACLE_SO)
        L.pso_row_size.restype = C.c_size_t
        L.pso_row_size.argtypes = [C.c_int, C.c_int64]
        L.pso_vec_dot_type.restype = C.c_int
        L.pso_vec_dot.restype = C.c_float
        L.pso_vec_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.pso_vec_dot_f32.restype = C.c_float
        L.pso_vec_dot_f32.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
        L.pso_from_float.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.pso_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.pso_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_int]
        L.pso_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float]
        L.pso_rope.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.pso_rope_cache.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
        L.pso_softmax_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_float]
        L.pso_silu_hadamard.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.pso_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        L.pso_get_embedding.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        L.pso_model_create.restype = C.c_void_p
        L.pso_model_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pso_model_destroy.argtypes = [C.c_void_p]
        L.pso_model_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64]
        L.pso_model_kv_position.restype = C.c_size_t
        L.pso_model_kv_position.argtypes = [C.c_void_p]
        L.pso_model_reset.argtypes = [C.c_void_p]
        L.pso_model_rollback.argtypes = [C.c_void_p, C.c_size_t]
        L.pso_model_rollback.restype = None
        L.pso_model_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.pso_model_forward_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.pso_model_kv_move.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.pso_model_kv_move.restype = None
        L.pso_model_kv_advance.argtypes = [C.c_void_p, C.c_size_t]
        L.pso_model_kv_advance.restype = None
        L.pso_model_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        L.pso_model_k_cache.restype = C.c_void_p
        L.pso_model_k_cache.argtypes = [C.c_void_p, C.c_int]
        L.pso_model_v_cache.restype = C.c_void_p
        L.pso_model_v_cache.argtypes = [C.c_void_p, C.c_int]

    def row_size(self, t, k):
        return self.L.pso_row_size(t, k)

    def vec_dot_type(self, t):
        return self.L.pso_vec_dot_type(t)

    def from_float(self, vdt, x):
        x = _f32(x)
        out = np.zeros(self.row_size(vdt, x.size), dtype=np.uint8)
        self.L.pso_from_float(vdt, _p(x), _p(out), x.size)
        return out

    def dequantize(self, t, blocks, k):
        out = np.empty(k, dtype=np.float32)
        b = np.ascontiguousarray(blocks)
        self.L.pso_dequantize_row(t, _p(b), _p(out), k)
        return out

    def mul_mat(self, t, w, K, N, x, n_threads=1, want_act=False):
        """w: uint8 blocks [N*row_size]; x: [bs, K] float32 -> y [bs, N]."""
        x = _f32(x).reshape(-1, K)
        bs = x.shape[0]
        y = np.empty((bs, N), dtype=np.float32)
        w = np.ascontiguousarray(w)
        act = np.zeros(bs * self.row_size(self.vec_dot_type(t), K), dtype=np.uint8) if want_act else None
        self.L.pso_mul_mat(t, _p(w), K, N, _p(x), bs, _p(y), _p(act), n_threads)
        return (y, act) if want_act else y

    def rms_norm(self, x, w, eps):
        x = _f32(x)
        w = _f32(w)
        y = np.empty_like(x)
        self.L.pso_rms_norm(_p(x), _p(w), _p(y), x.shape[-1], x.size // x.shape[-1], eps)
        return y

    def rope(self, x, pos, rp: RopeParams):
        """x: [npos, n_heads, head_size]."""
        x = _f32(x)
        pos = _i32(pos)
        y = np.empty_like(x)
        self.L.pso_rope(_p(x), _p(y), x.shape[2], x.shape[1], x.shape[0], _p(pos), C.byref(rp))
        return y

    def rope_cache(self, p, ne0, rp: RopeParams):
        out = np.empty(ne0, dtype=np.float32)
        self.L.pso_rope_cache(int(p), ne0, C.byref(rp), _p(out))
        return out

    def softmax_ext(self, x, mask, scale):
        """x: [n_heads, bs, n_kv]; mask: [bs, n_kv] or None."""
        x = _f32(x)
        m = _f32(mask) if mask is not None else None
        y = np.empty_like(x)
        self.L.pso_softmax_ext(_p(x), _p(m), _p(y), x.shape[2], x.shape[1], x.shape[0], scale)
        return y

    def silu_hadamard(self, g, u):
        g = _f32(g)
        u = _f32(u)
        y = np.empty_like(g)
        self.L.pso_silu_hadamard(_p(g), _p(u), _p(y), g.size)
        return y

    def add(self, a, b):
        a = _f32(a)
        b = _f32(b)
        y = np.empty_like(a)
        self.L.pso_add(_p(a), _p(b), _p(y), a.shape[-1], a.size // a.shape[-1], int(b.size != a.size))
        return y

    def get_embedding(self, t, table, dim, tokens):
        tokens = _i32(tokens)
        out = np.empty((tokens.size, dim), dtype=np.float32)
        table = np.ascontiguousarray(table)
        self.L.pso_get_embedding(t, _p(table), dim, _p(tokens), tokens.size, _p(out))
        return out

    def model(self, cfg: LLMConfig, arch: str, tensors: dict, n_threads=1):
        return OracleModel(self, cfg, arch, tensors, n_threads)


class OracleModel:
    """tensors: {name: (ggml_type, np.uint8/np.float32 array, ne0, ne1)} — arrays are kept alive here."""

    def __init__(self, o: Oracle, cfg: LLMConfig, arch: str, tensors: dict, n_threads: int):
        self.o, self.cfg, self.keep = o, cfg, []
        self.h = o.L.pso_model_create(C.byref(cfg), int(arch == "qwen2"), n_threads)
        for name, (t, arr, ne0, ne1) in tensors.items():
            arr = np.ascontiguousarray(arr)
            self.keep.append(arr)
            o.L.pso_model_set_tensor(self.h, name.encode(), t, _p(arr), ne0, ne1)

    def close(self):
        if self.h:
            self.o.L.pso_model_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def position(self):
        return self.o.L.pso_model_kv_position(self.h)

    def reset(self):
        self.o.L.pso_model_reset(self.h)

    def rollback(self, n):
        self.o.L.pso_model_rollback(self.h, int(n))

    def forward(self, tokens, pos, lm_head=True):
        tokens, pos = _i32(tokens), _i32(pos)
        out = np.empty((tokens.size, self.cfg.vocab_size), dtype=np.float32) if lm_head else None
        rc = self.o.L.pso_model_forward(self.h, _p(tokens), tokens.size, _p(pos), int(lm_head), _p(out))
        assert rc == 0
        return out

    def forward_tree(self, tokens, rope_pos, tree=None, kv_vis=None, lm_head=True, advance=False):
        """Token-tree forward: tokens at the slots [position, position + n), column i rotated with rope_pos[i], batch visibility
        tree[i][j] (None: causal), cache-slot visibility kv_vis[n_ctx] (None: all).  Returns logits [n, vocab] or None."""
        tokens, rope_pos = _i32(tokens), _i32(rope_pos)
        n = tokens.size
        tr = np.ascontiguousarray(tree, dtype=np.uint8) if tree is not None else None
        kv = np.ascontiguousarray(kv_vis, dtype=np.uint8) if kv_vis is not None else None
        assert tr is None or tr.shape == (n, n)
        assert kv is None or kv.size == self.cfg.seq_len
        out = np.empty((n, self.cfg.vocab_size), dtype=np.float32) if lm_head else None
        rc = self.o.L.pso_model_forward_tree(self.h, _p(tokens), n, _p(rope_pos), _p(tr), _p(kv), int(lm_head), _p(out), int(advance))
        assert rc == 0
        return out

    def kv_move(self, dst, src):
        self.o.L.pso_model_kv_move(self.h, int(dst), int(src))

    def kv_advance(self, n):
        self.o.L.pso_model_kv_advance(self.h, int(n))

    def generate(self, prompt, batch_size, steps, want_logits=False):
        prompt = _i32(prompt)
        toks = np.empty(steps, dtype=np.int32)
        lg = np.empty((steps, self.cfg.vocab_size), dtype=np.float32) if want_logits else None
        tp, td = C.c_double(0), C.c_double(0)
        rc = self.o.L.pso_model_generate(self.h, _p(prompt), prompt.size, batch_size, steps, _p(toks), _p(lg),
                                         C.byref(tp), C.byref(td))
        assert rc == 0
        return toks, lg, tp.value, td.value

    def k_cache(self, L):
        n = self.cfg.seq_len * self.cfg.kv_dim
        return np.ctypeslib.as_array(C.cast(self.o.L.pso_model_k_cache(self.h, L), C.POINTER(C.c_float)), (n,)).reshape(
            self.cfg.seq_len, self.cfg.kv_dim)

    def v_cache(self, L):
        n = self.cfg.seq_len * self.cfg.kv_dim
        return np.ctypeslib.as_array(C.cast(self.o.L.pso_model_v_cache(self.h, L), C.POINTER(C.c_float)), (n,)).reshape(
            self.cfg.kv_dim, self.cfg.seq_len)


# --------------------------------------------------------------------------- real reference
class SamplerCfg(C.Structure):
    """plain-C view of HyperParams::SamplerConfig + the vocabulary ids the chain needs (psh_sampler_cfg / ref_sampler_cfg)"""
    _fields_ = [("seed", C.c_uint64), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_uint64),
                ("penalty_last_n", C.c_int32), ("penalty_repeat", C.c_float), ("penalty_freq", C.c_float),
                ("penalty_present", C.c_float), ("penalize_nl", C.c_int32), ("ignore_eos", C.c_int32), ("n_vocabs", C.c_int32),
                ("special_eos_id", C.c_int32), ("linefeed_id", C.c_int32)]


class RefTensor(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_uint64 * 4),
                ("data", C.c_void_p)]


def ref_tensor(arr, t, ne, nb=None, type_size=None, blck=1):
    """Contiguous by default; nb in bytes (ggml convention)."""
    ne = list(ne) + [1] * (4 - len(ne))
    if nb is None:
        ts = type_size if type_size is not None else arr.itemsize
        nb = [ts, ts * (ne[0] // blck), 0, 0]
        nb[2] = nb[1] * ne[1]
        nb[3] = nb[2] * ne[2]
    rt = RefTensor(t, 0, (C.c_int64 * 4)(*ne), (C.c_uint64 * 4)(*nb), arr.ctypes.data)
    rt._keep = arr
    return rt


class Ref:
    def __init__(self, n_threads=1, so=None):
        so = so or REF_SO
        if not os.path.exists(so):
            raise FileNotFoundError(so + " (build it in the dev container: make -C oracle ref ref_fast)")
        L = self.L = C.CDLL(so)
        L.ref_row_size.restype = C.c_size_t
        L.ref_row_size.argtypes = [C.c_int, C.c_int64]
        L.ref_type_size.restype = C.c_size_t
        L.ref_blck_size.restype = C.c_int64
        L.ref_vec_dot_type.restype = C.c_int
        L.ref_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.ref_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_from_float.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_ctx_create.restype = C.c_void_p
        L.ref_ctx_destroy.argtypes = [C.c_void_p]
        L.ref_mul_mat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.ref_rope.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_softmax_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.ref_add.argtypes = [C.c_void_p] * 4
        L.ref_dup.argtypes = [C.c_void_p] * 3
        L.ref_silu_hadamard.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.ref_get_embedding.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "ref_token_tree_run"):
            L.ref_token_tree_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
        L.ref_sampler_create.restype = C.c_void_p
        L.ref_sampler_create.argtypes = [C.c_void_p]
        L.ref_sampler_free.argtypes = [C.c_void_p]
        L.ref_sampler_sample.restype = C.c_int32
        L.ref_sampler_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ref_model_create.restype = C.c_void_p
        L.ref_model_create.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        L.ref_model_destroy.argtypes = [C.c_void_p]
        L.ref_model_kv_position.restype = C.c_size_t
        L.ref_model_kv_position.argtypes = [C.c_void_p]
        L.ref_model_reset.argtypes = [C.c_void_p]
        L.ref_model_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_model_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        self.n_threads = n_threads
        self.ctx = L.ref_ctx_create(n_threads)

    def close(self):
        if getattr(self, "ctx", None):
            self.L.ref_ctx_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def row_size(self, t, k):
        return self.L.ref_row_size(t, k)

    def vec_dot_type(self, t):
        return self.L.ref_vec_dot_type(t)

    def quantize(self, t, w):
        """w: [N, K] float32 -> uint8 blocks (ggml_quantize_chunk)."""
        w = _f32(w)
        N, K = w.shape
        out = np.zeros(N * self.row_size(t, K), dtype=np.uint8)
        self.L.ref_quantize_chunk(t, _p(w), _p(out), N, K)
        return out

    def dequantize(self, t, blocks, k):
        out = np.empty(k, dtype=np.float32)
        b = np.ascontiguousarray(blocks)
        self.L.ref_dequantize_row(t, _p(b), _p(out), k)
        return out

    def from_float(self, vdt, x):
        x = _f32(x)
        out = np.zeros(self.row_size(vdt, x.size), dtype=np.uint8)
        self.L.ref_from_float(vdt, _p(x), _p(out), x.size)
        return out

    def mul_mat(self, t, w, K, N, x, want_act=False):
        x = _f32(x).reshape(-1, K)
        bs = x.shape[0]
        y = np.empty((bs, N), dtype=np.float32)
        w = np.ascontiguousarray(w)
        rs = self.row_size(t, K)
        ts, blk = self.L.ref_type_size(t), self.L.ref_blck_size(t)
        tw = ref_tensor(w, t, [K, N], nb=[ts, rs, rs * N, rs * N])
        tx = ref_tensor(x, F32, [K, bs])
        ty = ref_tensor(y, F32, [N, bs])
        vdt = self.vec_dot_type(t)
        act = np.zeros(bs * self.row_size(vdt, K), dtype=np.uint8) if (want_act and vdt != F32) else None
        rc = self.L.ref_mul_mat(self.ctx, C.byref(ty), C.byref(tw), C.byref(tx), _p(act), act.size if act is not None else 0)
        assert rc == 0
        return (y, act) if want_act else y

    def mul_mat_t(self, dst, src0, src1):
        """generic strided form with RefTensor descriptors"""
        rc = self.L.ref_mul_mat(self.ctx, C.byref(dst), C.byref(src0), C.byref(src1), None, 0)
        assert rc == 0

    def rms_norm(self, x, w, eps):
        x = _f32(x)
        w = _f32(w)
        y = np.empty_like(x)
        ne0 = x.shape[-1]
        rows = x.size // ne0
        self.L.ref_rms_norm(self.ctx, C.byref(ref_tensor(y, F32, [ne0, rows])), C.byref(ref_tensor(x, F32, [ne0, rows])),
                            C.byref(ref_tensor(w, F32, [ne0])), eps)
        return y

    def rope(self, x, pos, rp: RopeParams):
        x = _f32(x)
        pos = _i32(pos)
        y = np.empty_like(x)
        ne = [x.shape[2], x.shape[1], x.shape[0]]
        self.L.ref_rope(self.ctx, C.byref(ref_tensor(y, F32, ne)), C.byref(ref_tensor(x, F32, ne)), _p(pos), pos.size,
                        C.byref(rp))
        return y

    def softmax_ext(self, x, mask, scale):
        x = _f32(x)
        m = _f32(mask)
        y = np.empty_like(x)
        ne = [x.shape[2], x.shape[1], x.shape[0]]
        self.L.ref_softmax_ext(self.ctx, C.byref(ref_tensor(y, F32, ne)), C.byref(ref_tensor(x, F32, ne)),
                               C.byref(ref_tensor(m, F32, [m.shape[1], m.shape[0]])), scale, 0.0)
        return y

    def silu_hadamard(self, g, u):
        g = _f32(g)
        u = _f32(u)
        y = np.empty_like(g)
        self.L.ref_silu_hadamard(_p(y), _p(g), _p(u), g.size, 1)
        return y

    def add(self, a, b):
        a = _f32(a)
        b = _f32(b)
        y = np.empty_like(a)
        ne0 = a.shape[-1]
        rows = a.size // ne0
        tb = ref_tensor(b, F32, [ne0, b.size // ne0])
        self.L.ref_add(self.ctx, C.byref(ref_tensor(y, F32, [ne0, rows])), C.byref(ref_tensor(a, F32, [ne0, rows])),
                       C.byref(tb))
        return y

    def dup_t(self, dst, src):
        self.L.ref_dup(self.ctx, C.byref(dst), C.byref(src))

    def get_embedding(self, t, table, dim, vocab, tokens):
        tokens = _i32(tokens)
        out = np.empty((tokens.size, dim), dtype=np.float32)
        table = np.ascontiguousarray(table)
        rc = self.L.ref_get_embedding(t, _p(table), dim, vocab, _p(tokens), tokens.size, _p(out))
        assert rc == 0
        return out

    def model(self, gguf_path: str, arch: str, cfg: LLMConfig, n_threads=None):
        return RefModel(self, gguf_path, arch, cfg, n_threads or self.n_threads)


class RefModel:
    def __init__(self, r: Ref, path: str, arch: str, cfg: LLMConfig, n_threads: int):
        self.r, self.cfg = r, cfg
        self.h = r.L.ref_model_create(path.encode(), arch.encode(), C.byref(cfg), n_threads)
        assert self.h

    def close(self):
        if self.h:
            self.r.L.ref_model_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def position(self):
        return self.r.L.ref_model_kv_position(self.h)

    def reset(self):
        self.r.L.ref_model_reset(self.h)

    def forward(self, tokens, pos, lm_head=True):
        tokens, pos = _i32(tokens), _i32(pos)
        out = np.empty((tokens.size, self.cfg.vocab_size), dtype=np.float32) if lm_head else None
        self.r.L.ref_model_forward(self.h, _p(tokens), tokens.size, _p(pos), int(lm_head), _p(out))
        return out

    def generate(self, prompt, batch_size, steps, want_logits=False):
        prompt = _i32(prompt)
        toks = np.empty(steps, dtype=np.int32)
        lg = np.empty((steps, self.cfg.vocab_size), dtype=np.float32) if want_logits else None
        tp, td = C.c_double(0), C.c_double(0)
        self.r.L.ref_model_generate(self.h, _p(prompt), prompt.size, batch_size, steps, _p(toks), _p(lg), C.byref(tp),
                                    C.byref(td))
        return toks, lg, tp.value, td.value


# ---------------------------------------------------------------- the reference's token tree on scripted models
class SpecConfig(C.Structure):  # ref_spec_config (oracle/ref_token_tree.cpp) == psh_spec_config
    _fields_ = [("draft_batch_size", C.c_int32), ("top_k", C.c_int32), ("max_fan_out", C.c_int32), ("early_stop", C.c_int32),
                ("temperature", C.c_float), ("p_base", C.c_float), ("min_prob", C.c_float)]


class Script(C.Structure):  # ref_script: the two scripted models
    _fields_ = [("shared_seed", C.c_uint64), ("target_seed", C.c_uint64), ("draft_seed", C.c_uint64),
                ("shared_w", C.c_float), ("target_w", C.c_float), ("draft_w", C.c_float), ("vocab", C.c_int32), ("n_ctx", C.c_int32)]


def ref_token_tree_run(ref: "Ref", cfg: SpecConfig, script: Script, prefix, root_token: int, n_iterations: int):
    """src/speculative/token_tree.cpp driven as SpecTokenIterator::generate_tokens drives it.  Returns a dict of arrays:
    tokens (emitted), tree [it][bs][3] {token, position, parent}, masks [it][bs][bs], events [n][4] {model, op, a, b}."""
    bs = cfg.draft_batch_size
    prefix = np.ascontiguousarray(prefix, dtype=np.int32)
    out, n_out = np.zeros(n_iterations * bs, dtype=np.int32), np.zeros(1, dtype=np.int32)
    tree, masks = np.zeros((n_iterations, bs, 3), dtype=np.int32), np.zeros((n_iterations, bs, bs), dtype=np.uint8)
    cap = n_iterations * bs * 16
    events, n_ev = np.zeros((cap, 4), dtype=np.int32), np.zeros(1, dtype=np.int32)
    rc = ref.L.ref_token_tree_run(C.addressof(cfg), C.addressof(script), prefix.ctypes.data, prefix.size, root_token, n_iterations, out.ctypes.data,
                                  n_out.ctypes.data, tree.ctypes.data, masks.ctypes.data, events.ctypes.data, cap, n_ev.ctypes.data, None)
    assert rc == 0 and n_ev[0] <= cap
    return dict(tokens=out[:n_out[0]].copy(), tree=tree, masks=masks, events=events[:n_ev[0]].copy())


def ref_gguf_write(ref: "Ref", path: str, arch: str, name: str, alignment: int, tensors):
    """The reference's own GGUF writer (oracle/ref_gguf.cpp).  tensors: [(name, ggml type, ne tuple, uint8/float32 array)]"""
    n = len(tensors)
    names = (C.c_char_p * n)(*[t[0].encode() for t in tensors])
    types = np.array([t[1] for t in tensors], dtype=np.int32)
    ne = np.ones((n, 4), dtype=np.int64)
    for i, t in enumerate(tensors):
        ne[i, :len(t[2])] = t[2]
    keep = [np.ascontiguousarray(t[3]) for t in tensors]
    data = (C.c_void_p * n)(*[a.ctypes.data for a in keep])
    ref.L.ref_gguf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = ref.L.ref_gguf_write(path.encode(), arch.encode(), name.encode(), alignment, n, names, types.ctypes.data, ne.ctypes.data, data)
    assert rc == 0
