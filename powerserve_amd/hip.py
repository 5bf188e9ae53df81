"""ctypes binding of the C-ABI in include/ps_hip.h (powerserve_amd/lib/libps_hip.so).

This module never falls back to a CPU implementation: if the HIP library is missing or no GPU is
visible, it raises.  Tests and bench.py drive the backend through these wrappers; the C++ façade
(powerserve_amd/csrc/host) sits on the same ABI.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import gguf

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PS_HIP_LIB") or os.path.join(HERE, "lib", "libps_hip.so")  # PS_HIP_LIB: an A/B build (tools/ab_build.py)

F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K, I32 = 0, 1, 2, 8, 12, 13, 14, 15, 26
ATTN_TIMEOUT = 3  # PS_HIP_ATTN_TIMEOUT (include/ps_hip.h)
QUANT = (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K)

EXPORTS = [
    "ps_hip_abi_version", "ps_hip_build_contract", "ps_hip_device_count", "ps_hip_create", "ps_hip_destroy", "ps_hip_last_error",
    "ps_hip_device_name", "ps_hip_malloc", "ps_hip_free", "ps_hip_memcpy_h2d", "ps_hip_memcpy_d2h", "ps_hip_memset",
    "ps_hip_sync", "ps_hip_stream", "ps_hip_event_create", "ps_hip_event_record", "ps_hip_event_elapsed_ms",
    "ps_hip_event_destroy", "ps_hip_weight_upload", "ps_hip_weight_free", "ps_hip_weight_gguf_bytes",
    "ps_hip_weight_dtype", "ps_hip_vec_dot_type", "ps_hip_row_size", "ps_hip_quantize_act", "ps_hip_mul_mat",
    "ps_hip_rms_norm", "ps_hip_rope", "ps_hip_softmax_ext", "ps_hip_add", "ps_hip_dup", "ps_hip_silu_hadamard",
    "ps_hip_get_embedding", "ps_hip_get_mask", "ps_hip_argmax", "ps_hip_model_create", "ps_hip_model_destroy",
    "ps_hip_model_kv_position", "ps_hip_model_max_batch", "ps_hip_model_kv_truncate", "ps_hip_model_kv_advance", "ps_hip_model_kv_rollback", "ps_hip_model_kv_move",
    "ps_hip_model_forward", "ps_hip_model_decode_greedy", "ps_hip_model_logits", "ps_hip_model_scratch", "ps_hip_model_k_cache",
    "ps_hip_model_v_cache", "ps_hip_model_weight_bytes_per_token", "ps_hip_model_set_mode", "ps_hip_model_bench_gemv", "ps_hip_model_bench_matmul", "ps_hip_debug_timeline", "ps_hip_last_matmul_kernel", "ps_hip_debug_set", "ps_hip_debug_f16_gemm", "ps_hip_model_forward_tree", "ps_hip_model_prefill", "ps_hip_model_forward_lowered", "ps_hip_model_sync_check", "ps_hip_model_kv_mask",
    "ps_hip_soft_max", "ps_hip_model_kv_copy", "ps_hip_model_kv_save_tokens", "ps_hip_model_kv_unmask_tokens", "ps_hip_model_kv_append_tokens", "ps_hip_model_argmax",
]


class PSTensor(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("_pad", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_uint64 * 4),
                ("data", C.c_void_p)]


class RopeParams(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("n_ctx_orig", C.c_int32), ("freq_base", C.c_float),
                ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float),
                ("beta_fast", C.c_float), ("beta_slow", C.c_float), ("mode", C.c_int32)]


class LLMConfig(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("hidden_dim", C.c_uint32), ("n_layers", C.c_uint32),
                ("n_heads", C.c_uint32), ("n_kv_heads", C.c_uint32), ("seq_len", C.c_uint32),
                ("vocab_size", C.c_uint32), ("kv_dim", C.c_uint32), ("head_size", C.c_uint32),
                ("norm_eps", C.c_float), ("rope", RopeParams)]


class ModelDesc(C.Structure):
    _fields_ = [("cfg", LLMConfig), ("is_qwen2", C.c_int32), ("max_batch", C.c_int32),
                ("token_embd", C.c_void_p), ("output", C.c_void_p), ("output_norm", C.c_void_p),
                ("attn_norm", C.POINTER(C.c_void_p)), ("ffn_norm", C.POINTER(C.c_void_p)),
                ("attn_q", C.POINTER(C.c_void_p)), ("attn_k", C.POINTER(C.c_void_p)), ("attn_v", C.POINTER(C.c_void_p)),
                ("attn_output", C.POINTER(C.c_void_p)), ("ffn_gate", C.POINTER(C.c_void_p)),
                ("ffn_up", C.POINTER(C.c_void_p)), ("ffn_down", C.POINTER(C.c_void_p)),
                ("attn_q_bias", C.POINTER(C.c_void_p)), ("attn_k_bias", C.POINTER(C.c_void_p)),
                ("attn_v_bias", C.POINTER(C.c_void_p))]


def make_config(d: dict) -> LLMConfig:
    """d: llm_config of a PowerServe model.json (src/core/config.cpp:68-104; ext_factor forced to 0,
    beta_fast 32, beta_slow 0 exactly as the reference does at :97-101)."""
    r = d["rope_config"]
    rp = RopeParams(int(r["rope_dim"]), int(r["n_rope_ctx_orig"]), float(r["rope_freq_base"]),
                    float(r["rope_freq_scale"]), 0.0, float(r["rope_attn_factor"]), 32.0, 0.0, int(r["rope_type"]))
    return LLMConfig(int(d["embed_dim"]), int(d["ffn_dim"]), int(d["n_layers"]), int(d["n_attn_heads"]),
                     int(d["n_attn_kv_heads"]), int(d["n_ctx"]), int(d["vocab_size"]), int(d["kv_dim"]),
                     int(d["head_size"]), float(d["norm_eps"]), rp)


_LIB = None


def lib() -> C.CDLL:
    """Load libps_hip.so (raises if it has not been built: python -m powerserve_amd.build)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: the HIP backend is not built (run __graft_entry__.build()); "
                           "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    T = C.POINTER(PSTensor)
    sig = {
        "ps_hip_create": (i32, [i32, C.POINTER(vp)]), "ps_hip_destroy": (None, [vp]),
        "ps_hip_last_error": (C.c_char_p, [vp]), "ps_hip_device_name": (i32, [vp, C.c_char_p, sz]),
        "ps_hip_malloc": (i32, [vp, sz, C.POINTER(vp)]), "ps_hip_free": (i32, [vp, vp]),
        "ps_hip_memcpy_h2d": (i32, [vp, vp, vp, sz]), "ps_hip_memcpy_d2h": (i32, [vp, vp, vp, sz]),
        "ps_hip_memset": (i32, [vp, vp, i32, sz]), "ps_hip_sync": (i32, [vp]), "ps_hip_stream": (vp, [vp]),
        "ps_hip_event_create": (i32, [vp, C.POINTER(vp)]), "ps_hip_event_record": (i32, [vp, vp]),
        "ps_hip_event_elapsed_ms": (i32, [vp, vp, vp, C.POINTER(C.c_float)]), "ps_hip_event_destroy": (i32, [vp, vp]),
        "ps_hip_weight_upload": (i32, [vp, i32, vp, i64, i64, C.POINTER(vp)]), "ps_hip_weight_free": (None, [vp, vp]),
        "ps_hip_weight_gguf_bytes": (C.c_uint64, [vp]), "ps_hip_weight_dtype": (i32, [vp]),
        "ps_hip_vec_dot_type": (i32, [i32]), "ps_hip_row_size": (sz, [i32, i64]),
        "ps_hip_quantize_act": (i32, [vp, i32, vp, i64, i64, vp]),
        "ps_hip_mul_mat": (i32, [vp, T, T, T]), "ps_hip_rms_norm": (i32, [vp, T, T, T, C.c_float]),
        "ps_hip_rope": (i32, [vp, T, T, vp, i32, C.POINTER(RopeParams)]),
        "ps_hip_softmax_ext": (i32, [vp, T, T, T, C.c_float, C.c_float]), "ps_hip_add": (i32, [vp, T, T, T]),
        "ps_hip_dup": (i32, [vp, T, T]), "ps_hip_silu_hadamard": (i32, [vp, T, T, T]),
        "ps_hip_get_embedding": (i32, [vp, T, T, vp, i32]), "ps_hip_get_mask": (i32, [vp, T, vp, i32, vp]),
        "ps_hip_argmax": (i32, [vp, vp, i64, i64, vp]),
        "ps_hip_model_create": (i32, [vp, C.POINTER(ModelDesc), C.POINTER(vp)]), "ps_hip_model_destroy": (None, [vp]),
        "ps_hip_model_kv_position": (sz, [vp]), "ps_hip_model_kv_truncate": (i32, [vp, sz]), "ps_hip_model_kv_advance": (i32, [vp, sz]),
        "ps_hip_model_kv_rollback": (i32, [vp, sz]), "ps_hip_model_kv_move": (i32, [vp, sz, sz]),
        "ps_hip_model_forward": (i32, [vp, vp, i32, vp, vp, i32, vp]),
        "ps_hip_model_forward_lowered": (i32, [vp, vp, i32, vp, vp, i32]),
        "ps_hip_model_forward_tree": (i32, [vp, vp, i32, vp, vp, i32, vp, i32]), "ps_hip_model_prefill": (i32, [vp, vp, i32, i32]), "ps_hip_model_kv_mask": (i32, [vp, sz, i32]),
        "ps_hip_model_decode_greedy": (i32, [vp, i32, i32, vp]), "ps_hip_model_logits": (vp, [vp]), "ps_hip_model_scratch": (vp, [vp, i32]),
        "ps_hip_model_k_cache": (vp, [vp, i32]), "ps_hip_model_v_cache": (vp, [vp, i32]),
        "ps_hip_model_weight_bytes_per_token": (C.c_uint64, [vp]), "ps_hip_model_set_mode": (i32, [vp, i32]),
        "ps_hip_model_sync_check": (i32, [vp]), "ps_hip_soft_max": (i32, [vp, T, T]),
        "ps_hip_model_kv_copy": (i32, [vp, sz, sz]), "ps_hip_model_kv_save_tokens": (i32, [vp, sz]), "ps_hip_model_kv_unmask_tokens": (i32, [vp, sz]),
        "ps_hip_model_kv_append_tokens": (i32, [vp, sz, C.POINTER(sz)]), "ps_hip_model_argmax": (i32, [vp, i32, vp]),
        "ps_hip_debug_timeline": (i32, [vp, i32, vp, i32]),
        "ps_hip_last_matmul_kernel": (C.c_char_p, []),
        "ps_hip_debug_set": (i32, [i32, i32]),
        "ps_hip_debug_f16_gemm": (i32, [vp, i32, C.c_int64, C.c_int64, i32, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


class PSHipError(RuntimeError):
    pass


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Ctx:
    def __init__(self, device: int = 0):
        self.L = lib()
        if self.L.ps_hip_device_count() <= 0:
            raise PSHipError("no HIP device visible: the MI355X backend cannot run (there is no CPU fallback)")
        h = C.c_void_p()
        if self.L.ps_hip_create(device, C.byref(h)) != 0:
            raise PSHipError(f"ps_hip_create({device}) failed")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.ps_hip_destroy(self.h)
            self.h = None

    def check(self, rc: int):
        if rc != 0:
            raise PSHipError(self.L.ps_hip_last_error(self.h).decode())

    def name(self) -> str:
        b = C.create_string_buffer(256)
        self.check(self.L.ps_hip_device_name(self.h, b, 256))
        return b.value.decode()

    def sync(self):
        self.check(self.L.ps_hip_sync(self.h))

    # ---- memory
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(self.L.ps_hip_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, p: int):
        self.check(self.L.ps_hip_free(self.h, p))

    def to_device(self, a: np.ndarray) -> "DevArray":
        a = np.ascontiguousarray(a)
        d = DevArray(self, a.shape, a.dtype)
        self.check(self.L.ps_hip_memcpy_h2d(self.h, d.ptr, _ptr(a), a.nbytes))
        return d

    def empty(self, shape, dtype=np.float32) -> "DevArray":
        return DevArray(self, shape, dtype)

    # ---- events
    def event(self) -> int:
        e = C.c_void_p()
        self.check(self.L.ps_hip_event_create(self.h, C.byref(e)))
        return e.value

    def record(self, ev: int):
        self.check(self.L.ps_hip_event_record(self.h, ev))

    def elapsed_ms(self, a: int, b: int) -> float:
        ms = C.c_float()
        self.check(self.L.ps_hip_event_elapsed_ms(self.h, a, b, C.byref(ms)))
        return ms.value

    # ---- weights
    def upload_weight(self, dtype: int, blocks: np.ndarray, K: int, N: int) -> "Weight":
        blocks = np.ascontiguousarray(blocks)
        w = C.c_void_p()
        self.check(self.L.ps_hip_weight_upload(self.h, dtype, _ptr(blocks), K, N, C.byref(w)))
        return Weight(self, w.value, dtype, K, N)


class DevArray:
    """Row-major device array; shape is numpy-style (slowest first)."""

    def __init__(self, ctx: Ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(int(s) for s in np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx.malloc(max(self.nbytes, 16))

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(out), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = None

    def tensor(self, ne=None, nb=None, dtype=F32) -> PSTensor:
        """ps_tensor view; default: contiguous with ne = reversed numpy shape."""
        if ne is None:
            ne = list(reversed(self.shape))
        ne = list(ne) + [1] * (4 - len(ne))
        if nb is None:
            nb = [self.dtype.itemsize]
            for i in range(3):
                nb.append(nb[-1] * ne[i])
        return PSTensor(dtype, 0, (C.c_int64 * 4)(*ne), (C.c_uint64 * 4)(*nb), self.ptr)


class Weight:
    def __init__(self, ctx: Ctx, h: int, dtype: int, K: int, N: int):
        self.ctx, self.h, self.dtype, self.K, self.N = ctx, h, dtype, K, N

    @property
    def gguf_bytes(self) -> int:
        return self.ctx.L.ps_hip_weight_gguf_bytes(self.h)

    def tensor(self) -> PSTensor:
        rs = self.ctx.L.ps_hip_row_size(self.dtype, self.K)
        ts = {F32: 4, Q4_0: 18, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210}[self.dtype]
        return PSTensor(self.dtype, 0, (C.c_int64 * 4)(self.K, self.N, 1, 1), (C.c_uint64 * 4)(ts, rs, rs * self.N, rs * self.N),
                        self.h)

    def free(self):
        if self.h:
            self.ctx.L.ps_hip_weight_free(self.ctx.h, self.h)
            self.h = None


class Model:
    """ps_hip_model over a PowerServe model directory (model.json + ggml/weights.gguf)."""

    def __init__(self, ctx: Ctx, model_dir: str, max_batch: int = 128, n_ctx: int | None = None):
        self.ctx = ctx
        with open(os.path.join(model_dir, "model.json")) as f:
            mj = json.load(f)
        self.arch = mj["model_arch"]
        llm = dict(mj["llm_config"])
        if n_ctx is not None:
            llm["n_ctx"] = n_ctx  # explicit cap: the FP32 KV cache is sized by n_ctx
        self.cfg = make_config(llm)
        rd = gguf.GGUFReader(os.path.join(model_dir, "ggml", "weights.gguf"))
        self.weights, self.f32 = {}, {}
        L = self.cfg.n_layers

        def up(name):
            ti = rd.tensors[name]
            if ti.type == F32 and len(ti.ne) == 1:
                d = ctx.to_device(np.array(rd.data(name), dtype=np.float32))
                self.f32[name] = d
                return d.ptr
            w = ctx.upload_weight(ti.type, rd.data(name), ti.ne[0], ti.ne[1])
            self.weights[name] = w
            return w.h

        def arr(fmt):
            return (C.c_void_p * L)(*[up(fmt.format(i)) for i in range(L)])

        d = ModelDesc()
        d.cfg, d.is_qwen2, d.max_batch = self.cfg, int(self.arch == "qwen2"), max_batch
        d.token_embd = up("token_embd.weight")
        d.output = up("output.weight") if "output.weight" in rd.tensors else None
        d.output_norm = up("output_norm.weight")
        self._keep = []
        for field, fmt in (("attn_norm", "blk.{}.attn_norm.weight"), ("ffn_norm", "blk.{}.ffn_norm.weight"),
                           ("attn_q", "blk.{}.attn_q.weight"), ("attn_k", "blk.{}.attn_k.weight"),
                           ("attn_v", "blk.{}.attn_v.weight"), ("attn_output", "blk.{}.attn_output.weight"),
                           ("ffn_gate", "blk.{}.ffn_gate.weight"), ("ffn_up", "blk.{}.ffn_up.weight"),
                           ("ffn_down", "blk.{}.ffn_down.weight")):
            a = arr(fmt)
            self._keep.append(a)
            setattr(d, field, a)
        if self.arch == "qwen2":
            for field, fmt in (("attn_q_bias", "blk.{}.attn_q.bias"), ("attn_k_bias", "blk.{}.attn_k.bias"),
                               ("attn_v_bias", "blk.{}.attn_v.bias")):
                a = arr(fmt)
                self._keep.append(a)
                setattr(d, field, a)
        h = C.c_void_p()
        ctx.check(ctx.L.ps_hip_model_create(ctx.h, C.byref(d), C.byref(h)))
        self.h, self.max_batch = h, max_batch

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.ps_hip_model_destroy(self.h)
            self.h = None
            for w in self.weights.values():
                w.free()
            for a in self.f32.values():
                a.free()

    @property
    def position(self) -> int:
        return self.ctx.L.ps_hip_model_kv_position(self.h)

    def reset(self):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_truncate(self.h, 0))

    def set_mode(self, mode: int):
        self.ctx.L.ps_hip_model_set_mode(self.h, mode)

    @property
    def weight_bytes_per_token(self) -> int:
        return self.ctx.L.ps_hip_model_weight_bytes_per_token(self.h)

    def forward(self, tokens, pos, lm_head=True, tree=None, want_logits=True):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        n = tokens.size
        am = np.empty(n, dtype=np.int32)
        tr = np.ascontiguousarray(tree, dtype=np.uint8) if tree is not None else None
        self.ctx.check(self.ctx.L.ps_hip_model_forward(self.h, _ptr(tokens), n, _ptr(pos), _ptr(tr) if tr is not None else None,
                                                       int(lm_head), _ptr(am)))
        if not lm_head:
            return None, None
        logits = None
        if want_logits:
            logits = np.empty((n, self.cfg.vocab_size), dtype=np.float32)
            self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(logits), self.ctx.L.ps_hip_model_logits(self.h), logits.nbytes))
        return logits, am

    def prefill(self, tokens, chunk: int):
        """ModelTokenIterator's prefill: the tokens appended at the cache position in reference chunks of `chunk`, no logits
        (ps_hip_model_prefill: same bits as forward() per chunk, several chunks per launch sequence when max_batch allows)."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        self.ctx.check(self.ctx.L.ps_hip_model_prefill(self.h, _ptr(tokens), tokens.size, int(chunk)))

    def forward_tree(self, tokens, rope_pos, tree=None, lm_head=True, want_logits=False, advance=False):
        """Token-tree forward (src/speculative/token_tree.cpp): tokens appended at the current KV position, column i rotated
        with rope_pos[i], visible batch columns given by tree[i][j] (None: causal).  Returns (logits | None, argmax)."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        rp = np.ascontiguousarray(rope_pos, dtype=np.int32)
        n = tokens.size
        am = np.empty(n, dtype=np.int32)
        tr = np.ascontiguousarray(tree, dtype=np.uint8) if tree is not None else None
        self.ctx.check(self.ctx.L.ps_hip_model_forward_tree(self.h, _ptr(tokens), n, _ptr(rp), _ptr(tr) if tr is not None else None,
                                                            int(lm_head), _ptr(am), int(advance)))
        logits = None
        if lm_head and want_logits:
            logits = np.empty((n, self.cfg.vocab_size), dtype=np.float32)
            self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(logits), self.ctx.L.ps_hip_model_logits(self.h), logits.nbytes))
        return logits, am

    def kv_mask(self, index: int, visible: bool):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_mask(self.h, int(index), int(bool(visible))))

    def kv_move(self, dst: int, src: int):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_move(self.h, int(dst), int(src)))

    def kv_advance(self, n: int):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_advance(self.h, int(n)))

    def kv_rollback(self, n: int):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_rollback(self.h, int(n)))

    # the rest of KVCacheInterface (core/kv_cache.hpp:120-162)
    def kv_copy(self, dst_cache_index: int, src_token_index: int):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_copy(self.h, int(dst_cache_index), int(src_token_index)))

    def kv_save_tokens(self, n: int):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_save_tokens(self.h, int(n)))

    def kv_unmask_tokens(self, n: int):
        self.ctx.check(self.ctx.L.ps_hip_model_kv_unmask_tokens(self.h, int(n)))

    def kv_append_tokens(self, n: int) -> int:
        old = C.c_size_t()
        self.ctx.check(self.ctx.L.ps_hip_model_kv_append_tokens(self.h, int(n), C.byref(old)))
        return old.value

    def forward_lowered(self, tokens, pos, lm_head=True):
        """Enqueue only (what HIPBackend::run_lowered issues); the result counts after sync_check() / kv_advance()."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.ctx.check(self.ctx.L.ps_hip_model_forward_lowered(self.h, _ptr(tokens), tokens.size, _ptr(pos), None, int(lm_head)))

    def sync_check(self) -> int:
        """0, or ATTN_TIMEOUT (the pending lowered forward has no valid result: run it again)."""
        return self.ctx.L.ps_hip_model_sync_check(self.h)

    def logits(self, n: int = 1) -> np.ndarray:
        out = np.empty((n, self.cfg.vocab_size), dtype=np.float32)
        self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(out), self.ctx.L.ps_hip_model_logits(self.h), out.nbytes))
        return out

    def decode_greedy(self, token: int, steps: int) -> np.ndarray:
        out = np.empty(steps, dtype=np.int32)
        self.ctx.check(self.ctx.L.ps_hip_model_decode_greedy(self.h, int(token), steps, _ptr(out)))
        return out

    def generate(self, prompt, batch_size: int, steps: int):
        """ModelTokenIterator semantics (src/model/model.hpp:117-184): prefill all but the last prompt token in
        chunks of batch_size without lm_head, then `steps` greedy single-token steps."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        self.reset()
        done = 0
        while done < prompt.size - 1:
            bs = min(batch_size, prompt.size - 1 - done)
            self.forward(prompt[done:done + bs], np.arange(self.position, self.position + bs), lm_head=False)
            done += bs
        return self.decode_greedy(int(prompt[-1]), steps)

    def scratch(self, which: int, rows: int) -> np.ndarray:
        """Diagnostics: last-layer scratch tensor of the most recent forward (0 x, 1 q, 2 att, 3 ffn hidden, 4 scores)."""
        width = {0: self.cfg.dim, 1: self.cfg.dim, 2: self.cfg.dim, 3: self.cfg.hidden_dim, 4: self.cfg.n_heads * self.cfg.seq_len}[which]
        out = np.empty((rows, width), dtype=np.float32)
        self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(out), self.ctx.L.ps_hip_model_scratch(self.h, which), out.nbytes))
        return out

    def k_cache(self, layer: int) -> np.ndarray:
        out = np.empty((self.cfg.seq_len, self.cfg.kv_dim), dtype=np.float32)
        self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(out), self.ctx.L.ps_hip_model_k_cache(self.h, layer), out.nbytes))
        return out

    def v_cache(self, layer: int) -> np.ndarray:
        out = np.empty((self.cfg.kv_dim, self.cfg.seq_len), dtype=np.float32)
        self.ctx.check(self.ctx.L.ps_hip_memcpy_d2h(self.ctx.h, _ptr(out), self.ctx.L.ps_hip_model_v_cache(self.h, layer), out.nbytes))
        return out
