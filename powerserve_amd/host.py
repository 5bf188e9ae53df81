"""ctypes driver for the C++ host facade (powerserve_amd/lib/libps_host.so): the Graph -> Executor -> HIPBackend
mirror of the reference's host side, sitting on the C-ABI of include/ps_hip.h."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libps_host.so")
EXPORTS = ["psh_last_error", "psh_model_load", "psh_model_free", "psh_model_set_fused", "psh_model_plan_stats", "psh_model_kv_position", "psh_model_reset",
           "psh_model_vocab", "psh_model_forward", "psh_model_generate", "psh_spec_generate", "psh_spec_generate_sampled", "psh_draft_sample", "psh_sampler_create", "psh_sampler_free", "psh_sampler_sample",
           "psh_model_generate_sampled", "psh_token_tree_run", "psh_gguf_summary", "psh_config_summary",
           "psh_model_decode", "psh_model_prefill", "psh_model_plan_cache_hits", "psh_model_set_plan_cache", "psh_graph_softmax", "psh_backend_get_n_tasks", "psh_backend_add_cache", "psh_model_kv_read", "psh_kv_op"]
_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.psh_last_error.restype = C.c_char_p
        L.psh_model_load.restype = C.c_void_p
        L.psh_model_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.psh_model_free.argtypes = [C.c_void_p]
        L.psh_model_set_fused.argtypes = [C.c_void_p, C.c_int]
        L.psh_model_plan_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.psh_model_kv_position.restype = C.c_size_t
        L.psh_model_kv_position.argtypes = [C.c_void_p]
        L.psh_model_reset.argtypes = [C.c_void_p]
        L.psh_model_vocab.restype = C.c_uint32
        L.psh_model_vocab.argtypes = [C.c_void_p]
        L.psh_model_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.psh_model_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.psh_spec_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.psh_spec_generate_sampled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.psh_draft_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.psh_sampler_create.restype = C.c_void_p
        L.psh_sampler_create.argtypes = [C.c_void_p]
        L.psh_sampler_free.argtypes = [C.c_void_p]
        L.psh_sampler_sample.restype = C.c_int32
        L.psh_sampler_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.psh_model_generate_sampled.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.psh_token_tree_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int] + [C.c_void_p] * 5
        L.psh_model_plan_cache_hits.argtypes = [C.c_void_p]
        L.psh_model_set_plan_cache.argtypes = [C.c_void_p, C.c_int]
        L.psh_model_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.psh_model_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.psh_graph_softmax.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.psh_backend_get_n_tasks.argtypes = [C.c_void_p]
        L.psh_backend_add_cache.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.psh_model_kv_read.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.psh_kv_op.restype = C.c_int64
        L.psh_kv_op.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64]
        L.psh_config_summary.restype = C.c_int64
        L.psh_config_summary.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.psh_gguf_summary.restype = C.c_int64
        L.psh_gguf_summary.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        _LIB = L
    return _LIB


class SamplerCfg(C.Structure):
    """HyperParams::SamplerConfig (src/core/config.hpp:34-47) + the vocabulary ids SamplerChain::build_from_config takes
    from the tokenizer (n_vocabs, special_eos_id, linefeed_id; -1 = none)."""
    _fields_ = [("seed", C.c_uint64), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_uint64),
                ("penalty_last_n", C.c_int32), ("penalty_repeat", C.c_float), ("penalty_freq", C.c_float),
                ("penalty_present", C.c_float), ("penalize_nl", C.c_int32), ("ignore_eos", C.c_int32), ("n_vocabs", C.c_int32),
                ("special_eos_id", C.c_int32), ("linefeed_id", C.c_int32)]

    @classmethod
    def make(cls, n_vocabs, seed=0, temperature=0.8, top_p=0.95, top_k=40, penalty_last_n=64, penalty_repeat=1.0, penalty_freq=0.0,
             penalty_present=0.0, penalize_nl=False, ignore_eos=False, special_eos_id=-1, linefeed_id=-1):
        return cls(seed, temperature, top_p, top_k, penalty_last_n, penalty_repeat, penalty_freq, penalty_present, int(penalize_nl),
                   int(ignore_eos), n_vocabs, special_eos_id, linefeed_id)


class HostError(RuntimeError):
    pass


class HostModel:
    """load_model(dir) + Model::forward / generate through the C++ facade."""

    def __init__(self, model_dir: str, device: int = 0, max_batch: int = 128, n_ctx: int = 0):
        self.L = lib()
        self.h = self.L.psh_model_load(model_dir.encode(), device, max_batch, n_ctx)
        if not self.h:
            raise HostError(self.L.psh_last_error().decode())
        self.vocab = self.L.psh_model_vocab(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.psh_model_free(self.h)
            self.h = None

    def plan_stats(self):
        """(graphs handed to HIPBackend::plan, graphs it lowered to the fused launch plan)"""
        a, b = C.c_int(), C.c_int()
        self.L.psh_model_plan_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def plan_cache_hits(self) -> int:
        """forwards that ran a lowered launch sequence without building a second graph (the plan cache keyed on (batch size, lm_head))"""
        return self.L.psh_model_plan_cache_hits(self.h)

    def set_plan_cache(self, on: bool):
        self.L.psh_model_set_plan_cache(self.h, int(on))

    def set_fused(self, fused: bool):
        """True: fused kernels + hipGraph (default).  False: op-by-op Graph/Executor path."""
        self.L.psh_model_set_fused(self.h, int(fused))

    @property
    def position(self) -> int:
        return self.L.psh_model_kv_position(self.h)

    def reset(self):
        self.L.psh_model_reset(self.h)

    def forward(self, tokens, pos, lm_head=True):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        p = np.ascontiguousarray(pos, dtype=np.int32)
        out = np.empty((t.size, self.vocab), dtype=np.float32) if lm_head else None
        rc = self.L.psh_model_forward(self.h, t.ctypes.data, t.size, p.ctypes.data, int(lm_head), out.ctypes.data if lm_head else None)
        if rc:
            raise HostError(self.L.psh_last_error().decode())
        return out

    def decode(self, tokens, pos):
        """Model::decode (greedy): ids only -- a lowered graph hands back the device arg-max, 4 bytes per token."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        p = np.ascontiguousarray(pos, dtype=np.int32)
        out = np.empty(t.size, dtype=np.int32)
        if self.L.psh_model_decode(self.h, t.ctypes.data, t.size, p.ctypes.data, out.ctypes.data):
            raise HostError(self.L.psh_last_error().decode())
        return out

    def prefill(self, tokens, batch_size: int):
        """ModelTokenIterator's prefill loop (chunks of batch_size, no logits) for tokens appended at the cache position."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        if self.L.psh_model_prefill(self.h, t.ctypes.data, t.size, int(batch_size)):
            raise HostError(self.L.psh_last_error().decode())

    # ---- boundary members no model graph uses (src/graph/graph.cpp:118, ggml.hpp:227,233, core/kv_cache.hpp:120-162)
    def graph_softmax(self, x):
        """Graph::softmax(x) through Executor::run (the SOFTMAX op -> HIPBackend::softmax -> ps_hip_soft_max); x [rows, n]."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        if self.L.psh_graph_softmax(self.h, x.ctypes.data, x.shape[1], x.shape[0], out.ctypes.data):
            raise HostError(self.L.psh_last_error().decode())
        return out

    def get_n_tasks(self) -> int:
        return self.L.psh_backend_get_n_tasks(self.h)

    def add_cache(self, layer: int, k, v):
        """GGMLBackend::add_cache (deprecated in the reference): the batch's K / V rows [bs, kv_dim] behind the cache position."""
        k = np.ascontiguousarray(k, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        if self.L.psh_backend_add_cache(self.h, layer, k.ctypes.data, v.ctypes.data, k.shape[0]):
            raise HostError(self.L.psh_last_error().decode())

    def kv_read(self, layer: int, slot: int, kv_dim: int):
        k, v = np.empty(kv_dim, np.float32), np.empty(kv_dim, np.float32)
        if self.L.psh_model_kv_read(self.h, layer, slot, k.ctypes.data, v.ctypes.data):
            raise HostError(self.L.psh_last_error().decode())
        return k, v

    KV_OPS = {"copy": 0, "move": 1, "mask": 2, "unmask": 3, "save_tokens": 4, "unmask_tokens": 5, "advance_tokens": 6, "rollback_tokens": 7,
              "truncate_tokens": 8, "append_tokens": 9}

    def kv(self, op: str, a: int = 0, b: int = 0) -> int:
        """One KVCacheInterface member on the device cache; returns what the member returns (old position for the *_tokens movers)."""
        r = self.L.psh_kv_op(self.h, self.KV_OPS[op], int(a), int(b))
        if r < 0:
            raise HostError(self.L.psh_last_error().decode())
        return int(r)

    def generate(self, prompt, batch_size: int, steps: int):
        p = np.ascontiguousarray(prompt, dtype=np.int32)
        out = np.empty(steps, dtype=np.int32)
        if self.L.psh_model_generate(self.h, p.ctypes.data, p.size, batch_size, steps, out.ctypes.data):
            raise HostError(self.L.psh_last_error().decode())
        return out


def config_summary(work_folder: str) -> dict:
    """Config(work_folder, work_folder/workspace.json) of the C++ host (csrc/host/json_gguf.cpp): hparams incl. the sampler
    section, main / draft model directories — as a dict of strings."""
    L = lib()
    need = L.psh_config_summary(work_folder.encode(), None, 0)
    if need < 0:
        raise HostError(L.psh_last_error().decode())
    buf = C.create_string_buffer(need)
    L.psh_config_summary(work_folder.encode(), buf, need)
    return dict(line.split("=", 1) for line in buf.value.decode().splitlines())


def gguf_summary(path: str):
    """What the C++ GGUF reader (csrc/host/json_gguf.cpp) sees in a file: (tensors, strings, numbers) with
    tensors = [(name, type, nbytes, fnv1a-64 hex of the data, ne tuple)] in file order."""
    L = lib()
    need = L.psh_gguf_summary(path.encode(), None, 0)
    if need < 0:
        raise HostError(L.psh_last_error().decode())
    buf = C.create_string_buffer(need)
    L.psh_gguf_summary(path.encode(), buf, need)
    tensors, strings, numbers = [], {}, {}
    for line in buf.value.decode().splitlines():
        kind, rest = line[0], line[2:]
        if kind == "T":
            name, t, nb, h, *ne = rest.split(" ")
            tensors.append((name, int(t), int(nb), h, tuple(int(x) for x in ne)))
        else:
            k, _, v = rest.partition(" ")
            if kind == "S":
                strings[k] = v
            else:
                numbers[k] = float(v)
    return tensors, strings, numbers


def draft_sample(logits, top_k: int = 15, temperature: float = 1.5):
    """TopK -> Temperature -> Softmax of the C++ mirror (csrc/host/speculative.cpp): (tokens, probs) sorted by probability."""
    L = lib()
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    k = min(top_k, lg.size)
    toks, probs = np.empty(k, dtype=np.int32), np.empty(k, dtype=np.float32)
    n = L.psh_draft_sample(lg.ctypes.data, lg.size, top_k, temperature, toks.ctypes.data, probs.ctypes.data)
    if n < 0:
        raise HostError(L.psh_last_error().decode())
    return toks[:n], probs[:n]


def spec_generate(target: HostModel, draft: HostModel, prompt, batch_size: int, steps: int, draft_batch_size: int = 12):
    """SpeculativeModel::generate of the C++ mirror: (token ids, stats dict)."""
    L = lib()
    p = np.ascontiguousarray(prompt, dtype=np.int32)
    out, st = np.empty(steps, dtype=np.int32), np.zeros(5, dtype=np.uint64)
    if L.psh_spec_generate(target.h, draft.h, p.ctypes.data, p.size, batch_size, steps, draft_batch_size, out.ctypes.data, st.ctypes.data):
        raise HostError(L.psh_last_error().decode())
    keys = ("n_draft_times", "n_draft_tokens", "n_accepted_tokens", "n_iterations", "n_generated_tokens")
    return out, {k: int(v) for k, v in zip(keys, st)}


def spec_generate_sampled(target: HostModel, draft: HostModel, prompt, batch_size: int, steps: int, sampler: "Sampler", eos: int = -1, draft_batch_size: int = 12):
    """SpeculativeModel::generate with the verify going through a sampler chain (sampler.apply + greedy pick per tree node,
    src/speculative/token_tree.cpp:214-216; the chain's accept() is never called, as in the reference) and an optional stop token.  Returns (ids, stats); ids may
    be shorter than `steps` (stop token emitted, or no room left in the caches for another tree)."""
    L = lib()
    p = np.ascontiguousarray(prompt, dtype=np.int32)
    out, n_out, st = np.empty(max(steps, 1), dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(5, dtype=np.uint64)
    if L.psh_spec_generate_sampled(target.h, draft.h, p.ctypes.data, p.size, batch_size, steps, draft_batch_size, sampler.h, eos, out.ctypes.data, n_out.ctypes.data, st.ctypes.data):
        raise HostError(L.psh_last_error().decode())
    keys = ("n_draft_times", "n_draft_tokens", "n_accepted_tokens", "n_iterations", "n_generated_tokens")
    return out[:int(n_out[0])].copy(), {k: int(v) for k, v in zip(keys, st)}


class SpecConfig(C.Structure):
    """psh_spec_config: plain-C view of SpeculativeConfig (csrc/host/speculative.hpp); defaults = the reference's."""
    _fields_ = [("draft_batch_size", C.c_int32), ("top_k", C.c_int32), ("max_fan_out", C.c_int32), ("early_stop", C.c_int32),
                ("temperature", C.c_float), ("p_base", C.c_float), ("min_prob", C.c_float)]

    @classmethod
    def make(cls, draft_batch_size=12, top_k=15, max_fan_out=3, early_stop=True, temperature=1.5, p_base=0.9, min_prob=0.2):
        return cls(draft_batch_size, top_k, max_fan_out, int(early_stop), temperature, p_base, min_prob)


class SpecBackendCallbacks(C.Structure):
    """psh_spec_backend: the seven calls the token tree makes on a model, as C function pointers."""
    KV_POSITION = C.CFUNCTYPE(C.c_int64, C.c_void_p)
    FORWARD_ONE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float))
    FORWARD_TREE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_int32))
    KV_MASK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int32)
    KV_MOVE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64)
    KV_N = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64)
    _fields_ = [("user", C.c_void_p), ("kv_position", KV_POSITION), ("forward_one", FORWARD_ONE), ("forward_tree", FORWARD_TREE),
                ("kv_mask", KV_MASK), ("kv_move", KV_MOVE), ("kv_advance", KV_N), ("kv_rollback", KV_N), ("vocab_size", C.c_int32)]

    @classmethod
    def wrap(cls, model):
        """`model`: any object with kv_position(), forward_one(token, position, want_logits) -> logits | None,
        forward_tree(tokens, positions, mask[n][n]) -> argmax[n], kv_mask(slot, visible), kv_move(dst, src), kv_advance(n),
        kv_rollback(n) and a vocab_size attribute.  An exception in a callback is reported as a failed call."""
        def guard(fn, bad=1):
            def run(*a):
                try:
                    r = fn(*a)
                    return 0 if r is None else r
                except Exception:  # noqa: BLE001 — must not propagate through the C frame
                    import traceback
                    traceback.print_exc()
                    return bad
            return run

        def fwd1(_, tok, pos, out):
            lg = model.forward_one(int(tok), int(pos), bool(out))
            if out:
                C.memmove(out, np.ascontiguousarray(lg, dtype=np.float32).ctypes.data, model.vocab_size * 4)

        def fwdt(_, toks, n, poss, mask, am):
            t, p = np.ctypeslib.as_array(toks, (n,)).copy(), np.ctypeslib.as_array(poss, (n,)).copy()
            mk = np.ctypeslib.as_array(mask, (n, n)).copy()
            np.ctypeslib.as_array(am, (n,))[:] = np.asarray(model.forward_tree(t, p, mk), dtype=np.int32)

        cb = cls(None, cls.KV_POSITION(guard(lambda _: int(model.kv_position()), -1)), cls.FORWARD_ONE(guard(fwd1)), cls.FORWARD_TREE(guard(fwdt)),
                 cls.KV_MASK(guard(lambda _, s, v: model.kv_mask(int(s), bool(v)))), cls.KV_MOVE(guard(lambda _, d, s: model.kv_move(int(d), int(s)))),
                 cls.KV_N(guard(lambda _, n: model.kv_advance(int(n)))), cls.KV_N(guard(lambda _, n: model.kv_rollback(int(n)))), int(model.vocab_size))
        return cb


def token_tree_run(target, draft, cfg: SpecConfig, root_token: int, n_iterations: int):
    """TokenTree (csrc/host/speculative.cpp) over two caller-supplied models (see SpecBackendCallbacks.wrap):
    n_iterations rounds of draft / tree forward / verify.  Returns (emitted tokens, trees, stats) where trees[it] is an
    int32 array [n_nodes][6] of {token, position, parent, cache_index, accepted, depth}."""
    L = lib()
    tcb, dcb = SpecBackendCallbacks.wrap(target), SpecBackendCallbacks.wrap(draft)
    bs = cfg.draft_batch_size
    out, n_out = np.zeros(n_iterations * bs, dtype=np.int32), np.zeros(1, dtype=np.int32)
    tree, n_nodes, st = np.zeros((n_iterations, bs, 6), dtype=np.int32), np.zeros(n_iterations, dtype=np.int32), np.zeros(5, dtype=np.uint64)
    if L.psh_token_tree_run(C.addressof(tcb), C.addressof(dcb), C.addressof(cfg), root_token, n_iterations, out.ctypes.data, n_out.ctypes.data,
                            tree.ctypes.data, n_nodes.ctypes.data, st.ctypes.data):
        raise HostError(L.psh_last_error().decode())
    keys = ("n_draft_times", "n_draft_tokens", "n_accepted_tokens", "n_iterations", "n_generated_tokens")
    return out[:n_out[0]].copy(), [tree[i, :n_nodes[i]].copy() for i in range(n_iterations)], {k: int(v) for k, v in zip(keys, st)}


class Sampler:
    """SamplerChain of the C++ mirror (csrc/host/sampler.cpp): sample(logits) = apply + probs[0] + accept."""

    def __init__(self, cfg: SamplerCfg):
        self.L = lib()
        self.cfg = cfg
        self.h = self.L.psh_sampler_create(C.byref(cfg))
        if not self.h:
            raise HostError(self.L.psh_last_error().decode())

    def sample(self, logits) -> int:
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        t = self.L.psh_sampler_sample(self.h, lg.ctypes.data, lg.size)
        if t < 0:
            raise HostError(self.L.psh_last_error().decode())
        return int(t)

    def close(self):
        if getattr(self, "h", None):
            self.L.psh_sampler_free(self.h)
            self.h = None


def generate_sampled(model: HostModel, prompt, batch_size: int, steps: int, cfg: SamplerCfg):
    """Model::generate with a sampler chain (host loop: forward, logits to the host, sample, accept)."""
    L = lib()
    p = np.ascontiguousarray(prompt, dtype=np.int32)
    out = np.empty(steps, dtype=np.int32)
    if L.psh_model_generate_sampled(model.h, p.ctypes.data, p.size, batch_size, steps, C.byref(cfg), out.ctypes.data):
        raise HostError(L.psh_last_error().decode())
    return out


class Workspace:
    """A PowerServe work folder (workspace.json -> hparams file, main and optional draft model directories;
    src/core/config.cpp:121-152) opened on the HIP backend: what `powerserve-run --work-folder` sets up, ids in / ids out.
    generate(): through the sampler chain the hparams describe (top_k = 1 is plain greedy); with a draft model configured
    the text is produced by the token tree and the chain picks at every verified node."""

    def __init__(self, work_folder: str, device: int = 0, n_ctx: int = 0):
        self.config = config_summary(work_folder)
        if not self.config["model_main"]:
            raise HostError("workspace.json names no model_main")
        self.batch_size = int(self.config["batch_size"])
        self.main = HostModel(self.config["model_main"], device, max_batch=self.batch_size, n_ctx=n_ctx)
        self.draft = HostModel(self.config["model_draft"], device, max_batch=self.batch_size, n_ctx=n_ctx) if self.config["model_draft"] else None

    def sampler_cfg(self, special_eos_id: int = -1, linefeed_id: int = -1) -> SamplerCfg:
        c = self.config
        return SamplerCfg.make(self.main.vocab, seed=int(c["seed"]), temperature=float(c["temperature"]), top_p=float(c["top_p"]), top_k=int(c["top_k"]),
                               penalty_last_n=int(c["penalty_last_n"]), penalty_repeat=float(c["penalty_repeat"]), penalty_freq=float(c["penalty_freq"]),
                               penalty_present=float(c["penalty_present"]), penalize_nl=c["penalize_nl"] == "1", ignore_eos=c["ignore_eos"] == "1",
                               special_eos_id=special_eos_id, linefeed_id=linefeed_id)

    def generate(self, prompt, steps: int, special_eos_id: int = -1):
        cfg = self.sampler_cfg(special_eos_id=special_eos_id)
        if self.draft is not None:  # the verify samples through the same chain a plain run would use (spec_model.hpp:105)
            sampler = Sampler(cfg)
            try:
                eos = special_eos_id if self.config["ignore_eos"] != "1" else -1
                return spec_generate_sampled(self.main, self.draft, prompt, self.batch_size, steps, sampler, eos=eos)[0]
            finally:
                sampler.close()
        return generate_sampled(self.main, prompt, self.batch_size, steps, cfg)

    def close(self):
        self.main.close()
        if self.draft is not None:
            self.draft.close()
