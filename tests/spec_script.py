"""Scripted models for token-tree tests: the Python twin of ScriptedKV / ScriptedModel in oracle/ref_token_tree.cpp (same
integer hash, same three float operations), behind the callback interface of the product's TokenTree
(powerserve_amd.host.SpecBackendCallbacks).  A model's logits depend on the SET of (token, position) entries the new
token can see, so any wrong mask / move / advance shows up in what is generated next, not only in the call log."""
import numpy as np

FORWARD1, FORWARD_TREE, COPY, MOVE, MASK, UNMASK, ADVANCE, ROLLBACK, FORWARD1_NO_LOGITS = range(1, 10)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
C1, C2, G, P2 = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB), np.uint64(0x9E3779B97F4A7C15), np.uint64(0xD1B54A32D192ED03)


def mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30)); z = z * C1
        z = z ^ (z >> np.uint64(27)); z = z * C2
        return z ^ (z >> np.uint64(31))


def entry_hash(tok, pos):
    with np.errstate(over="ignore"):
        return mix(np.uint64(tok) * G + np.uint64(pos) * P2 + np.uint64(1))


def unit(seed, ctx, vocab):
    with np.errstate(over="ignore"):
        v = (np.arange(vocab, dtype=np.uint64) + np.uint64(1)) * G
    return (mix(np.uint64(seed) ^ np.uint64(ctx) ^ v) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


class ScriptedModel:
    def __init__(self, model_id, n_ctx, log, shared_seed, own_seed, shared_w, own_w, vocab, prefix=()):
        self.id, self.log, self.vocab_size = model_id, log, vocab
        self.shared_seed, self.own_seed, self.shared_w, self.own_w = shared_seed, own_seed, np.float32(shared_w), np.float32(own_w)
        self.tok, self.pos, self.vis = np.full(n_ctx, -1, np.int64), np.full(n_ctx, -1, np.int64), np.zeros(n_ctx, bool)
        n = len(prefix)
        self.tok[:n], self.pos[:n], self.vis[:n] = prefix, np.arange(n), True
        self.position = n

    def _past(self):
        ctx = np.uint64(0)
        with np.errstate(over="ignore"):
            for s in range(self.position):
                if self.vis[s]:
                    ctx = ctx + entry_hash(self.tok[s], self.pos[s])
        return ctx

    def _logits(self, ctx):
        return self.shared_w * unit(self.shared_seed, ctx, self.vocab_size) + self.own_w * unit(self.own_seed, ctx, self.vocab_size)

    # ---- the seven calls
    def kv_position(self):
        return self.position

    def forward_one(self, token, position, want_logits):
        self.log.append((self.id, FORWARD1 if want_logits else FORWARD1_NO_LOGITS, token, position))
        out = None
        if want_logits:
            with np.errstate(over="ignore"):
                out = self._logits(self._past() + entry_hash(token, position))
        s = self.position
        self.tok[s], self.pos[s], self.vis[s] = token, position, True
        self.position += 1
        return out

    def forward_tree(self, tokens, positions, mask):
        n, base = len(tokens), self.position
        self.log.append((self.id, FORWARD_TREE, n, base))
        past, am = self._past(), np.zeros(n, np.int32)
        for i in range(n):
            ctx = past
            with np.errstate(over="ignore"):
                for j in range(n):
                    if mask[i, j]:
                        ctx = ctx + entry_hash(tokens[j], positions[j])
            am[i] = int(np.argmax(self._logits(ctx)))  # first maximum, like ProbArray::greedy_sample
        self.tok[base:base + n], self.pos[base:base + n] = tokens, positions  # staged, not yet part of the cache
        return am

    def kv_mask(self, slot, visible):
        self.log.append((self.id, UNMASK if visible else MASK, slot, 0))
        self.vis[slot] = visible

    def kv_move(self, dst, src):
        self.log.append((self.id, MOVE, dst, src))
        self.tok[dst], self.pos[dst] = self.tok[src], self.pos[src]

    def kv_advance(self, n):
        self.log.append((self.id, ADVANCE, n, 0))
        self.vis[self.position:self.position + n] = True
        self.position += n

    def kv_rollback(self, n):
        self.log.append((self.id, ROLLBACK, n, 0))
        assert n <= self.position
        self.position -= n
        self.vis[self.position:self.position + n] = True  # ps_hip_model_kv_rollback un-hides the slots it frees


def normalize_reference_events(ev):
    """The reference's target-side sequence `tree forward (advances by bs); rollback(bs); copy(dst, u)` is the product's
    `tree forward (no advance); move(dst, staging + u)`; everything else is call-for-call the same."""
    out, staging = [], None
    ev = [tuple(int(x) for x in e) for e in ev]
    i = 0
    while i < len(ev):
        m, op, a, b = ev[i]
        if m == 0 and op == FORWARD_TREE:
            assert ev[i + 1] == (0, ROLLBACK, a, 0), "target tree forward must be followed by rollback_tokens(batch)"
            staging = b
            out.append((m, op, a, b))
            i += 2
            continue
        if m == 0 and op == COPY:
            out.append((0, MOVE, a, staging + b))
        else:
            out.append((m, op, a, b))
        i += 1
    return out
