"""Speculative decoding with a token tree on one GPU: host-side mirror of src/speculative/{token_tree.cpp, spec_model.hpp}
over the C-ABI (powerserve_amd.hip.Model).  The draft model grows a tree of candidate continuations with single-token
forwards (branches are switched by hiding / showing its own cache slots), the target model scores the whole tree in ONE
batched forward with a tree attention mask and per-node RoPE positions, and the longest path the target agrees with is
kept by moving its KV entries into place.  Under greedy sampling the output is exactly the target's own greedy output.

Names, defaults and control flow follow the reference (file:line in the docstrings); the arithmetic all happens in the
HIP backend (ps_hip_model_forward_tree / _kv_mask / _kv_move / _kv_advance / _kv_rollback)."""
from __future__ import annotations

import heapq
from dataclasses import dataclass, field

import numpy as np


@dataclass
class SpeculativeConfig:  # speculative_config.hpp:21-36
    draft_batch_size: int = 12
    top_k: int = 15
    temperature: float = 1.5
    p_base: float = 0.9
    max_fan_out: int = 3
    min_prob: float = 0.2
    early_stop: bool = True


NO_PARENT = -1
NOT_IN_CACHE = -1


@dataclass
class _Node:  # token_tree.hpp:57-72
    parent: int = NO_PARENT
    depth: int = 0
    token: int = 0
    position: int = 0
    cache_index: int = NOT_IN_CACHE
    current_prob: float = 1.0
    accepted: bool = False
    children: list = field(default_factory=list)


def draft_sampler(logits: np.ndarray, top_k: int, temperature: float):
    """TopK -> Temperature -> Softmax (token_tree.cpp:35-39, sampler.cpp): returns (tokens, probs), sorted by prob."""
    k = min(top_k, logits.size)
    idx = np.argpartition(-logits, k - 1)[:k]
    idx = idx[np.argsort(-logits[idx], kind="stable")]
    v = (logits[idx] / np.float32(temperature)).astype(np.float32)
    e = np.exp(v - v.max(), dtype=np.float32)
    return idx.astype(np.int64), (e / e.sum(dtype=np.float32)).astype(np.float32)


class TokenTree:
    def __init__(self, config: SpeculativeConfig):
        self.config = config
        self.nodes: list[_Node] = []
        self.stat = dict(n_draft_times=0, n_draft_tokens=0, n_accepted_tokens=0, n_iterations=0, n_generated_tokens=0)

    # ---- token_tree.cpp:60-94
    def tokens(self):
        return np.array([n.token for n in self.nodes], dtype=np.int32)

    def positions(self):
        return np.array([n.position for n in self.nodes], dtype=np.int32)

    def attention_mask(self):
        bs = len(self.nodes)
        m = np.zeros((bs, bs), dtype=np.uint8)
        for u in range(bs):
            x = u
            while x != NO_PARENT:
                m[u, x] = 1
                x = self.nodes[x].parent
        return m

    # ---- token_tree.cpp:279-315
    def _lca(self, u, v):
        n = self.nodes
        if n[u].depth < n[v].depth:
            u, v = v, u
        while n[u].depth > n[v].depth:
            u = n[u].parent
        while u != v:
            u, v = n[u].parent, n[v].parent
        return u

    def _switch_parent(self, draft_model, old_parent, new_parent):
        if old_parent == new_parent:
            return
        p = self._lca(old_parent, new_parent)
        while old_parent != p:
            draft_model.kv_mask(self.nodes[old_parent].cache_index, False)
            old_parent = self.nodes[old_parent].parent
        while new_parent != p:
            draft_model.kv_mask(self.nodes[new_parent].cache_index, True)
            new_parent = self.nodes[new_parent].parent

    # ---- token_tree.cpp:96-176
    def draft(self, draft_model, batch_size: int, root_token: int, should_stop=lambda tok: False):
        cfg = self.config
        self.nodes = [_Node() for _ in range(batch_size)]
        main_heap, leaf_heap, seq = [], [], 0  # max-heaps on cumulative_prob (heapq is a min-heap: negate; seq breaks ties)
        heapq.heappush(main_heap, (-1.0, seq, int(root_token), NO_PARENT, 1.0))
        last_parent, n_nodes, n_saved = NO_PARENT, 0, 0
        while n_nodes < batch_size:
            is_leaf = not main_heap
            heap = leaf_heap if is_leaf else main_heap
            if not heap:
                break
            neg_cum, _, token, parent, current_prob = heapq.heappop(heap)
            cumulative_prob = -neg_cum
            u = n_nodes
            n_nodes += 1
            node = self.nodes[u]
            node.token, node.current_prob = token, current_prob
            if parent == NO_PARENT:
                node.position = draft_model.position
            else:
                node.position = self.nodes[parent].position + 1
                node.parent, node.depth = parent, self.nodes[parent].depth + 1
                self.nodes[parent].children.append(u)
            if (is_leaf or should_stop(token) or n_nodes + (len(main_heap) // 2 if cfg.early_stop else 0) >= batch_size
                    or cumulative_prob < cfg.min_prob):
                continue
            if last_parent != NO_PARENT:
                self._switch_parent(draft_model, last_parent, parent)
            node.cache_index = draft_model.position
            logits, _ = draft_model.forward_tree([token], [node.position], None, lm_head=True, want_logits=True, advance=True)
            n_saved += 1
            last_parent = u
            toks, probs = draft_sampler(logits[0], cfg.top_k, cfg.temperature)
            min_prob = probs[0] * cfg.p_base
            for i, (t, pr) in enumerate(zip(toks, probs)):
                leaf_only = i >= cfg.max_fan_out or pr < min_prob
                seq += 1
                heapq.heappush(leaf_heap if leaf_only else main_heap, (-(cumulative_prob * float(pr)), seq, int(t), u, float(pr)))
        self.stat["n_draft_times"] += n_saved
        self.stat["n_draft_tokens"] += n_nodes - 1
        draft_model.kv_rollback(n_saved)

    # ---- token_tree.cpp:178-234 (greedy target sampler: the backend returns the arg-max of every node)
    def verify(self, target_model, draft_model, target_argmax, enqueue_token):
        assert target_model.position == draft_model.position
        self.stat["n_iterations"] += 1
        base = target_model.position  # the tree's KV sits at cache slots base + u
        u, n_generated = 0, 0
        while True:
            node = self.nodes[u]
            node.accepted = True
            assert draft_model.position == node.position and target_model.position == node.position
            target_model.kv_move(node.position, base + u)  # kv_cache->copy(node.position, u)
            target_model.kv_advance(1)
            if node.cache_index == NOT_IN_CACHE:  # the draft model catches up with the target
                draft_model.forward_tree([node.token], [node.position], None, lm_head=False, advance=True)
            else:
                assert node.cache_index >= node.position
                draft_model.kv_move(node.position, node.cache_index)
                draft_model.kv_advance(1)
            next_token = int(target_argmax[u])
            enqueue_token(next_token)
            n_generated += 1
            nxt = [v for v in node.children if self.nodes[v].token == next_token]
            if not nxt:
                break
            u = nxt[0]
            self.stat["n_accepted_tokens"] += 1
        self.stat["n_generated_tokens"] += n_generated


class SpeculativeModel:
    """spec_model.hpp: SpecTokenIterator + SpeculativeModel::generate, token ids in / token ids out, greedy target."""

    def __init__(self, target_model, draft_model, config: SpeculativeConfig | None = None):
        self.target_model, self.draft_model = target_model, draft_model
        self.config = config or SpeculativeConfig()
        self.token_tree = TokenTree(self.config)

    def generate(self, prompt, steps: int, batch_size: int = 128):
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        tm, dm, cfg = self.target_model, self.draft_model, self.config
        tm.reset()
        dm.reset()
        for m in (tm, dm):  # prefill all but the last prompt token (spec_model.hpp:54-68)
            done = 0
            while done < prompt.size - 1:
                bs = min(batch_size, prompt.size - 1 - done)
                m.forward(prompt[done:done + bs], np.arange(done, done + bs), lm_head=False)
                done += bs
        out, last = [], int(prompt[-1])
        while len(out) < steps:
            queue = []
            self.token_tree.draft(dm, cfg.draft_batch_size, last)
            _, am = tm.forward_tree(self.token_tree.tokens(), self.token_tree.positions(), self.token_tree.attention_mask(),
                                    lm_head=True, advance=False)  # forward + rollback_tokens(draft_batch_size)
            self.token_tree.verify(tm, dm, am, queue.append)
            out.extend(queue)
            last = queue[-1]
        return np.array(out[:steps], dtype=np.int32)

    def stat(self):
        s = dict(self.token_tree.stat)
        it = max(s["n_iterations"], 1)
        s["tokens_per_iteration"] = s["n_generated_tokens"] / it
        s["draft_forwards_per_iteration"] = s["n_draft_times"] / it
        s["accept_ratio"] = s["n_accepted_tokens"] / max(s["n_draft_tokens"], 1)
        return s
