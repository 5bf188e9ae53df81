"""Host-side sampler chain (powerserve_amd/csrc/host/sampler.cpp) against the reference's own sampler classes compiled
in place into oracle/_ref (src/sampler/sampler.cpp, prob_array.cpp; chained in the order of sampler_chain.cpp:19-51):
identical token sequences for identical logits, configuration and seed — std::mt19937 + discrete_distribution included."""
import ctypes as C

import numpy as np
import pytest

CASES = [
    dict(),                                                                    # reference defaults: top-k 40, T 0.8, top-p 0.95
    dict(top_k=1),                                                             # greedy through the chain
    dict(top_k=1000, top_p=1.0, temperature=1.0),                              # plain multinomial over the vocabulary
    dict(temperature=1.7, top_p=0.6, top_k=12, seed=99),
    dict(penalty_repeat=1.3, penalty_freq=0.2, penalty_present=0.1, penalty_last_n=16, penalize_nl=True),
    dict(penalty_repeat=1.15, penalty_last_n=8, penalize_nl=False, linefeed_id=13, ignore_eos=True, special_eos_id=2, seed=5),
]


@pytest.mark.parametrize("kw", CASES)
def test_sampler_chain_matches_reference(ref, kw):
    from oracle import binding as B
    from powerserve_amd import host
    n_vocab, steps = 1000, 300
    cfg = host.SamplerCfg.make(n_vocab, **kw)
    mine = host.Sampler(cfg)
    rc = B.SamplerCfg.from_buffer_copy(bytes(cfg))  # same plain-C layout on both sides
    rh = ref.L.ref_sampler_create(C.byref(rc))
    rng = np.random.default_rng(17)
    got, want = [], []
    for s in range(steps):
        lg = (rng.standard_normal(n_vocab) * 2.5).astype(np.float32)
        if s % 7 == 0:
            lg[rng.integers(0, n_vocab, 3)] = lg.max()  # exact ties at the top
        got.append(mine.sample(lg))
        want.append(ref.L.ref_sampler_sample(rh, lg.ctypes.data, n_vocab))
    ref.L.ref_sampler_free(rh)
    mine.close()
    assert got == want
    assert len(set(got)) > (1 if kw.get("top_k") != 1 else 0)
