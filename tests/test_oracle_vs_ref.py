"""CPU, dev container only: restatement vs the live reference library (oracle/_ref), randomized, incl. Q4_K/Q5_K/Q6_K
which the reference can only run at op level (its loader aborts on K-quants).  Skipped where _ref is absent."""
import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("t", [2, 8, 12, 13, 14])
@pytest.mark.parametrize("K,N,bs", [(256, 40, 1), (1024, 24, 5), (4096, 16, 2)])
def test_mul_mat(oracle, ref, t, K, N, bs):
    from powerserve_amd import synth
    rng = np.random.default_rng(t + K + bs)
    w = synth.random_blocks(rng, t, N, K)
    x = (rng.standard_normal((bs, K)) * rng.choice([0.01, 1, 30])).astype(np.float32)
    yr, ar = ref.mul_mat(t, w, K, N, x, want_act=True)
    yo, ao = oracle.mul_mat(t, w, K, N, x, want_act=True)
    assert np.array_equal(ar, ao)          # activation blocks
    assert np.array_equal(bits(yr), bits(yo))
    # weights produced by the reference's own quantizer (realistic scale statistics)
    wf = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    wq = ref.quantize(t, wf)
    assert np.array_equal(bits(ref.mul_mat(t, wq, K, N, x)), bits(oracle.mul_mat(t, wq, K, N, x)))
    assert np.array_equal(ref.dequantize(t, wq[: ref.row_size(t, K)], K), oracle.dequantize(t, wq[: oracle.row_size(t, K)], K))


def test_thread_count_invariance(ref):
    from oracle import binding as B
    from powerserve_amd import synth
    rng = np.random.default_rng(5)
    w = synth.random_blocks(rng, 12, 64, 1024)
    x = rng.standard_normal((3, 1024)).astype(np.float32)
    r1 = B.Ref(1)
    assert np.array_equal(bits(r1.mul_mat(12, w, 1024, 64, x)), bits(ref.mul_mat(12, w, 1024, 64, x)))
    r1.close()


def test_embedding(oracle, ref):
    from powerserve_amd import synth
    rng = np.random.default_rng(9)
    for t in (0, 2, 8):
        tab = synth.random_blocks(rng, t, 50, 256)
        toks = [0, 49, 7]
        assert np.array_equal(ref.get_embedding(t, tab, 256, 50, toks), oracle.get_embedding(t, tab, 256, toks))
