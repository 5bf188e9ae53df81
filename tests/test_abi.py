"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/ps_hip.h declares."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "ps_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from powerserve_amd import hip
    L = hip.lib()
    names = declared()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(hip.EXPORTS) == names, set(hip.EXPORTS) ^ set(names)
    assert L.ps_hip_abi_version() == 2  # round 5: + ps_hip_soft_max, the KVCacheInterface leftovers, PS_HIP_ATTN_TIMEOUT


def test_type_helpers_match_ggml():
    from powerserve_amd import hip
    L = hip.lib()
    assert L.ps_hip_row_size(2, 4096) == 4096 // 32 * 18
    assert L.ps_hip_row_size(8, 896) == 896 // 32 * 34
    assert L.ps_hip_row_size(12, 14336) == 14336 // 256 * 144
    assert L.ps_hip_row_size(14, 4096) == 4096 // 256 * 210
    assert L.ps_hip_row_size(15, 4096) == 4096 // 256 * 292
    assert [L.ps_hip_vec_dot_type(t) for t in (2, 8, 12, 14, 0)] == [8, 8, 15, 15, 0]


def test_no_gpu_fails_loudly():
    """On a box without a GPU the backend must refuse to run — never fall back to a CPU path."""
    from powerserve_amd import hip
    if hip.lib().ps_hip_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(hip.PSHipError):
        hip.Ctx(0)


def test_product_never_touches_oracle():
    """powerserve_amd/ (the product) must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "powerserve_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "ps_oracle" not in txt and "libps_ref" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(dp, f)


def test_host_facade_exports():
    """the C++ host facade builds and exports its driver API (no GPU needed to load it)"""
    from powerserve_amd import host
    L = host.lib()
    assert not [n for n in host.EXPORTS if not hasattr(L, n)]


def test_draft_sampler_mirror():
    """TopK(15) -> Temperature(1.5) -> Softmax of the C++ mirror (sampler.cpp:19-58, prob_array.cpp:37-59) against a
    numpy restatement: same tokens in the same order, probabilities equal to rounding, sum 1."""
    from powerserve_amd import host
    rng = np.random.default_rng(3)
    for n in (7, 1024, 128256):
        lg = rng.standard_normal(n).astype(np.float32) * 3
        toks, probs = host.draft_sample(lg, 15, 1.5)
        k = min(15, n)
        order = np.argsort(-lg, kind="stable")[:k]
        assert np.array_equal(toks, order)
        v = lg[order].astype(np.float64) / np.float64(np.float32(1.5))
        want = np.exp(v - v[0]); want /= want.sum()
        assert np.allclose(probs, want, rtol=1e-5, atol=1e-7) and abs(float(probs.sum()) - 1) < 1e-5
