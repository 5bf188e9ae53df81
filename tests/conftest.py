import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (checker only)."""
    from oracle import binding
    return binding.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference compiled from /root/reference (oracle/_ref); skipped where it was never built."""
    from oracle import binding
    if not binding.have_ref():
        pytest.skip("oracle/_ref/libps_ref.so not built (needs /root/reference)")
    return binding.Ref(2)


@pytest.fixture(scope="session")
def ctx():
    from powerserve_amd import hip
    c = hip.Ctx(0)  # raises loudly when the HIP library or the GPU is missing
    yield c
    c.close()


def load_tensors(path):
    """{name: (ggml_type, raw bytes, ne0, ne1)} of a GGUF file, as oracle.binding.OracleModel takes them"""
    from powerserve_amd import gguf
    rd = gguf.GGUFReader(path)
    out = {}
    for name, ti in rd.tensors.items():
        ne = list(ti.ne) + [1]
        out[name] = (ti.type, np.array(rd.data(name)), ne[0], ne[1])
    return out


def rel_err(a, b):
    """max |a-b| / max |b|  (tensor-relative)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
