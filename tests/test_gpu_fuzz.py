"""A fixed, seeded stretch of tools/gpu_fuzz.py inside the GPU suite: shapes nobody named (random context windows, prompt lengths, chunkings, max_batch,
token trees behind hidden cache slots, head sizes 32 / 96, 1-8 query heads per kv head), through the C-ABI and through the reference-API path, on bits."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,odd", [(11, 0.0), (12, 0.9)])
def test_seeded_random_sweep(seed, odd):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "--seed", str(seed), "--max-draws", "24", "--seconds", "300",
                        "--ops-share", "0.25", "--odd-share", str(odd)], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "0 failures" in r.stdout, tail
