"""A short run of the randomised parity stress (tools/stress_parity.py): sorted-size vectors, duplicate / crowded /
out-of-range initial centres, pruned and few-valued data, against the oracle bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_short():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "3", "10"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "10 cases, 0 mismatches" in r.stdout
