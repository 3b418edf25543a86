import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Goldens:
    def __init__(self):
        d = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(d, "ref_goldens.json")) as f:
            self.manifest = json.load(f)
        self.arrays = np.load(os.path.join(d, "ref_goldens.npz"))
        self.cases = self.manifest["cases"]

    def arr(self, name):
        return self.arrays[name]

    def keys(self, prefix):
        return sorted(k for k in self.cases if k.startswith(prefix))


_G = None


def goldens():
    global _G
    if _G is None:
        _G = Goldens()
    return _G


@pytest.fixture(scope="session")
def gold():
    return goldens()
