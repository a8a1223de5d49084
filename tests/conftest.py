import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


class Golden:
    """npz fixture reader; keys ending in _bf16 hold bf16 bit patterns."""

    def __init__(self, name):
        self._d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.files = self._d.files

    def __contains__(self, k):
        return k in self._d.files or (k + "_bf16") in self._d.files

    def __getitem__(self, k):
        if k + "_bf16" in self._d.files:
            a = self._d[k + "_bf16"]
            return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        a = self._d[k]
        if a.dtype.kind in "US":
            return a
        return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def synth_weight(n, k, seed, sigma=0.02, heavy=False):
    """Same generator as tests/golden/make_golden.py (large cases are stored as seed + shape)."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n, k, generator=g) * sigma
    if heavy:
        m = torch.rand(n, k, generator=g) < 1e-3
        w = torch.where(m, w * 20, w)
    return w.to(torch.bfloat16)


def synth_scaler(k, seed, zeros=0):
    g = torch.Generator().manual_seed(seed)
    s = torch.exp(torch.randn(k, generator=g) * 0.5) ** 2 * 2048.0
    idx = torch.randperm(k, generator=g)[:4]
    s[idx] *= 900.0
    if zeros:
        s[torch.randperm(k, generator=g)[:zeros]] = 0.0
    return s.float()
