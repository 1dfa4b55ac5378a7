"""pytest configuration: the ``gpu`` marker and golden-vector helpers.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol export.
``-m gpu``      : HIP kernels (through the C ABI) vs oracle / golden vectors on an MI355X.
"""
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy npz reader returning torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __contains__(self, k):
        return k in self._z.files

    def keys(self):
        return self._z.files

    def np(self, k):
        return self._z[k]

    def __getitem__(self, k):
        a = self._z[k]
        return torch.from_numpy(a.copy()) if a.dtype.kind in "fiub" else a


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


def load_weights(tag):
    g = golden(f"weights_{tag}.npz")
    return {k: g[k] for k in g.keys()}


@pytest.fixture(scope="session")
def weights_seed0():
    return load_weights("seed0")


@pytest.fixture(scope="session")
def weights_dtu():
    return load_weights("dtu")
