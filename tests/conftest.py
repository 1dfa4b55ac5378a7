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


def grad_slice(g, n=256):
    """``n`` evenly spaced elements of a gradient tensor (all of it when smaller); the indices tests/golden/make_golden.py
    recorded the reference's gradients at"""
    flat = g.detach().reshape(-1)
    if flat.numel() <= n:
        return flat.clone()
    idx = torch.linspace(0, flat.numel() - 1, n).round().long().to(flat.device)
    return flat[idx]


def check_gradient_slices(g, tag, grads, rel_l2, min_cos, min_checked=10):
    """Full-tensor gradient check against the reference's recorded gradient slices (``{tag}.grad_slices`` of a training
    golden): for every parameter the reference differentiates, relative L2 error <= ``rel_l2`` and cosine >= ``min_cos`` of
    the sliced gradient -- a norm cannot see a wrong direction, this can.  ``grads``: name -> gradient tensor (or None).
    Returns (worst relative L2, worst cosine, parameters checked)."""
    names = [str(x) for x in g.np(f"{tag}.grad_names")]
    lens = [int(x) for x in g.np(f"{tag}.grad_slice_len")]
    flat = g[f"{tag}.grad_slices"].double()
    worst_l2, worst_cos, checked, off = 0.0, 1.0, 0, 0
    for name, n in zip(names, lens):
        want = flat[off:off + n]
        off += n
        if n == 0:
            continue
        got = grads[name]
        assert got is not None, name
        got = grad_slice(got).double().cpu()
        assert got.numel() == n, name
        wn = float(want.norm())
        if wn < 1e-4:                       # rounding-noise gradients (e.g. the bias in front of a softmax: exactly 0 in theory)
            assert float((got - want).norm()) <= 1e-4 * max(rel_l2 / 1e-3, 1.0), name
            continue
        l2 = float((got - want).norm()) / wn
        cos = float((got * want).sum()) / (wn * max(float(got.norm()), 1e-30))
        assert l2 <= rel_l2 and cos >= min_cos, (name, l2, cos)
        worst_l2, worst_cos, checked = max(worst_l2, l2), min(worst_cos, cos), checked + 1
    assert checked >= min_checked
    return worst_l2, worst_cos, checked


def run_ranks(worker, world, *args, attempts=3, timeout=180):
    """Spawn ``world`` CPU processes running ``worker(rank, world, port, *args, queue)`` and return their
    queue items sorted by rank; retried with a fresh rendezvous port if a rank fails to come up."""
    import socket
    import torch.multiprocessing as mp
    last = None
    for _ in range(attempts):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=timeout) for _ in range(world)]
            for p in procs:
                p.join(timeout=60)
            if all(p.exitcode == 0 for p in procs):
                return sorted(res, key=lambda t: t[0])
            last = RuntimeError(f"exit codes {[p.exitcode for p in procs]}")
        except Exception as e:  # noqa: BLE001
            last = e
        for p in procs:
            if p.is_alive():
                p.terminate()
    raise last
