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
    # A/B runs of the suite against another BUILD of the library (tools/build_variants.sh, e.g. -DITERMVS_EXACT_DIV): test
    # infrastructure only -- the product package has no such switch
    alt = os.environ.get("ITERMVS_TEST_LIB")
    if alt:
        from itermvs_amd import _lib
        _lib.LIB_PATH = os.path.abspath(alt)


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


TAP_CASES = ["cfg1_l1", "cfg1_init", "mid_l2", "mid_l3", "mid_l1_smooth", "b2_l2", "behind_l1", "behind_l3"]


def tap_planes(ix, iy, h1, w1):
    """floor / bounds decisions of grid_sample(bilinear, zeros, align_corners=True) on un-normalised coordinates
    [B,N,H,W] -> int32 [B,N,3,H,W] = (floor(ix), floor(iy), bits), with the conventions of itermvs_tap_indices and of
    tests/golden/make_golden.py:tap_planes (NaN -> INT32_MIN, saturation at +-2^30; bit 0: x0 inside, 1: x0+1, 2: y0, 3: y0+1)"""
    fx, fy = torch.floor(ix), torch.floor(iy)

    def to_int(f):
        o = torch.clamp(torch.nan_to_num(f, nan=0.0, posinf=2.0 ** 30, neginf=-2.0 ** 30), -2.0 ** 30, 2.0 ** 30).to(torch.int32)
        return torch.where(torch.isnan(f), torch.full_like(o, -2 ** 31), o)

    bits = ((fx >= 0) & (fx <= w1 - 1)).int() | (((fx + 1 >= 0) & (fx + 1 <= w1 - 1)).int() << 1) | \
           (((fy >= 0) & (fy <= h1 - 1)).int() << 2) | (((fy + 1 >= 0) & (fy + 1 <= h1 - 1)).int() << 3)
    return torch.stack([to_int(fx), to_int(fy), bits.to(torch.int32)], 2)


def grad_slice(g, n=256):
    """``n`` evenly spaced elements of a gradient tensor (all of it when smaller); the indices tests/golden/make_golden.py
    recorded the reference's gradients at"""
    flat = g.detach().reshape(-1)
    if flat.numel() <= n:
        return flat.clone()
    idx = torch.linspace(0, flat.numel() - 1, n).round().long().to(flat.device)
    return flat[idx]


def oracle_training_step(weights, sample, gt, mk, iteration, regress, feature_storage=None, perturb=0.0):
    """one training step of the pinned CPU oracle -> (loss, {parameter name: gradient or None}); ``perturb`` adds
    N(0, perturb) noise to the images (the size of a convolution back-end's rounding) for chaos-floor measurements"""
    from oracle import itermvs_oracle as O
    w = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in weights.items()}
    imgs = {k: v for k, v in sample["imgs"].items()}
    if perturb:
        gen = torch.Generator().manual_seed(99)
        imgs["level_0"] = imgs["level_0"] + perturb * torch.randn(imgs["level_0"].shape, generator=gen)
    out = O.pipeline_forward(w, imgs, sample["proj_matrices"], sample["depth_min"], sample["depth_max"], iteration=iteration,
                             test=False, training=True, feature_storage=feature_storage)
    loss = O.full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mk, sample["depth_min"], sample["depth_max"], regress)
    loss.backward()
    return float(loss.item()), {k: v.grad for k, v in w.items() if v.requires_grad}


def gradient_chaos_floor(weights, sample, gt, mk, iteration, regress, feature_storage=None, perturb=2e-6):
    """Per-parameter relative L2 change of the (sliced) oracle gradient when the images move by ``perturb`` -- rounding-size
    noise.  The training graph is chaotic like inference (an arg-max flip moves a regression window; some gradients are small
    sums of large cancelling terms, e.g. the bias in front of a soft-max): a back-end can only be asked to agree with the
    reference to a few times this floor.  Returns ({name: floor}, base gradients, base loss)."""
    l0, g0 = oracle_training_step(weights, sample, gt, mk, iteration, regress, feature_storage)
    _, g1 = oracle_training_step(weights, sample, gt, mk, iteration, regress, feature_storage, perturb)
    floor = {}
    for n, a in g0.items():
        if a is None:
            continue
        a, b = grad_slice(a).double(), grad_slice(g1[n]).double()
        floor[n] = float((a - b).norm()) / max(float(a.norm()), 1e-30)
    return floor, g0, l0


def check_gradient_slices(g, tag, grads, rel_l2, min_cos, min_checked=10, floor=None, floor_factor=4.0, want=None):
    """Full-tensor gradient check: for every parameter the reference differentiates, the sliced gradient (256 evenly spaced
    elements, conftest.grad_slice) must agree with the reference's recorded one (``{tag}.grad_slices`` of a training golden;
    or ``want``: name -> gradient tensor) in relative L2 (<= ``rel_l2``) and cosine (>= ``min_cos``) -- a norm cannot see a
    wrong direction, this can.  ``floor`` (gradient_chaos_floor): parameters whose gradient is ill-conditioned get
    max(rel_l2, floor_factor * floor) instead, and at least ``min_checked`` parameters must have been held to the base
    tolerance.  ``grads``: name -> gradient tensor (or None).  Returns a report dict."""
    if want is None:
        names = [str(x) for x in g.np(f"{tag}.grad_names")]
        lens = [int(x) for x in g.np(f"{tag}.grad_slice_len")]
        flat = g[f"{tag}.grad_slices"].double()
        want, off = {}, 0
        for name, n in zip(names, lens):
            want[name] = flat[off:off + n] if n else None
            off += n
    else:
        want = {k: (None if v is None else grad_slice(v).double()) for k, v in want.items()}
    rows, strict = [], 0
    for name, ws in want.items():
        if ws is None:
            continue
        got = grads[name]
        assert got is not None, name
        got = grad_slice(got).double().cpu()
        assert got.numel() == ws.numel(), name
        wn = float(ws.norm())
        if wn < 1e-4:                       # rounding-noise gradients (e.g. the bias in front of a softmax: exactly 0 in theory)
            assert float((got - ws).norm()) <= 1e-4 * max(rel_l2 / 1e-3, 1.0), name
            continue
        l2 = float((got - ws).norm()) / wn
        cos = float((got * ws).sum()) / (wn * max(float(got.norm()), 1e-30))
        fl = floor.get(name, 0.0) if floor else 0.0
        bound = max(rel_l2, floor_factor * fl)
        rows.append((l2, cos, fl, bound, name))
        strict += bound == rel_l2
    rows.sort(reverse=True)
    bad = [r for r in rows if r[0] > r[3] or (r[3] == rel_l2 and r[1] < min_cos)]
    report = {"worst_l2": rows[0][0], "worst_cos": min(r[1] for r in rows), "checked": len(rows), "strict": strict,
              "median_l2": sorted(r[0] for r in rows)[len(rows) // 2],
              "worst": [(n, round(l2, 4), round(c, 5), round(f, 4)) for l2, c, f, _, n in rows[:5]]}
    assert not bad, [(n, l2, c, f) for l2, c, f, _, n in bad[:6]]
    assert strict >= min_checked, report
    return report


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
