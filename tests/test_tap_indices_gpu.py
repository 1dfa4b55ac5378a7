"""`north_star`: "pixel indices bit-exact".  The fused correlation kernels never store a sampling position, so
itermvs_tap_indices evaluates floor(ix), floor(iy) and the bounds bits with the kernels' own device functions
(iter_hypothesis / init_hypothesis, ray_dir, project_fast with the library's reciprocal-based division, make_taps) and these
tests hold every (pixel, hypothesis, view, level) of

  * tests/golden/tap_cases.npz -- the reference's OWN sampling grids, observed at F.grid_sample while models/module.py:68-125 ran
    on hypotheses the reference built itself (itermvs.py:11-19, :290-293): 1.1 M footprints incl. the batch == 2 branch and
    behind-camera pixels;
  * the full cfg-1 / cfg-3 / cfg-5 shapes, initialisation and iteration branch, noise and smooth depth -- against the oracle's
    warp_source_coords(ray_dot="fma"), which tests/test_oracle_golden.py pins bit for bit on the same fixture on any host
    (the reference's own torch.matmul rounds its K = 3 ray product differently on Intel and AMD hosts),

to EQUALITY (reference models/module.py:99-119, ATen GridSampler.h:31,205-207)."""
import pytest
import torch

from conftest import TAP_CASES, golden, tap_planes
from oracle import itermvs_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def ops():
    from itermvs_amd import ops as _ops
    return _ops


def offsets(level):
    from itermvs_amd.engine import sample_offsets
    return sample_offsets()[level]


def proj12(proj44):
    """[B,S,4,4] composed projections -> the [B,S,12] rows [R | t] the kernels read"""
    return proj44[:, :, :3, :4].reshape(proj44.shape[0], proj44.shape[1], 12).contiguous()


def mismatch_report(got, want, what):
    bad = got != want
    n = int(bad.sum())
    if n:
        idx = bad.nonzero()[:5].tolist()
        return f"{what}: {n} of {want.numel()} tap decisions differ, first at {idx}"
    return None


@pytest.mark.parametrize("case", TAP_CASES)
def test_tap_indices_equal_the_references_own_grids(case):
    g = golden("tap_cases.npz")
    lvl, h, w, h1, w1, init = (int(v) for v in g.np(f"{case}.meta"))
    proj, depth, want = g[f"{case}.proj"], g[f"{case}.depth"], g[f"{case}.taps"]
    p12, inv_min, inv_max = proj12(proj).to(DEV), g[f"{case}.inv_min"].to(DEV), g[f"{case}.inv_max"].to(DEV)
    # (a) explicit hypotheses: the reference's own depth samples
    got = ops().tap_indices(p12, inv_min, inv_max, (h, w), (h1, w1), depth=depth.to(DEV)).cpu()
    assert got.shape == want.shape
    msg = mismatch_report(got, want, f"{case} explicit hypotheses")
    assert msg is None, msg
    # (b) hypotheses built in-kernel, as the engine runs (initial planes / offsets around the normalised depth)
    if init:
        got = ops().tap_indices(p12, inv_min, inv_max, (h, w), (h1, w1), init_samples=depth.shape[1]).cpu()
    else:
        got = ops().tap_indices(p12, inv_min, inv_max, (h, w), (h1, w1), norm_depth=g[f"{case}.nd"].to(DEV), offsets=offsets(lvl)).cpu()
    msg = mismatch_report(got, want, f"{case} generated hypotheses")
    assert msg is None, msg


SHAPES = {"cfg1": (5, 512, 640), "cfg3": (5, 1152, 1600), "cfg5": (11, 1280, 1920)}


@pytest.mark.parametrize("shape", list(SHAPES))
def test_tap_indices_full_size_against_the_pinned_oracle(shape):
    """every footprint of a depth map at the BASELINE shapes: initialisation branch (32 planes at level 3) and the three
    levels of the iteration branch for a noise and a smooth normalised depth map, projections composed in fp32 on the host
    like module.py:77-90"""
    from itermvs_amd import synthetic
    views, hh, ww = SHAPES[shape]
    sm = synthetic.make_sample(1, views, hh, ww, seed=3)
    gen = torch.Generator().manual_seed(5)
    inv_min, inv_max = (1.0 / sm["depth_min"]).view(1, 1, 1, 1), (1.0 / sm["depth_max"]).view(1, 1, 1, 1)
    total = bad = cdiff = 0
    reports = []
    for lvl in (3, 1, 2):
        pm = sm["proj_matrices"][f"level_{lvl}"]
        proj = torch.stack([O.compose_projection(pm[:, v], pm[:, 0]) for v in range(1, views)], 1)      # [1,S,4,4]
        p12 = proj12(proj).to(DEV)
        h1, w1 = hh >> lvl, ww >> lvl
        jobs = []
        if lvl == 3:
            jobs.append(("init", (hh // 8, ww // 8), dict(init_samples=32), O.initial_depth_samples(inv_min, inv_max, hh // 8, ww // 8)))
        h, w = hh // 4, ww // 4
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        for kind, nd in (("noise", torch.rand((1, 1, h, w), generator=gen)),
                         ("smooth", (0.3 + 0.3 * xx + 0.1 * yy + 0.002 * torch.randn((h, w), generator=gen)).view(1, 1, h, w))):
            jobs.append((f"iter-{kind}", (h, w), dict(norm_depth=nd.to(DEV), offsets=offsets(lvl)),
                         O.iteration_depth_samples(nd, inv_min, inv_max)[lvl]))
        for name, grid, kw, depth in jobs:
            got, coords = ops().tap_indices(p12, inv_min.view(1).to(DEV), inv_max.view(1).to(DEV), grid, (h1, w1), want_coords=True, **kw)
            got, coords = got.cpu(), coords.cpu()
            for s in range(views - 1):
                # ray_dot="fma": the reference's arithmetic as its BLAS rounded it on the golden host (host-independent;
                # torch.matmul on this box's EPYC rounds the K = 3 dot without fma -- oracle docstring, tools/tap_probe.py)
                ix, iy, _ = O.warp_source_coords(proj[:, s], depth, h1, w1, ray_dot="fma")
                want = tap_planes(ix, iy, h1, w1)
                msg = mismatch_report(got[:, s], want, f"{shape} level {lvl} {name} view {s}")
                total += want.numel()
                # the coordinates themselves (hence the bilinear weights): the reciprocal-based division of project_fast against
                # four IEEE divisions per footprint, NaN == NaN
                cdiff += int(((coords[:, s, :, 0] != ix) & ~(ix.isnan() & coords[:, s, :, 0].isnan())).sum()
                             + ((coords[:, s, :, 1] != iy) & ~(iy.isnan() & coords[:, s, :, 1].isnan())).sum())
                if msg:
                    bad += int((got[:, s] != want).sum())
                    reports.append(msg)
    print(f"{shape}: {total} tap decisions compared, {bad} differ; {cdiff} of {total // 3 * 2} coordinates differ in any bit")
    assert bad == 0, "\n".join(reports[:8])
    # (the checker emulates fma through fp64: a second rounding in ~2^-29 of the ray components is its own, not the kernel's)
    assert cdiff <= 1e-6 * total, cdiff
