#!/usr/bin/env python3
"""Where do the kernels' sampling coordinates differ from the oracle's on THIS host?  (GPU box; diagnostic for
tests/test_tap_indices_gpu.py.)  Compares, element by element, the un-normalised coordinates (ix, iy) of
itermvs_tap_indices with the oracle's warp_source_coords and with variants of the oracle that replace ONE step by an
explicitly ordered evaluation, to find the step whose bits depend on the host's BLAS / vector code."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import ops, synthetic  # noqa: E402
from itermvs_amd.engine import sample_offsets  # noqa: E402
from oracle import itermvs_oracle as O  # noqa: E402


def fma32(a, b, c):
    """fp32 fma emulated through fp64 (the product of two floats is exact in double; one extra rounding in ~2^-29 of the cases)"""
    return (a.double() * b.double() + c.double()).float()


def coords_variant(proj, depth, h1, w1, ray="matmul"):
    b, n, h, w = depth.shape
    rot, trans = proj[:, :3, :3], proj[:, :3, 3]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    xs = (xs * (w1 / w)).reshape(-1)
    ys = (ys * (h1 / h)).reshape(-1)
    if ray == "matmul":
        pix = torch.stack((xs, ys, torch.ones(h * w)))
        r = torch.matmul(rot, pix.unsqueeze(0).expand(b, 3, h * w))
    elif ray == "fma_k012":      # fma(m2, 1, fma(m1, y, m0 * x)): a k-ordered dot with one accumulator
        r = torch.stack([fma32(rot[:, i, 1:2], ys.view(1, -1), rot[:, i, 0:1] * xs.view(1, -1)) + rot[:, i, 2:3] for i in range(3)], 1)
    elif ray == "mul_add":       # no fma at all
        r = torch.stack([(rot[:, i, 0:1] * xs.view(1, -1) + rot[:, i, 1:2] * ys.view(1, -1)) + rot[:, i, 2:3] for i in range(3)], 1)
    elif ray == "fma_two_acc":   # (m0 * x + m2) + m1 * y with fma: two accumulators
        r = torch.stack([fma32(rot[:, i, 1:2], ys.view(1, -1), fma32(rot[:, i, 0:1], xs.view(1, -1), rot[:, i, 2:3].expand(b, h * w))) for i in range(3)], 1)
    pts = r.unsqueeze(2) * depth.reshape(b, 1, n, h * w) + trans.view(b, 3, 1, 1)
    X, Y, Z = pts[:, 0], pts[:, 1], pts[:, 2]
    ok = Z > 1e-2
    X = torch.where(ok, X, torch.full_like(X, float(w)))
    Y = torch.where(ok, Y, torch.full_like(Y, float(h)))
    Z = torch.where(ok, Z, torch.ones_like(Z))
    px, py = X / Z, Y / Z
    gx = px / ((w1 - 1) / 2) - 1
    gy = py / ((h1 - 1) / 2) - 1
    return (((gx + 1) / 2) * (w1 - 1)).reshape(b, n, h, w), (((gy + 1) / 2) * (h1 - 1)).reshape(b, n, h, w)


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    print("torch", torch.__version__, "threads", torch.get_num_threads(), "cpu", os.cpu_count())
    print(torch.__config__.show().split("\n")[2:6])
    for views, hh, ww in ((5, 512, 640), (5, 1152, 1600)):
        sm = synthetic.make_sample(1, views, hh, ww, seed=3)
        gen = torch.Generator().manual_seed(5)
        inv_min, inv_max = (1.0 / sm["depth_min"]).view(1, 1, 1, 1), (1.0 / sm["depth_max"]).view(1, 1, 1, 1)
        for lvl in (1, 3):
            pm = sm["proj_matrices"][f"level_{lvl}"]
            proj = torch.stack([O.compose_projection(pm[:, v], pm[:, 0]) for v in range(1, views)], 1)
            p12 = proj[:, :, :3, :4].reshape(1, views - 1, 12).contiguous().to(dev)
            h, w, h1, w1 = hh // 4, ww // 4, hh >> lvl, ww >> lvl
            nd = torch.rand((1, 1, h, w), generator=gen)
            depth = O.iteration_depth_samples(nd, inv_min, inv_max)[lvl]
            _, coords = ops.tap_indices(p12, inv_min.view(1).to(dev), inv_max.view(1).to(dev), (h, w), (h1, w1),
                                        norm_depth=nd.to(dev), offsets=sample_offsets()[lvl], want_coords=True)
            coords = coords.cpu()
            # the kernel's hypotheses vs torch's
            _, cexp = ops.tap_indices(p12, inv_min.view(1).to(dev), inv_max.view(1).to(dev), (h, w), (h1, w1), depth=depth.to(dev), want_coords=True)
            print(f"{ww}x{hh} level {lvl}: generated vs explicit hypotheses: {int((cexp.cpu() != coords).sum())} coordinates differ")
            for s in range(views - 1):
                line = f"  view {s}:"
                for ray in ("matmul", "fma_k012", "fma_two_acc", "mul_add"):
                    ix, iy = coords_variant(proj[:, s], depth, h1, w1, ray)
                    dx = int((ix != coords[:, s, :, 0]).sum() + (iy != coords[:, s, :, 1]).sum())
                    fl = int((torch.floor(ix) != torch.floor(coords[:, s, :, 0])).sum() + (torch.floor(iy) != torch.floor(coords[:, s, :, 1])).sum())
                    line += f"  {ray}: {dx} coords / {fl} floors differ of {2 * ix.numel()};"
                print(line, flush=True)


if __name__ == "__main__":
    main()
