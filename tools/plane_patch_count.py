#!/usr/bin/env python3
"""corr_init as a plane sweep through LDS: how large is the source patch of a pixel tile on one plane?  (CPU, oracle coordinates.)

The 32 initial hypotheses are planes (itermvs.py:11-19): all pixels of a tile share the depth, so the four taps of its pixels
fall into ONE compact patch of each source map.  For the synthetic DTU-like rig at the cfg-1 / cfg-3 / cfg-5 shapes and several
tile shapes, per (tile, plane, view): bounding box of the footprints (the rectangle a staged copy must hold), distinct source
pixels touched, against the 4 taps per pixel the gather form loads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import synthetic  # noqa: E402
from oracle import itermvs_oracle as O  # noqa: E402


def main():
    for name, views, hh, ww in (("cfg 1", 5, 512, 640), ("cfg 3", 5, 1152, 1600), ("cfg 5", 11, 1280, 1920)):
        sm = synthetic.make_sample(1, views, hh, ww, seed=0)
        h, w = hh // 8, ww // 8
        inv_min, inv_max = (1.0 / sm["depth_min"]).view(1, 1, 1, 1), (1.0 / sm["depth_max"]).view(1, 1, 1, 1)
        depth = O.initial_depth_samples(inv_min, inv_max, h, w)
        pm = sm["proj_matrices"]["level_3"]
        print(f"{name}: {views - 1} source views, level-3 grid {w}x{h}")
        for tw, th in ((16, 2), (8, 8), (16, 4), (16, 8), (16, 16)):
            box, distinct, box_w, box_h, inside = [], [], [], [], []
            for s in range(1, views):
                proj = O.compose_projection(pm[:, s], pm[:, 0])
                ix, iy, _ = O.warp_source_coords(proj, depth, h, w)           # [1,32,h,w]
                x0, y0 = torch.floor(ix)[0], torch.floor(iy)[0]
                ok = (x0 >= -1) & (x0 <= w - 1) & (y0 >= -1) & (y0 <= h - 1)  # at least one tap column / row can be inside
                for ty in range(0, h - th + 1, th):
                    for tx in range(0, w - tw + 1, tw):
                        xs, ys, m = x0[:, ty:ty + th, tx:tx + tw], y0[:, ty:ty + th, tx:tx + tw], ok[:, ty:ty + th, tx:tx + tw]
                        for n in range(0, 32, 5):                             # every 5th plane
                            if not bool(m[n].all()):
                                continue
                            bx0, bx1 = int(xs[n].min()), int(xs[n].max()) + 1
                            by0, by1 = int(ys[n].min()), int(ys[n].max()) + 1
                            bw, bh = bx1 - bx0 + 1, by1 - by0 + 1
                            key = set()
                            fx, fy = xs[n].reshape(-1).tolist(), ys[n].reshape(-1).tolist()
                            for a, b in zip(fx, fy):
                                a, b = int(a), int(b)
                                key.update(((a, b), (a + 1, b), (a, b + 1), (a + 1, b + 1)))
                            box.append(bw * bh); distinct.append(len(key)); box_w.append(bw); box_h.append(bh)
            t = torch.tensor
            taps = 4 * tw * th
            bq = torch.quantile(t(box, dtype=torch.float32), torch.tensor([0.5, 0.99, 1.0]))
            print(f"  tile {tw:2d}x{th:<2d}: {taps:4d} taps | distinct pixels median {int(t(distinct).median()):4d} ({t(distinct).float().median() / taps:.2f} of the taps) | "
                  f"bounding box median {int(bq[0]):4d} px ({bq[0] / taps:.2f}), 99 % {int(bq[1]):4d}, max {int(bq[2]):4d}; "
                  f"box width max {max(box_w)}, height max {max(box_h)}")


if __name__ == "__main__":
    main()
