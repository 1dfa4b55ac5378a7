#!/usr/bin/env python3
"""How much of corr_init's gather is repeated along a pixel's epipolar walk?  (CPU only; oracle coordinates.)

The initialisation branch (models/itermvs.py:11-19, 48-51) projects the 32 uniform-inverse-depth hypotheses of every 1/8-res
pixel into each source view: 32 bilinear 2x2 footprints along ONE epipolar segment.  This script counts, on the bench's
synthetic DTU-like cameras (itermvs_amd/synthetic.py) and with the oracle's fp32 coordinate arithmetic
(oracle/itermvs_oracle.py: warp_source_coords), per (pixel, view):
  * the length of the segment in source pixels and the step between consecutive hypotheses,
  * how many of the 32 x 4 tap loads are REPEATS when a pixel's hypotheses are walked in order and the previous footprint
    is kept (same footprint: 4 repeats; footprint moved by one column or one row: 2; diagonal: 1),
  * the number of DISTINCT taps per (pixel, view) (the floor of any register / LDS reuse scheme over the whole walk),
for BASELINE cfg 1 / cfg 3 / cfg 5 shapes.  One JSON line per configuration.

    python tools/epipolar_reuse.py > profiles/r04/r04_epipolar_reuse.txt
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import synthetic  # noqa: E402
from oracle import itermvs_oracle as O  # noqa: E402


def count(views: int, height: int, width: int, seed: int = 0):
    s = synthetic.make_sample(1, views, height, width, seed=seed)
    p3 = s["proj_matrices"]["level_3"]
    h3, w3 = height // 8, width // 8
    inv_min, inv_max = (1.0 / s["depth_min"]).view(1, 1, 1, 1), (1.0 / s["depth_max"]).view(1, 1, 1, 1)
    depth = O.initial_depth_samples(inv_min, inv_max, h3, w3)                        # [1,32,h3,w3]
    tot = dict(loads=0, rep_prev=0, distinct=0, pairs=0, same=0, col=0, row=0, diag=0, none=0)
    seg, step = [], []
    for v in range(1, views):
        m = O.compose_projection(p3[:, v], p3[:, 0])
        ix, iy, _ = O.warp_source_coords(m, depth, h3, w3)                            # [1,32,h3,w3] source coordinates
        x0, y0 = torch.floor(ix[0]).long(), torch.floor(iy[0]).long()                # [32,h3,w3]
        inb = (x0 >= -1) & (x0 < w3) & (y0 >= -1) & (y0 < h3)                         # footprints with at least one tap in range
        dx, dy = x0[1:] - x0[:-1], y0[1:] - y0[:-1]
        both = inb[1:] & inb[:-1]
        same = both & (dx == 0) & (dy == 0)
        col = both & (dx.abs() == 1) & (dy == 0)
        row = both & (dx == 0) & (dy.abs() == 1)
        diag = both & (dx.abs() == 1) & (dy.abs() == 1)
        n_in = int(inb.sum())
        tot["loads"] += 4 * n_in
        tot["rep_prev"] += 4 * int(same.sum()) + 2 * int(col.sum()) + 2 * int(row.sum()) + int(diag.sum())
        tot["pairs"] += int(both.sum())
        for k, t in (("same", same), ("col", col), ("row", row), ("diag", diag)):
            tot[k] += int(t.sum())
        tot["none"] += int((both & ~(same | col | row | diag)).sum())
        # distinct taps of the whole walk per pixel: unique (x, y) over the 32 x 4 taps
        taps = torch.stack([(y0 + oy) * (w3 + 4) + (x0 + ox + 2) for oy in (0, 1) for ox in (0, 1)], 0)   # [4,32,h3,w3]
        taps = torch.where(inb.unsqueeze(0), taps, torch.full_like(taps, -1)).reshape(128, -1).t()        # [P,128]
        srt, _ = torch.sort(taps, 1)
        distinct = ((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum(1) + (srt[:, 0] >= 0).long()
        tot["distinct"] += int(distinct.sum())
        ex, ey = ix[0, -1] - ix[0, 0], iy[0, -1] - iy[0, 0]
        seg.append(torch.sqrt(ex * ex + ey * ey).flatten())
        sx, sy = ix[0, 1:] - ix[0, :-1], iy[0, 1:] - iy[0, :-1]
        step.append(torch.sqrt(sx * sx + sy * sy).flatten())
    seg, step = torch.cat(seg), torch.cat(step)
    return {"views": views, "height": height, "width": width, "level3": [h3, w3],
            "tap_loads_in_range": tot["loads"],
            "repeats_of_previous_footprint": tot["rep_prev"], "repeat_fraction": tot["rep_prev"] / max(tot["loads"], 1),
            "distinct_taps": tot["distinct"], "distinct_fraction": tot["distinct"] / max(tot["loads"], 1),
            "consecutive_pairs": {k: tot[k] / max(tot["pairs"], 1) for k in ("same", "col", "row", "diag", "none")},
            "segment_px_median": float(seg.median()), "segment_px_p90": float(seg.quantile(0.9)),
            "step_px_median": float(step.median()), "step_px_p90": float(step.quantile(0.9))}


if __name__ == "__main__":
    for name, (v, h, w) in (("cfg1", (5, 512, 640)), ("cfg3", (5, 1152, 1600)), ("cfg5", (11, 1280, 1920))):
        r = count(v, h, w)
        r["config"] = name
        print(json.dumps(r))
