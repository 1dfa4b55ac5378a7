#!/usr/bin/env python3
"""itermvs_fuse_depth (eval.py:154-269) on one MI355X: reference views per second at the DTU
evaluation size, HBM roofline of the kernel, and the numpy restatement (oracle/fusion_oracle.py) on the host beside it.
One JSON line.  usage: fusion_bench.py [--height 1152 --width 1600 --src 10 --steps 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from itermvs_amd import fusion, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--src", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    from test_fusion import _scene
    views = _scene(a.height, a.width, a.src + 1, 0, 0.002)
    dev = torch.device("cuda")
    k, e, d, conf = views[0]
    mats = torch.from_numpy(np.stack([fusion.pair_matrices(k, e, v[0], v[1]) for v in views[1:]])).to(dev)
    dref, cref = torch.from_numpy(d).to(dev), torch.from_numpy(conf).to(dev)
    srcs = [torch.from_numpy(v[2]).to(dev) for v in views[1:]]
    run = lambda: ops.fuse_depth(dref, cref, srcs, mats)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    px = a.height * a.width
    alg = px * ((2 + a.src) * 4 + 8 + 3 + 4)          # maps read once; float64 average, three masks and the count written
    out = {"metric": "reference views fused per second (geometric + photometric filter, eval.py:154-269)",
           "value": 1e3 / ms, "unit": "ref-views/s", "ms_per_view": ms, "dtype": "f64 geometry, f32 maps",
           "config": {"workload": f"{a.src} source views, {a.width}x{a.height} depth maps, synthetic plane scene"},
           "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": alg, "traffic": None}}
    if not a.no_cpu_baseline:
        from oracle import fusion_oracle as FO
        n_src = min(a.src, 2)                        # bounded sample: two source views, scaled to all of them
        t0 = time.perf_counter()
        FO.fuse_reference_view(d, conf, k, e, [v[2] for v in views[1:1 + n_src]], [v[0] for v in views[1:1 + n_src]],
                               [v[1] for v in views[1:1 + n_src]])
        dt = (time.perf_counter() - t0) * a.src / n_src
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "ref-views/s", "cores": 1, "kind": "port",
                               "sample": f"numpy restatement on {n_src} of the {a.src} source views, scaled linearly"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
