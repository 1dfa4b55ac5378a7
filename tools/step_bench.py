#!/usr/bin/env python3
"""ms per depth map of one captured hipGraph (GraphedRunner) with another build of the library (GPU box only; tools A/B runs).

    step_bench.py [--lib path/to/libitermvs_X.so] [--height H --width W --views V --iters I --feature-dtype T] [--steps N]

Knock-out libraries (tools/build_variants.sh) give WRONG results; only the time is meaningful."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--height", type=int, default=512)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--feature-dtype", default="fp32")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--conv-arithmetic", default="bf16x3")
ap.add_argument("--side-branch", action="store_true")
args = ap.parse_args()
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from itermvs_amd import synthetic  # noqa: E402
from itermvs_amd.engine import GraphedRunner, InferenceEngine  # noqa: E402
from itermvs_amd.net import Pipeline  # noqa: E402

dev = torch.device("cuda")
m = Pipeline(iteration=args.iters, test=True)
m.load_state_dict(synthetic.random_state_dict(0))
m = m.to(dev).eval()
eng = InferenceEngine(m.weights(), args.iters, args.feature_dtype, conv_arithmetic=args.conv_arithmetic, side_branch=args.side_branch)
s = synthetic.make_sample(batch=1, num_views=args.views, height=args.height, width=args.width, seed=0)
pj = {l: s["proj_matrices"][f"level_{l}"].float().to(dev) for l in (1, 2, 3)}
r = GraphedRunner(eng, s["imgs"]["level_0"].float().to(dev), pj, s["depth_min"].float().to(dev), s["depth_max"].float().to(dev))
for _ in range(10):
    r.replay()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        r.replay()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / args.steps)
print(f"library {_lib.LIB_PATH}\n{args.views} views {args.width}x{args.height} {args.iters} iters {args.feature_dtype} conv {args.conv_arithmetic} side_branch {args.side_branch}: "
      f"{best:.4f} ms per depth map ({1e3 / best:.1f} depth-maps/s)")
