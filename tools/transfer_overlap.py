#!/usr/bin/env python3
"""Does an H2D copy of one sample overlap a depth-map graph replay on this box?  (GPU box only)
Times, with HIP events: the pinned-host -> device copy alone, one graph replay alone, and both issued together on two
streams.  `together ~ max(copy, replay)` = overlapped; `~ copy + replay` = serialised by the runtime."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import synthetic  # noqa: E402
from itermvs_amd.engine import GraphedRunner, InferenceEngine  # noqa: E402
from itermvs_amd.net import Pipeline  # noqa: E402

dev = torch.device("cuda")
m = Pipeline(iteration=4, test=True)
m.load_state_dict(synthetic.random_state_dict(0))
m = m.to(dev).eval()
eng = InferenceEngine(m.weights(), 4)
s = synthetic.make_sample(1, 5, 512, 640, seed=0)
imgs = s["imgs"]["level_0"].to(dev)
pj = {l: s["proj_matrices"][f"level_{l}"].to(dev) for l in (1, 2, 3)}
r = GraphedRunner(eng, imgs, pj, s["depth_min"].to(dev), s["depth_max"].to(dev))
host = s["imgs"]["level_0"].float().pin_memory()
dst = torch.empty_like(imgs)
s_copy, s_cmp = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def copy_only():
    with torch.cuda.stream(s_copy):
        dst.copy_(host, non_blocking=True)


def replay_only():
    with torch.cuda.stream(s_cmp):
        r(r.imgs, r.projs, r.depth_min, r.depth_max)


def both():
    copy_only()
    replay_only()


print(f"H2D {host.numel() * 4 / 1e6:.1f} MB alone: {wall(copy_only):.3f} ms; replay alone: {wall(replay_only):.3f} ms; "
      f"both on two streams: {wall(both):.3f} ms per pair")
for env in ("HSA_ENABLE_SDMA", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG"):
    print(env, "=", os.environ.get(env))
