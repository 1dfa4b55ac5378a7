#!/usr/bin/env python3
"""Micro-benchmark of the fused correlation kernels on cfg-1-shaped inputs (GPU box only).

Times back-to-back launches with HIP events (torch.cuda.Event on the launch stream) and checks
that the kernel forms agree, for a noise-like and a smooth normalised depth map (the spatial statistics
decide the cache behaviour of the gathers).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import _lib  # noqa: E402
if "--lib" in sys.argv:     # A/B between library builds (tools only): must be set before the first load
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from itermvs_amd import ops, synthetic  # noqa: E402
from itermvs_amd.engine import sample_offsets  # noqa: E402


def build(h, w, views, depth_kind, dev, dtype=torch.float32, vw_interleaved=False):
    gen = torch.Generator().manual_seed(0)
    s = views - 1
    sm = synthetic.make_sample(1, views, h, w, seed=0)
    feats = {1: torch.randn((views, 16, h // 2, w // 2), generator=gen), 2: torch.randn((views, 32, h // 4, w // 4), generator=gen),
             3: torch.randn((views, 48, h // 8, w // 8), generator=gen)}
    cl = {l: f.to(dev).contiguous(memory_format=torch.channels_last).to(dtype) for l, f in feats.items()}      # feature STORAGE type
    src = {l: [cl[l][i:i + 1] for i in range(1, views)] for l in cl}
    ref = {l: cl[l][0:1] for l in cl}
    projs = torch.stack([sm["proj_matrices"][f"level_{l}"] for l in (1, 2, 3)]).to(dev)
    proj = ops.compose_proj(projs.reshape(3, views, 4, 4)).view(3, 1, s, 12)
    hq, wq = h // 4, w // 4
    if depth_kind == "noise":
        nd = torch.rand((1, 1, hq, wq), generator=gen)
    else:
        yy, xx = torch.meshgrid(torch.linspace(0, 1, hq), torch.linspace(0, 1, wq), indexing="ij")
        nd = (0.3 + 0.3 * xx + 0.1 * yy + 0.002 * torch.randn((hq, wq), generator=gen)).view(1, 1, hq, wq)
    vw = torch.rand((1, s, hq, wq), generator=gen).to(dev)
    if vw_interleaved:      # stored [B,H,W,S] (what the engine's view_aggregate_up writes), logical [B,S,H,W]
        vw = vw.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    inv_min = torch.tensor([1 / 425.0], device=dev)
    inv_max = torch.tensor([1 / 935.0], device=dev)
    ref_q = ops.ref_quarter(ref[1], ref[2], ref[3])
    return dict(src=src, ref=ref, ref_q=ref_q, proj=proj, nd=nd.to(dev), vw=vw.to(dev), inv_min=inv_min, inv_max=inv_max)


def time_it(fn, n=20, reps=5):
    """us per launch: n launches captured into one hipGraph (no host launch cost between them), best of `reps` replays"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        g.capture_begin()
        for _ in range(n):
            fn()
        g.capture_end()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--lib", default=None, help="another build of libitermvs_hip.so (A/B runs)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "bf16"], help="feature storage type")
    ap.add_argument("--vw", default="interleaved", choices=["planar", "interleaved"], help="view-weight layout handed to corr_iter")
    args = ap.parse_args()
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    print(f"feature storage {args.dtype}, view weights {args.vw}")
    print("library:", _lib.LIB_PATH)
    dev = torch.device("cuda")
    offs = sample_offsets()
    for kind in ("noise", "smooth"):
        d = build(args.height, args.width, args.views, kind, dev, dt, args.vw == "interleaved")
        buf = [torch.empty((1, len(offs[l]), 8, args.height // 4, args.width // 4), device=dev) for l in (1, 2, 3)]
        run = lambda: ops.corr_iter(d["src"], d["ref_q"], d["proj"], d["vw"], d["inv_min"], d["inv_max"],
                                    norm_depth=d["nd"], offsets=offs, out=buf)
        us = time_it(run)
        print(f"corr_iter depth={kind:6s}: {us:8.2f} us/launch   checksums {[round(float(t.double().abs().sum()), 3) for t in buf]}", flush=True)
    d = build(args.height, args.width, args.views, "noise", dev, dt)
    out = torch.empty((1, args.views - 1, 32, 8, args.height // 8, args.width // 8), device=dev)
    us = time_it(lambda: ops.corr_init(d["src"][3], d["ref"][3], d["proj"][2], d["inv_min"], d["inv_max"], 32, out=out))
    print(f"corr_init: {us:8.2f} us/launch   checksum {float(out.double().abs().sum()):.3f}")


if __name__ == "__main__":
    main()
