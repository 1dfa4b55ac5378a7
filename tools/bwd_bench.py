#!/usr/bin/env python3
"""Micro-benchmark of the fused correlation BACKWARD kernels (csrc/corr_bwd.hip) at the cfg-4 per-GPU shape (GPU box only).

Times `itermvs_corr_iter_backward` / `itermvs_corr_init_backward` through their autograd functions (zero-fill of the
gradient tensors included, like a training step pays it) for a noise-like and a smooth normalised depth map, and prints
checksums of the gradients so two library builds (`--lib`) can be compared.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import _lib  # noqa: E402
if "--lib" in sys.argv:     # A/B between library builds (tools only): must be set before the first load
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from itermvs_amd import ops  # noqa: E402
from itermvs_amd.engine import sample_offsets  # noqa: E402
from kernel_bench import build  # noqa: E402


def time_ms(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--lib", default=None, help="another build of libitermvs_hip.so (A/B runs)")
    args = ap.parse_args()
    print("library:", _lib.LIB_PATH)
    dev = torch.device("cuda")
    offs = sample_offsets()
    v = args.views
    gen = torch.Generator().manual_seed(3)
    for kind in ("noise", "smooth"):
        d = build(args.height, args.width, v, kind, dev)
        feats = {l: torch.cat([d["ref"][l]] + d["src"][l], 0).contiguous(memory_format=torch.channels_last).requires_grad_(True) for l in (1, 2, 3)}
        stored = None if args.dtype == "fp32" else {l: feats[l].detach().to(torch.bfloat16) for l in feats}
        rq = d["ref_q"].clone().requires_grad_(True)
        outs = ops.corr_iter_train(feats, 1, v, rq, d["proj"], d["vw"], d["inv_min"], d["inv_max"], d["nd"], offs, stored=stored)
        gouts = [torch.randn(o.shape, generator=gen).to(dev) for o in outs]
        leaves = [feats[1], feats[2], feats[3], rq]
        run = lambda: torch.autograd.grad(outs, leaves, gouts, retain_graph=True)
        ms = time_ms(run)
        g = run()
        print(f"corr_iter_backward depth={kind:6s}: {ms * 1e3:9.1f} us   checksums {[round(float(t.double().abs().sum()), 2) for t in g]}", flush=True)
    d = build(args.height, args.width, v, "noise", dev)
    f3 = torch.cat([d["ref"][3]] + d["src"][3], 0).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    stored3 = None if args.dtype == "fp32" else f3.detach().to(torch.bfloat16)
    out = ops.corr_init_train(f3, 1, v, d["proj"][2], d["inv_min"], d["inv_max"], 32, stored=stored3)
    gout = torch.randn(out.shape, generator=gen).to(dev)
    run = lambda: torch.autograd.grad([out], [f3], [gout], retain_graph=True)
    ms = time_ms(run)
    print(f"corr_init_backward: {ms * 1e3:9.1f} us   checksum {float(run()[0].double().abs().sum()):.2f}")


if __name__ == "__main__":
    main()
