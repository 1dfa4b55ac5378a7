#!/usr/bin/env python3
"""Per-layer micro-benchmark of itermvs_conv2d on the layer shapes of a cfg-1 depth map (GPU box only).
The kernel variant is chosen by the environment (ITERMVS_CONV_TILE=0|1, ITERMVS_CONV_SPLITK=0|1);
an optional argument filters the layers by name (for rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import ops  # noqa: E402

# name, N, Cin, Cout, H, W (input), k, stride, dil, count per depth map
LAYERS = [
    ("fn.conv1 3>8", 5, 3, 8, 512, 640, 3, 1, 1, 1),
    ("fn.l1 8>16 s2", 5, 8, 16, 512, 640, 3, 2, 1, 2),
    ("fn.l1 16>16", 5, 16, 16, 256, 320, 3, 1, 1, 3),
    ("fn.l2 16>32 s2", 5, 16, 32, 256, 320, 3, 2, 1, 2),
    ("fn.l2 32>32", 5, 32, 32, 128, 160, 3, 1, 1, 3),
    ("fn.l3 32>48 s2", 5, 32, 48, 128, 160, 3, 2, 1, 2),
    ("fn.l3 48>48", 5, 48, 48, 64, 80, 3, 1, 1, 4),
    ("fn.inner2 1x1", 5, 32, 48, 128, 160, 1, 1, 1, 1),
    ("fn.out2 48>32", 5, 48, 32, 128, 160, 3, 1, 1, 1),
    ("fn.inner1 1x1", 5, 16, 48, 256, 320, 1, 1, 1, 1),
    ("fn.out1 48>16", 5, 48, 16, 256, 320, 3, 1, 1, 1),
    ("pvw 8>16", 128, 8, 16, 64, 80, 3, 1, 1, 1),
    ("corrnet c0 8>8", 10, 8, 8, 128, 160, 3, 1, 1, 8),
    ("corrnet c1 8>16 s2", 10, 8, 16, 128, 160, 3, 2, 1, 4),
    ("corrnet c2 16>32 s2", 10, 16, 32, 64, 80, 3, 2, 1, 4),
    ("gru 43>32 d2", 1, 43, 32, 128, 160, 3, 1, 2, 12),
    ("head 32>32 d2", 1, 32, 32, 128, 160, 3, 1, 2, 6),
    ("head 1x1 32>64", 1, 32, 64, 128, 160, 1, 1, 1, 5),
    ("head 1x1 64>256", 1, 64, 256, 128, 160, 1, 1, 1, 5),
    ("up 32>64", 1, 32, 64, 128, 160, 3, 1, 1, 1),
    ("up 1x1 64>144", 1, 64, 144, 128, 160, 1, 1, 1, 1),
]


def main():
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(0)
    total = 0.0
    print(f"variant: ITERMVS_CONV_TILE={os.environ.get('ITERMVS_CONV_TILE', '1')} "
          f"SPLITK={os.environ.get('ITERMVS_CONV_SPLITK', '1')}")
    only = sys.argv[1] if len(sys.argv) > 1 else ""     # substring filter on the layer name
    for name, n, cin, cout, h, w, k, stride, dil, count in LAYERS:
        if only not in name:
            continue
        x = torch.randn((n, cin, h, w), generator=gen).to(dev)
        wt = ops.MfmaWeight((torch.randn((cout, cin, k, k), generator=gen) / (cin * k * k) ** 0.5).to(dev))
        pad = dil * (k // 2)
        run = lambda: ops.conv2d(x, wt, None, ksize=k, stride=stride, pad=pad, dilation=dil, act="relu")
        out = run()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        flops = 2.0 * out.numel() * cin * k * k
        total += us * count
        print(f"{name:22s} N={n:3d} {cin:3d}>{cout:3d} {h}x{w} k{k} s{stride} d{dil}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  x{count}")
    print(f"weighted total per depth map: {total:.0f} us")


if __name__ == "__main__":
    main()
