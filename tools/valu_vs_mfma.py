#!/usr/bin/env python3
"""VALU one-thread-per-pixel convolution vs the matrix-core kernels on the layers with almost no contraction
(3->8 full resolution, 8->8 / 8->1 CorrNet, 8->16 PixelViewWeight); hipGraph replays of 20 launches (GPU box only).
Result that set engine._pack_weights: VALU wins for 3->8 (31 vs 39 us) and 8->1 (6.7 vs 10.2 us), loses elsewhere."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import ops
dev = torch.device("cuda")
gen = torch.Generator().manual_seed(0)
for (n, cin, cout, h, w, stride) in [(5, 3, 8, 512, 640, 1), (10, 8, 8, 128, 160, 1), (10, 8, 1, 128, 160, 1), (128, 8, 16, 64, 80, 1)]:
    x = torch.randn((n, cin, h, w), generator=gen).to(dev)
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) / (cin * 9) ** 0.5).to(dev)
    for fmt, pk in (("mfma", ops.MfmaWeight(wt)), ("valu", ops.pack_conv_weight(wt))):
        run = lambda: ops.conv2d(x, pk, None, ksize=3, stride=stride, pad=1, act="relu")
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(); run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); g.capture_begin()
            for _ in range(20): run()
            g.capture_end(); g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): g.replay()
            e1.record(); torch.cuda.synchronize()
        print(f"{cin}>{cout} {h}x{w} N={n} {fmt}: {e0.elapsed_time(e1)/60*1e3:.1f} us")
