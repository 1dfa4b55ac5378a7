#!/usr/bin/env python3
"""Per-layer micro-benchmark of itermvs_conv2d on the 3x3 / 1x1 layer shapes of a cfg-1 depth map (GPU box only).

Every layer is captured into a hipGraph of REPS launches and the replay is timed, so the figure is the
kernel time plus the ~4.5 us launch floor of a graph kernel node (no Python launch overhead).
  conv_bench.py [filter]            default dispatch
  conv_bench.py --sweep [filter]    every (tile shape, channel blocking) of the tiled kernel via ITERMVS_TILE_FORCE
The sweep and the other overrides (ITERMVS_TILE_PERSIST, ITERMVS_TILE_MINWORK, ITERMVS_STEM_TH ...) exist only in a library
built with `make -C itermvs_amd/csrc clean all TUNING=1`; the product build has no environment switches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import _lib  # noqa: E402
if "--lib" in sys.argv:     # A/B between library builds (tools only): must be set before the first load
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from itermvs_amd import ops  # noqa: E402

# name, N, Cin, Cout, H, W (input), k, stride, dil, launches per depth map
LAYERS = [
    ("fn.conv1 3>8", 5, 3, 8, 512, 640, 3, 1, 1, 1),
    ("fn.l1 8>16+16 s2", 5, 8, 32, 512, 640, 3, 2, 1, 1),
    ("fn.l1 16>16", 5, 16, 16, 256, 320, 3, 1, 1, 3),
    ("fn.l2 16>32+32 s2", 5, 16, 64, 256, 320, 3, 2, 1, 1),
    ("fn.l2 32>32", 5, 32, 32, 128, 160, 3, 1, 1, 3),
    ("fn.l3 32>48+48 s2", 5, 32, 96, 128, 160, 3, 2, 1, 1),
    ("fn.l3 48>48", 5, 48, 48, 64, 80, 3, 1, 1, 4),
    ("fn.inner2 1x1", 5, 32, 48, 128, 160, 1, 1, 1, 1),
    ("fn.out2 48>32", 5, 48, 32, 128, 160, 3, 1, 1, 1),
    ("fn.inner1 1x1", 5, 16, 48, 256, 320, 1, 1, 1, 1),
    ("fn.out1 48>16", 5, 48, 16, 256, 320, 3, 1, 1, 1),
    ("pvw 8>16", 128, 8, 16, 64, 80, 3, 1, 1, 1),
    ("corrnet c0 8>8", 10, 8, 8, 128, 160, 3, 1, 1, 8),
    ("corrnet c1 8>16 s2", 10, 8, 16, 128, 160, 3, 2, 1, 4),
    ("corrnet c2 16>32 s2", 10, 16, 32, 64, 80, 3, 2, 1, 4),
    ("gru zr 43>64 d2", 1, 43, 64, 128, 160, 3, 1, 2, 4),
    ("gru q 43>32 d2", 1, 43, 32, 128, 160, 3, 1, 2, 4),
    ("head 32>32 d2", 1, 32, 32, 128, 160, 3, 1, 2, 6),
    ("up 32>64", 1, 32, 64, 128, 160, 3, 1, 1, 1),
    ("up 1x1 64>144", 1, 64, 144, 128, 160, 1, 1, 1, 1),
]
REPS = 20


def time_layer(x, wt, k, stride, dil):
    pad = dil * (k // 2)
    run = lambda: ops.conv2d(x, wt, None, ksize=k, stride=stride, pad=pad, dilation=dil, act="relu")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        try:
            out = run()
        except RuntimeError:
            return None, None
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        g.capture_begin()
        for _ in range(REPS):
            run()
        g.capture_end()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * REPS) * 1e3, out


def main():
    if "--fp32" in sys.argv:        # the exact fp32 MFMA form of the layers that default to the bf16x3 split
        ops.MFMA_SPLIT3_DEFAULT = False
        sys.argv.remove("--fp32")
    sweep3 = "--sweep3" in sys.argv  # every (tile shape, channel blocking) of the bf16x3 kernel via ITERMVS_TILE3_FORCE
    if sweep3:
        sys.argv.remove("--sweep3")
    argv = [a for a in sys.argv[1:] if a != "--sweep"]
    sweep = "--sweep" in sys.argv
    only = argv[0] if argv else ""
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(0)
    total = 0.0
    print("library:", _lib.LIB_PATH)
    for name, n, cin, cout, h, w, k, stride, dil, count in LAYERS:
        if only not in name:
            continue
        x = torch.randn((n, cin, h, w), generator=gen).to(dev)
        wt = ops.MfmaWeight((torch.randn((cout, cin, k, k), generator=gen) / (cin * k * k) ** 0.5).to(dev))
        os.environ.pop("ITERMVS_TILE_FORCE", None)
        us, out = time_layer(x, wt, k, stride, dil)
        if us is None:
            print(f"{name:22s} not covered by this build / form", flush=True)
            continue
        flops = 2.0 * out.numel() * cin * k * k
        total += us * count
        line = f"{name:22s} N={n:3d} {cin:3d}>{cout:3d} {h}x{w} k{k} s{stride} d{dil}: {us:7.1f} us {flops / us / 1e6:6.1f} TFLOP/s x{count}"
        if sweep and k == 3:
            res = []
            for shape in (2, 1, 0):
                for mb in (3, 2, 1):
                    os.environ["ITERMVS_TILE_FORCE"] = f"{shape},{mb}"
                    t, _ = time_layer(x, wt, k, stride, dil)
                    if t is not None:
                        res.append((t, shape, mb))
            os.environ.pop("ITERMVS_TILE_FORCE", None)
            res.sort()
            line += " | " + " ".join(f"s{sh}m{mb}:{t:.1f}" for t, sh, mb in res)
        if sweep3 and k == 3 and cin > 8:
            res = []
            for shape in (2, 1, 0):
                for mb in (3, 2, 1):
                    os.environ["ITERMVS_TILE3_FORCE"] = f"{shape},{mb}"
                    t, _ = time_layer(x, wt, k, stride, dil)
                    if t is not None:
                        res.append((t, shape, mb))
            os.environ.pop("ITERMVS_TILE3_FORCE", None)
            res.sort()
            line += " | " + " ".join(f"s{sh}m{mb}:{t:.1f}" for t, sh, mb in res)
        print(line, flush=True)
    print(f"weighted total per depth map: {total:.0f} us")


if __name__ == "__main__":
    main()
