"""Time itermvs_lateral_conv3x3 against the two launches it replaces (lateral layer with fused F.interpolate, then the bf16x3 3x3
layer), level 1 of cfg 1 (5 x 16 x 256 x 320 fine, 5 x 48 x 128 x 160 coarse), 10 launches per hipGraph replay:
    python tools/lat_conv_bench.py [--lib <other libitermvs_hip.so>] [--only-fused]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
if "--lib" in sys.argv:
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from itermvs_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n, h, w = 5, 256, 320
fine = torch.randn((n, 16, h, w), generator=g).relu().to(dev)
coarse = torch.randn((n, 48, h // 2, w // 2), generator=g).to(dev)
wl = (torch.randn((48, 16, 1, 1), generator=g) * 0.25).to(dev)
bl = (torch.randn((48,), generator=g) * 0.2).to(dev)
wo = (torch.randn((16, 48, 3, 3), generator=g) * 0.07).to(dev)
bo = (torch.randn((16,), generator=g) * 0.2).to(dev)
pl, po = ops.MfmaWeight(wl), ops.MfmaWeight(wo, split3=True)
out = torch.empty((n, 16, h, w), device=dev, memory_format=torch.channels_last)
mid = torch.empty((n, 48, h, w), device=dev)


def fused():
    ops.lateral_conv3x3(fine, coarse, pl, bl, po, bo, out=out, channels_last_out=True)


def two():
    ops.conv2d(fine, pl, bl, ksize=1, pad=0, add=coarse, add_up2=True, out=mid)
    ops.conv2d(mid, po, bo, out=out, channels_last_out=True)


def timeit(run, reps=10, rounds=8):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        gr.capture_begin()
        for _ in range(reps):
            run()
        gr.capture_end()
        best = 1e9
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


if "--eager" in sys.argv:          # for counter passes (rocprofv3 --pmc does not see kernels replayed from a hipGraph)
    for _ in range(int(sys.argv[sys.argv.index("--eager") + 1])):
        fused()
    torch.cuda.synchronize()
    sys.exit(0)
print(f"library {_lib.LIB_PATH}")
print(f"lateral_conv3x3 (one launch): {timeit(fused):.1f} us")
if "--only-fused" not in sys.argv:
    r = out.clone()
    print(f"lateral layer + 3x3 layer (two launches): {timeit(two):.1f} us")
    print("max |fused - two| / max |two| =", float((r - out).abs().max() / out.abs().max()))
