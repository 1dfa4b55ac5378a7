"""Time itermvs_res_chain16 against the three itermvs_conv2d launches it replaces, at the cfg-1 shape (5 x 16 x 256 x 320),
10 launches per hipGraph replay:   python tools/res_chain_bench.py [--lib <other libitermvs_hip.so>] [--only-chain]"""
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
y1 = torch.randn((n, 16, h, w), generator=g).relu().to(dev)
sc = torch.randn((n, 16, h, w), generator=g).to(dev)
wts = [(torch.randn((16, 16, 3, 3), generator=g) * 0.12).to(dev) for _ in range(3)]
bs = [(torch.randn((16,), generator=g) * 0.2).to(dev) for _ in range(3)]
pk = [ops.MfmaWeight(wt, split3=True) for wt in wts]
out = torch.empty_like(y1)
ta, tb = torch.empty_like(y1), torch.empty_like(y1)


def chain():
    ops.res_chain16(y1, sc, pk, bs, out=out)


yq, sq = (t.reshape(n, 4, 4, h, w).permute(0, 1, 3, 4, 2).contiguous() for t in (y1, sc))


def chain_quads():
    ops.res_chain16(yq, sq, pk, bs, out=out, quads=True)


def three():
    ops.conv2d(y1, pk[0], bs[0], act="relu", add=sc, out=ta)
    ops.conv2d(ta, pk[1], bs[1], act="relu", out=tb)
    ops.conv2d(tb, pk[2], bs[2], act="relu", add=ta, out=out)


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


print(f"library {_lib.LIB_PATH if hasattr(_lib, 'LIB_PATH') else ''}")
print(f"res_chain16 (one launch): {timeit(chain):.1f} us")
print(f"res_chain16 (one launch, channel-quad inputs): {timeit(chain_quads):.1f} us")
if "--only-chain" not in sys.argv:
    r = out.clone()
    print(f"three conv_tile3 launches: {timeit(three):.1f} us")
    print("max |chain - three| / max |three| =", float((r - out).abs().max() / out.abs().max()))
