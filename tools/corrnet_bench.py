"""Time itermvs_corrnet at the cfg-1 shape (the 10 maps of one GRU iteration, 128 x 160), 20 launches per hipGraph replay:
    python tools/corrnet_bench.py [reps] [--lib <other libitermvs_hip.so>]
Two arms on the same box: x 16-byte aligned (float4 tile staging) and x offset by one float (dword row staging)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
if "--lib" in sys.argv:
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from itermvs_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
shape = (10, 8, 128, 160)
n = 10 * 8 * 128 * 160
buf = torch.zeros((n + 4,), device=dev)
data = torch.randn(shape, generator=g).to(dev)
names = {"conv0.conv.weight": (8, 8, 3, 3), "conv1.conv.weight": (16, 8, 3, 3), "conv2.conv.weight": (32, 16, 3, 3),
         "conv3.weight": (32, 16, 3, 3), "conv4.weight": (16, 8, 3, 3), "conv5.weight": (1, 8, 3, 3), "conv5.bias": (1,)}
wts = {f"p{l}." + k: (torch.randn(v, generator=g) * 0.2).to(dev) for l in range(3) for k, v in names.items()}
packs_by = {s3: [ops.pack_corrnet_weights(wts, f"p{l}.", split3=s3) for l in range(3)] for s3 in (False, True)}
outs = {}
for name, off, s3 in (("aligned (float4 staging), fp32 MFMA", 0, False), ("offset by 4 bytes (dword staging), fp32 MFMA", 1, False),
                      ("aligned, conv0 bf16x3", 0, True), ("offset by 4 bytes, conv0 bf16x3", 1, True)):
    packs = packs_by[s3]
    x = buf[off:off + n].view(shape)
    x.copy_(data)
    out = torch.empty((10, 1, 128, 160), device=dev)
    run = lambda: ops.corrnet(x, packs, (4, 8), out=out)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        gr.capture_begin()
        for _ in range(20):
            run()
        gr.capture_end()
        best = 1e9
        for _ in range(max(3, reps // 20)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    outs[name] = out.clone()
    print(f"corrnet, x {name}: {best:.1f} us per launch")
a, b, c, d = outs.values()
print("fp32 arms bit-identical:", bool(torch.equal(a, b)), " bf16x3 arms bit-identical:", bool(torch.equal(c, d)))
print("max |fp32 - bf16x3| / max |fp32| =", float((a - c).abs().max() / a.abs().max()))
