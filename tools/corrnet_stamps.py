"""Phase stamps of corrnet_kernel<true> (a TUNING build whose out2 pointer receives s_memtime stamps of workgroup (7, 3)):
    python tools/corrnet_stamps.py --lib tools/ubench/variants/libitermvs_cn_stamps.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
_i = sys.argv.index("--lib")
_lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
from itermvs_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn((10, 8, 128, 160), generator=g).to(dev)
names = {"conv0.conv.weight": (8, 8, 3, 3), "conv1.conv.weight": (16, 8, 3, 3), "conv2.conv.weight": (32, 16, 3, 3),
         "conv3.weight": (32, 16, 3, 3), "conv4.weight": (16, 8, 3, 3), "conv5.weight": (1, 8, 3, 3), "conv5.bias": (1,)}
wts = {f"p{l}." + k: (torch.randn(v, generator=g) * 0.2).to(dev) for l in range(3) for k, v in names.items()}
packs = [ops.pack_corrnet_weights(wts, f"p{l}.", split3=True) for l in range(3)]
out = torch.empty((10, 1, 128, 160), device=dev)
st = torch.zeros((10, 1, 128, 160), device=dev)
for _ in range(3):
    ops.corrnet(x, packs, (4, 8), out=out, out2=st)
torch.cuda.synchronize()
t = st.view(-1)[:16 * 64 * 2].view(torch.int64).view(16, 64).cpu()
names = ["loads + split + LDS stores", "barrier", "conv0 (bf16x3)", "barrier", "c0 store + weights, barrier", "conv1 + barrier", "weights + barrier",
         "conv2 + barrier", "weights + barrier", "deconv u1 + barrier", "weights + barrier", "deconv u0 + barrier", "(y starts)"]
for wv in (0, 5, 10, 15):
    row = t[wv]
    d = [int(row[j + 1] - row[j]) for j in range(13)]
    print(f"wave {wv:2d}: " + " | ".join(f"{names[j]} {d[j]}" for j in range(12)) + f" | start -> y: {int(row[13] - row[0])} cycles")
