"""Phase stamps of lat_conv_kernel (a TUNING build whose out2 pointer receives cycle-counter stamps of workgroup 8's waves):
    python tools/lat_conv_stamps.py --lib tools/ubench/variants/libitermvs_lc_stamps.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
_i = sys.argv.index("--lib")
_lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
from itermvs_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n, h, w = 5, 256, 320
fine = torch.randn((n, 16, h, w), generator=g).relu().to(dev)
coarse = torch.randn((n, 48, h // 2, w // 2), generator=g).to(dev)
wl = (torch.randn((48, 16, 1, 1), generator=g) * 0.25).to(dev)
wo = (torch.randn((16, 48, 3, 3), generator=g) * 0.07).to(dev)
pl, po = ops.MfmaWeight(wl), ops.MfmaWeight(wo, split3=True)
out = torch.empty((n, 16, h, w), device=dev, memory_format=torch.channels_last)
st = torch.zeros((n, 16, h, w), device=dev)
for _ in range(3):
    ops.lateral_conv3x3(fine, coarse, pl, None, po, None, out=out, channels_last_out=True, out2=st)
torch.cuda.synchronize()
t = st.view(-1)[:8 * 256 * 2].view(torch.int64).view(8, 256).cpu()
names = ["loop top", "loads", "first half", "second half", "bar", "loads", "first half", "second half", "bar", "stage", "bar", "setup+loads", "first half", "second half"]
for wv in (0, 3, 5, 7):
    row = t[wv]
    k = 0
    tile_i = 0
    print(f"wave {wv} (first half = {'B' if (wv < 4) else 'A'}, second half = {'A' if (wv < 4) else 'B'})")
    while k + 15 <= 256 and int(row[k + 14]) != 0:
        d = [int(row[k + j + 1] - row[k + j]) for j in range(13)]
        print(f"  tile {tile_i}: " + "  ".join(f"{names[j + 1]} {d[j]}" for j in range(13)) + f"  | next loop top {int(row[k + 14] - row[k + 13])} | total {int(row[k + 14] - row[k])}")
        k += 14
        tile_i += 1
