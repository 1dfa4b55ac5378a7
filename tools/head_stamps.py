"""Phase stamps of head_coop_kernel<false, true, true> (a TUNING build -- csrc/experiments/head_stamps.patch -- whose HEAD_STAMPS
address receives cycle-counter stamps of workgroup 17):   python tools/head_stamps.py --lib tools/ubench/variants/libitermvs_head_stamps.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from itermvs_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
hidden = torch.randn((1, 32, 128, 160), generator=g).to(dev)
w0_f32 = (torch.randn((32, 32, 3, 3), generator=g) * 0.1).to(dev)
w1 = (torch.randn((64, 32, 1, 1), generator=g) * 0.1).to(dev)
w2 = (torch.randn((256, 64, 1, 1), generator=g) * 0.1).to(dev)
b2 = torch.randn((256,), generator=g).to(dev)
hw1, _ = ops.pack_head_weights(w1, w2)
hw2 = ops.pack_head_w2_split3(w2)
w0 = ops.pack_head_w0_split3(w0_f32)
hx = torch.zeros((1, 43, 128, 160), device=dev)
st = torch.zeros(4 * 64, dtype=torch.int64, device=dev)
os.environ["HEAD_STAMPS"] = str(st.data_ptr())
for _ in range(3):
    ops.head_fused(hidden, w0, hw1, hw2, b2, nd_out=[(hx, 32)])
torch.cuda.synchronize()
t = st.view(4, 64).cpu()
per = ["barrier", "3x3 layer", "stash + fetch", "barrier", "32 -> 64 layer", "barrier", "64 -> 256 layer + logits", "barrier", "softmax, arg-max, regression"]
names = ["set-up (weights)", "first fetch + stash"] + per * 3
for wv in range(4):
    row = t[wv]
    n = int((row != 0).sum()) - 1
    d = [int(row[j + 1] - row[j]) for j in range(n)]
    print(f"wave {wv}: " + " | ".join(f"{names[j]} {d[j]}" for j in range(n)) + f" | total {int(row[n] - row[0])}")
