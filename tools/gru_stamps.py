"""Phase stamps of gru_conv_kernel<4> (a TUNING build -- csrc/experiments/gru_stamps.patch -- whose GRU_STAMPS address receives
cycle-counter stamps of workgroup 17):   python tools/gru_stamps.py --lib tools/ubench/variants/libitermvs_gru_stamps.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
_i = sys.argv.index("--lib")
_lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
from itermvs_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
hx = torch.randn((1, 43, 128, 160), generator=g).to(dev)
w = (torch.randn((64, 43, 3, 3), generator=g) * 0.1).to(dev)
wp = ops.pack_gru_conv_split3(w)
bias = torch.zeros(64, device=dev)
zb, rh = torch.empty((1, 32, 128, 160), device=dev), torch.empty((1, 32, 128, 160), device=dev)
st = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
os.environ["GRU_STAMPS"] = str(st.data_ptr())
hid = torch.tanh(torch.randn((1, 32, 128, 160), generator=g)).to(dev)
wq = ops.pack_gru_conv_split3(w[:32].contiguous())
for mode in (0, 1):
    st.zero_()
    for _ in range(3):
        if mode == 0:
            ops.gru_conv(hx, wp, bias, hx[:, :32], zb, out2=rh)
        else:
            ops.gru_conv(hx, wq, bias[:32], hid, rh, z=zb)
    torch.cuda.synchronize()
    t = st.view(8, 64).cpu()
    print("mode", mode, "(stamps: start | per pair: before barrier, after barrier, matrix done, P stored, epilogue + operands done, loads waited, stashed, advanced, fetch issued)")
    for wv in (0, 3, 4, 7):
        row = t[wv]
        n = int((row != 0).sum())
        print(f"wave {wv}: " + " ".join(str(int(row[j + 1] - row[j])) for j in range(n - 1)) + f" | total {int(row[n - 1] - row[0])}")
