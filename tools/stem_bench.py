"""Time itermvs_stem at the cfg-1 shape (5 views, 640x512): python tools/stem_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn((5, 3, 512, 640), generator=g).to(dev)
w0, b0 = torch.randn((8, 3, 3, 3), generator=g).to(dev) * 0.2, torch.randn((8,), generator=g).to(dev)
w1, b1 = torch.randn((16, 8, 3, 3), generator=g).to(dev) * 0.1, torch.randn((16,), generator=g).to(dev)
wd, bd = torch.randn((16, 8, 3, 3), generator=g).to(dev) * 0.1, torch.randn((16,), generator=g).to(dev)
pk = ops.pack_stem_weights(w0, b0, w1, b1, wd, bd)
for _ in range(10):
    ops.stem(x, *pk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.stem(x, *pk)
e1.record()
torch.cuda.synchronize()
print(f"stem TH={os.environ.get('ITERMVS_STEM_TH', '8')}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
