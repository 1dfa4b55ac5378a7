"""Time itermvs_head_fused at the cfg-1 shape (hidden state 32 x 128 x 160): python tools/head_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
hidden = torch.randn((1, 32, 128, 160), generator=g).to(dev)
w0 = ops.MfmaWeight((torch.randn((32, 32, 3, 3), generator=g) * 0.1).to(dev))
w1 = (torch.randn((64, 32, 1, 1), generator=g) * 0.1).to(dev)
w2 = (torch.randn((256, 64, 1, 1), generator=g) * 0.1).to(dev)
b2 = torch.randn((256,), generator=g).to(dev)
hw1, hw2 = ops.pack_head_weights(w1, w2)
hx = torch.zeros((1, 43, 128, 160), device=dev)
run = lambda: ops.head_fused(hidden, w0, hw1, hw2, b2, nd_out=[(hx, 32)])
for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"head wgs/cu={os.environ.get('ITERMVS_HEAD_WGS', '2')}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
