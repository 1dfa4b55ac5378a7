"""Time itermvs_corrnet at the cfg-1 shape (the 10 maps of one GRU iteration, 128 x 160): python tools/corrnet_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn((10, 8, 128, 160), generator=g).to(dev)
packs = [(torch.randn((ops.CORRNET_WEIGHT_FLOATS,), generator=g) * 0.1).to(dev) for _ in range(3)]
out = torch.empty((10, 1, 128, 160), device=dev)
run = lambda: ops.corrnet(x, packs, (4, 8), out=out)
for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"corrnet mode={os.environ.get('CN_MODE', '0')}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
