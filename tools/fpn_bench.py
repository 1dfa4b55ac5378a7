"""Time itermvs_fpn_level at the cfg-1 shapes (5 views): python tools/fpn_bench.py [level] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import ops

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
c = 16 * level
h, w = 512 // (2 * level), 640 // (2 * level)
lat = torch.randn((5, c, h, w), generator=g).to(dev)
top = torch.randn((5, 48, h // 2, w // 2), generator=g).to(dev)
wi, bi = torch.randn((48, c, 1, 1), generator=g).to(dev) * 0.1, torch.randn((48,), generator=g).to(dev)
wo, bo = torch.randn((c, 48, 3, 3), generator=g).to(dev) * 0.1, torch.randn((c,), generator=g).to(dev)
pk = ops.pack_fpn_weights(wi, bi, wo, bo)
out = torch.empty((5, c, h, w), device=dev, memory_format=torch.channels_last)
planar = torch.empty((5, c, h, w), device=dev) if level == 2 else None
t_out = torch.empty((5, 48, h, w), device=dev) if level == 2 else None
run = lambda: ops.fpn_level(lat, top, *pk, out=out, out_planar=planar, t_out=t_out)
for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"fpn level {level} mode={os.environ.get('FPN_MODE', '0')}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
