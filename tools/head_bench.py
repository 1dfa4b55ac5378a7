"""Time itermvs_head_fused at the cfg-1 shape (hidden state 32 x 128 x 160): python tools/head_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
if "--lib" in sys.argv:     # a TUNING build (make TUNING=1) reads ITERMVS_HEAD_WGS; the product build does not
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from itermvs_amd import ops

reps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
hidden = torch.randn((1, 32, 128, 160), generator=g).to(dev)
w0_f32 = (torch.randn((32, 32, 3, 3), generator=g) * 0.1).to(dev)
w0 = ops.MfmaWeight(w0_f32)
w1 = (torch.randn((64, 32, 1, 1), generator=g) * 0.1).to(dev)
w2 = (torch.randn((256, 64, 1, 1), generator=g) * 0.1).to(dev)
b2 = torch.randn((256,), generator=g).to(dev)
hw1, hw2 = ops.pack_head_weights(w1, w2)
hx = torch.zeros((1, 43, 128, 160), device=dev)
if "--w2-bf16x3" in sys.argv:      # the 64 -> 256 layer on the bf16 matrix instruction (w2_format 3)
    sys.argv.remove("--w2-bf16x3")
    hw2 = ops.pack_head_w2_split3(w2)
if "--all-bf16x3" in sys.argv:     # ... and the dilated 3x3 layer (w0_format 3)
    sys.argv.remove("--all-bf16x3")
    hw2 = ops.pack_head_w2_split3(w2)
    w0 = ops.pack_head_w0_split3(w0_f32)
run = lambda: ops.head_fused(hidden, w0, hw1, hw2, b2, nd_out=[(hx, 32)])
st = torch.cuda.Stream()
with torch.cuda.stream(st):            # 20 launches per graph replay: no host launch cost between them
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
print(f"head wgs/cu={os.environ.get('ITERMVS_HEAD_WGS', '2 (default)')}: {best:.1f} us per launch")
