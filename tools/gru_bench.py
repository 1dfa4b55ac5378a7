"""Same-box timing of the ConvGRU's two convolutions: the LDS-tiled launches (conv_tile fp32 gates, conv_tile3 candidate) against
itermvs_gru_conv (csrc/gru.hip), 20 iterations per hipGraph replay, best of several:   python tools/gru_bench.py [reps] [--lib variant.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from itermvs_amd import ops, synthetic
from itermvs_amd.engine import InferenceEngine
from itermvs_amd.net import Pipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
dev = torch.device("cuda:0")


def timed(name, run, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        gr.capture_begin()
        for _ in range(n):
            run()
        gr.capture_end()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
    print(f"{name:72s} {best:7.1f} us")
    return best


m = Pipeline(iteration=4, test=True)
m.load_state_dict(synthetic.random_state_dict(0))
m = m.to(dev).eval()
eng = InferenceEngine(m.weights(), 4)
g = torch.Generator().manual_seed(0)
hx = torch.randn((1, 43, 128, 160), generator=g).to(dev)
hx[:, :32] = torch.tanh(hx[:, :32])
hx2 = hx.clone()
zbuf, hidden = torch.empty((1, 32, 128, 160), device=dev), torch.empty((1, 32, 128, 160), device=dev)
state = hx.clone()
timed("gates: itermvs_conv2d (conv_tile fp32, two results)",
      lambda: ops.conv2d(hx, eng.pk_zr, eng.b_zr, pad=2, dilation=2, act="sigmoid", out=zbuf, aux1=hx[:, :32], split=(32, "gru_rh", hx2[:, :32])))
timed("gates: itermvs_gru_conv mode 0", lambda: ops.gru_conv(hx, eng.gru_zr, eng.b_zr, hx[:, :32], zbuf, out2=hx2[:, :32]))
timed("candidate + update: itermvs_conv2d (conv_tile3)",
      lambda: eng._conv(hx2, "iter_mvs.update.gru.convq.", bias=True, pad=2, dilation=2, act="gru_out", aux1=hx[:, :32], aux2=zbuf, out=state[:, :32], out2=hidden))
timed("candidate + update: itermvs_gru_conv mode 1", lambda: ops.gru_conv(hx2, eng.gru_q, eng.gru_bq, hx[:, :32], state[:, :32], out2=hidden, z=zbuf))
