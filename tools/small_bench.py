"""Same-box timings of the small launches around the GRU loop and of their fused replacements (round 6), each as 20 launches per
hipGraph replay, best of several replays:   python tools/small_bench.py [reps] [--lib other.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from itermvs_amd import _lib
if "--lib" in sys.argv:
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from itermvs_amd import ops, synthetic
from itermvs_amd.engine import InferenceEngine
from itermvs_amd.net import Pipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
r = lambda *s: torch.randn(s, generator=g).to(dev)

# up-sampling weights (itermvs.py:262-263) and hidden-state initialisation (:159-160)
x2, x3 = r(1, 32, 128, 160), r(1, 32, 64, 80)
u = "iter_mvs.upsample."
hi = "iter_mvs.update.hidden_init_head."
timed("upsample weights: conv3x3 32->64 + conv1x1 64->144, two launches", lambda: eng._conv(eng._conv(x2, u + "0.", act="relu"), u + "2.", ksize=1, pad=0))
timed("upsample weights: itermvs_conv3x3_conv1x1", lambda: ops.conv3x3_conv1x1(x2, eng.pk_up0, eng.up1, None, 144))
_u0, _u1 = ops.MfmaWeight(eng.w[u + "0.weight"], split3=False), ops.pack_conv1x1_operand(eng.w[u + "2.weight"])[0]
timed("upsample weights: itermvs_conv3x3_conv1x1, exact fp32 MFMA", lambda: ops.conv3x3_conv1x1(x2, _u0, _u1, None, 144))
timed("hidden init: conv3x3 32->64 + conv1x1 64->32, two launches", lambda: eng._conv(eng._conv(x3, hi + "0.", act="relu"), hi + "2.", bias=True, ksize=1, pad=0))
timed("hidden init: itermvs_conv3x3_conv1x1", lambda: ops.conv3x3_conv1x1(x3, eng.pk_hi0, eng.hi1, eng.hi1_bias, 32))
xh = r(1, 32, 64, 80)
hid, hx = torch.empty((1, 32, 128, 160), device=dev), torch.zeros((1, 43, 128, 160), device=dev)
timed("hidden init: bilinear x2 + tanh", lambda: ops.bilinear_up_into(xh, 2, hid, hx[:, :32], act="tanh"))

# depth head with / without the confidence head
hidden = torch.tanh(r(1, 32, 128, 160))
p = "iter_mvs.update.depth_head."
conf = torch.empty((1, 1, 128, 160), device=dev)
hx2 = torch.zeros_like(hx)
nd_out = [(hx, 32), (hx2, 32)]
timed("depth head (itermvs_head_fused)", lambda: ops.head_fused(hidden, eng.pk[p + "0.weight"], eng.head_w1, eng.head_w2, eng.w[p + "4.bias"], nd_out=nd_out))
timed("confidence head, own launch (conv relu_dot_sigmoid)", lambda: eng.confidence(hidden, conf))
timed("depth head + confidence head (itermvs_head_fused_conf)", lambda: ops.head_fused(hidden, eng.pk[p + "0.weight"], eng.head_w1, eng.head_w2, eng.w[p + "4.bias"],
                                                                                        nd_out=nd_out, conf=(eng.pk_conf, eng.conf_dot, conf)))

# reference features on the 1/4 grid with / without the camera composition; stem with / without it
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
r1, r2, r3 = cl(r(1, 16, 256, 320)), cl(r(1, 32, 128, 160)), cl(r(1, 48, 64, 80))
s = synthetic.make_sample(batch=1, num_views=5, height=512, width=640, seed=0)
mats = torch.stack([s["proj_matrices"][f"level_{l}"].float() for l in (1, 2, 3)]).reshape(3, 5, 4, 4).to(dev)
dmin, dmax = s["depth_min"].float().to(dev), s["depth_max"].float().to(dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
timed("ref_quarter", lambda: ops.ref_quarter(r1, r2, r3))
timed("ref_quarter_compose (r05 form)", lambda: ops.ref_quarter_compose(r1, r2, r3, mats, flag, (dmin, dmax)))
timed("compose_proj alone", lambda: ops.compose_proj(mats, flag, (dmin, dmax)))
imgs = s["imgs"]["level_0"].float().to(dev).reshape(5, 3, 512, 640).contiguous()
timed("stem", lambda: ops.stem(imgs, *eng.stem_w), n=10)
timed("stem_compose", lambda: ops.stem(imgs, *eng.stem_w, compose=(mats, flag, (dmin, dmax))), n=10)

# initialisation tail
corr_v = r(1, 4, 32, 8, 64, 80)
logit = r(4, 32, 64, 80)
vw = torch.rand((1, 4, 64, 80), generator=g).to(dev)
timed("softmax_max", lambda: ops.softmax_max(logit))
timed("view_aggregate_up", lambda: ops.view_aggregate_up(corr_v, vw, interleaved=True))
pv = "iter_mvs.evaluation.pixel_view_weight."
wpv = eng.w[pv + "conv.0.conv.weight"]
pk32, pk3 = ops.MfmaWeight(wpv, split3=False), ops.MfmaWeight(wpv, split3=True)
xin = corr_v.view(128, 8, 64, 80)
timed("PixelViewWeight conv (relu_dot), fp32 MFMA", lambda: ops.conv2d(xin, pk32, None, act="relu_dot", aux1=eng.pvw_dot))
timed("PixelViewWeight conv (relu_dot), bf16x3 tap pairs", lambda: ops.conv2d(xin, pk3, None, act="relu_dot", aux1=eng.pvw_dot))
a32, a3 = ops.conv2d(xin, pk32, None, act="relu_dot", aux1=eng.pvw_dot), ops.conv2d(xin, pk3, None, act="relu_dot", aux1=eng.pvw_dot)
print("  max |fp32 - bf16x3| / max |fp32| =", float((a32 - a3).abs().max() / a32.abs().max()))
