"""CPU oracle for the IterMVS matching hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain torch-CPU tensor
math, explicit per-pixel geometry, no nn.Module) of the algorithm implemented
by the reference's ``models/module.py``, ``models/itermvs.py`` and
``models/net.py``.  Every function cites the reference file:line it follows.

Rules (see DESIGN.md "Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import this module; the product package
    ``itermvs_amd`` never does;
  * parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here
    against golden vectors captured from the real reference implementation
    (imported read-only in the build container by
    ``tests/golden/make_golden.py``; the vectors are committed, the reference
    is not).

Third-party arithmetic on the path (PyTorch, pinned ``torch==1.4.0`` in the
reference's requirements.txt:1, oracle runs on torch 2.10) that is restated
explicitly instead of being called: ``F.grid_sample`` (bilinear, zeros padding,
align_corners=True; ATen/native/GridSampler.h:27-36), ``F.interpolate`` for the
exact x2 / x0.5 / x4 bilinear cases used (align_corners=False), ``F.unfold`` +
``ReplicationPad2d`` in the convex upsampler, ``softmax``/``argmax``/``gather``
in the depth regression.  Dense 2-D convolutions call ``F.conv2d`` /
``F.conv_transpose2d``.

All helper tensors are created on the device of the inputs, so the same restatement can be
run with PyTorch-ROCm tensors by the diagnostics under tests/ ("what does the reference's
algorithm give on this GPU with stock PyTorch ops").

Weights are passed as a flat ``dict[str, Tensor]`` keyed like the reference's
state_dict without the ``module.`` prefix.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]

GROUPS = 8            # itermvs.py:28
INIT_SAMPLES = 32     # itermvs.py:237
PROB_BINS = 256       # itermvs.py:134
WINDOW_RADIUS = 4     # itermvs.py:135
INTERVAL_SCALE = 1.0 / 256  # itermvs.py:229
# itermvs.py:231-235 (float32 constants, multiplied by INTERVAL_SCALE at use)
CORR_INTERVAL = {
    1: (-2.0, -2.0 / 3, 2.0 / 3, 2.0),
    2: (-8.0, -8.0 / 3, 8.0 / 3, 8.0),
    3: (-32.0, 32.0),
}


def strip_module_prefix(state: Dict[str, Tensor]) -> Weights:
    """eval.py:119,125 -- checkpoints are saved from a DataParallel wrapper."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}


# --------------------------------------------------------------------------
# depth <-> normalised inverse depth                      module.py:142-152
# --------------------------------------------------------------------------
def depth_normalization(depth: Tensor, inv_min: Tensor, inv_max: Tensor) -> Tensor:
    """module.py:142-146."""
    inv = 1.0 / (depth + 1e-5)
    return (inv - inv_max) / (inv_min - inv_max)


def depth_unnormalization(nd: Tensor, inv_min: Tensor, inv_max: Tensor) -> Tensor:
    """module.py:148-152.  n=0 -> depth_max, n=1 -> depth_min."""
    return 1.0 / (inv_max + nd * (inv_min - inv_max))


def initial_depth_samples(inv_min: Tensor, inv_max: Tensor, h: int, w: int,
                          num: int = INIT_SAMPLES) -> Tensor:
    """itermvs.py:11-19: ``num`` hypotheses uniform in inverse depth -> [B,num,h,w]."""
    b = inv_min.shape[0]
    k = torch.arange(num, dtype=torch.float32, device=inv_min.device).view(1, num, 1, 1)
    frac = k.expand(b, num, h, w) / (num - 1)
    return 1.0 / (inv_max + frac * (inv_min - inv_max))


# --------------------------------------------------------------------------
# exact-ratio bilinear resampling used on the path (align_corners=False)
# --------------------------------------------------------------------------
def resize_bilinear(x: Tensor, scale: float) -> Tensor:
    """Explicit restatement of ``F.interpolate(x, scale_factor=scale,
    mode='bilinear')`` (align_corners=False) for the ratios the path uses
    (itermvs.py:56,80,97,161,310,323; net.py:46,49,62,64).
    src = (dst + 0.5) / scale - 0.5, clamped below at 0; upper neighbour
    clamped at size-1."""
    b, c, h, w = x.shape
    oh, ow = int(math.floor(h * scale)), int(math.floor(w * scale))

    def axis(n_in: int, n_out: int):
        d = torch.arange(n_out, dtype=torch.float32, device=x.device)
        s = (d + 0.5) * (1.0 / scale) - 0.5
        s = torch.clamp(s, min=0.0)
        i0 = s.floor().to(torch.int64)
        i0 = torch.clamp(i0, max=n_in - 1)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        l1 = s - i0.to(torch.float32)
        return i0, i1, 1.0 - l1, l1

    y0, y1, hy0, hy1 = axis(h, oh)
    x0, x1, hx0, hx1 = axis(w, ow)
    top = x[:, :, y0][:, :, :, x0] * hx0 + x[:, :, y0][:, :, :, x1] * hx1
    bot = x[:, :, y1][:, :, :, x0] * hx0 + x[:, :, y1][:, :, :, x1] * hx1
    return top * hy0.view(1, 1, oh, 1) + bot * hy1.view(1, 1, oh, 1)


# --------------------------------------------------------------------------
# a1: homography warp + bilinear gather                    module.py:68-125
# --------------------------------------------------------------------------
def compose_projection(src_proj: Tensor, ref_proj: Tensor) -> Tensor:
    """module.py:77-87: ``src_proj @ inverse(ref_proj)`` per batch element."""
    inv = torch.stack([torch.inverse(ref_proj[i]) for i in range(ref_proj.shape[0])])
    assert not torch.isnan(inv).any(), "nan in inverse(ref_proj)"
    proj = torch.matmul(src_proj, inv)
    assert not torch.isnan(proj).any(), "nan in proj"
    return proj


def _fma32(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """fp32 fused multiply-add through fp64: the product of two floats is exact in double, so this differs from a hardware
    fma only by a second rounding in ~2^-29 of the cases."""
    return (a.double() * b.double() + c.double()).float()


def warp_source_coords(proj: Tensor, depth: Tensor, h1: int, w1: int, ray_dot: str = "matmul"
                       ) -> Tuple[Tensor, Tensor, Tensor]:
    """module.py:89-115 + GridSampler.h:31 -- source-pixel coordinates.

    proj  [B,4,4]  pre-multiplied src @ inv(ref)
    depth [B,N,H,W] hypotheses on the sample grid
    returns (ix, iy, valid) each [B,N,H,W]: un-normalised sampling position in
    the H1 x W1 source map and the (never requested) valid mask.

    ``ray_dot``: how ``rot @ xyz`` (module.py:99, a K = 3 dot product inside a BLAS sgemm) is rounded.  "matmul" calls
    ``torch.matmul`` like the reference -- whose bits depend on the host: MKL evaluates it as the k-ordered fma chain
    fma(r2, 1, fma(r1, y, r0 * x)) on the Intel host the golden vectors were captured on, and WITHOUT fma on an AMD
    EPYC host (7 % of the coordinates then differ in the last bit, ~1e-6 of the floors; tools/tap_probe.py).  "fma" is that
    k-ordered fma chain written out: host-independent, bit-identical to the reference on the golden host
    (tests/test_oracle_golden.py::test_sampling_positions_floor_equal_the_references) -- the checker of the GPU tap tests.
    """
    b, n, h, w = depth.shape
    rot = proj[:, :3, :3]
    trans = proj[:, :3, 3]
    dev = depth.device
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev),
                            torch.arange(w, dtype=torch.float32, device=dev), indexing="ij")
    xs = xs * (w1 / w)                      # module.py:95-96 (python float ratio)
    ys = ys * (h1 / h)
    if ray_dot == "fma":
        xf, yf = xs.reshape(1, -1), ys.reshape(1, -1)
        ray = torch.stack([_fma32(rot[:, i, 1:2], yf, rot[:, i, 0:1] * xf) + rot[:, i, 2:3] for i in range(3)], 1)
    else:
        pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(h * w, device=dev)))  # [3,HW]
        ray = torch.matmul(rot, pix.unsqueeze(0).expand(b, 3, h * w))       # [B,3,HW]
    pts = ray.unsqueeze(2) * depth.reshape(b, 1, n, h * w)                  # [B,3,N,HW]
    pts = pts + trans.view(b, 3, 1, 1)
    X, Y, Z = pts[:, 0], pts[:, 1], pts[:, 2]
    ok = Z > 1e-2                                                          # module.py:105
    X = torch.where(ok, X, torch.full_like(X, float(w)))   # sample-grid W/H, :106-107
    Y = torch.where(ok, Y, torch.full_like(Y, float(h)))
    Z = torch.where(ok, Z, torch.ones_like(Z))
    px = X / Z
    py = Y / Z
    valid = ok & (px >= 0) & (px < w) & (py >= 0) & (py < h)               # :110-111
    gx = px / ((w1 - 1) / 2) - 1                                           # :112-113
    gy = py / ((h1 - 1) / 2) - 1
    ix = ((gx + 1) / 2) * (w1 - 1)                                         # GridSampler.h:31
    iy = ((gy + 1) / 2) * (h1 - 1)
    shape = (b, n, h, w)
    return ix.reshape(shape), iy.reshape(shape), valid.reshape(shape)


def bilinear_gather(src: Tensor, ix: Tensor, iy: Tensor) -> Tensor:
    """F.grid_sample(bilinear, zeros, align_corners=True) restated on
    un-normalised coordinates: src [B,C,H1,W1]; ix,iy [B,N,H,W] -> [B,C,N,H,W].
    A tap contributes only if it lies inside the map; NaN/inf coordinates give 0."""
    b, c, h1, w1 = src.shape
    shp = ix.shape
    fx0 = torch.floor(ix)
    fy0 = torch.floor(iy)
    fx1 = fx0 + 1
    fy1 = fy0 + 1
    w_nw = (fx1 - ix) * (fy1 - iy)
    w_ne = (ix - fx0) * (fy1 - iy)
    w_sw = (fx1 - ix) * (iy - fy0)
    w_se = (ix - fx0) * (iy - fy0)
    flat = src.reshape(b, c, h1 * w1)
    out = torch.zeros((b, c) + tuple(shp[1:]), dtype=src.dtype, device=src.device)

    def tap(fx: Tensor, fy: Tensor, wt: Tensor) -> None:
        inside = (fx >= 0) & (fx <= w1 - 1) & (fy >= 0) & (fy <= h1 - 1)   # NaN -> False
        xi = torch.where(inside, fx, torch.zeros_like(fx)).to(torch.int64)
        yi = torch.where(inside, fy, torch.zeros_like(fy)).to(torch.int64)
        idx = (yi * w1 + xi).reshape(b, 1, -1).expand(b, c, -1)
        val = torch.gather(flat, 2, idx).reshape(out.shape)
        wt0 = torch.where(inside, wt, torch.zeros_like(wt)).unsqueeze(1)
        out.add_(val * wt0)

    tap(fx0, fy0, w_nw)
    tap(fx1, fy0, w_ne)
    tap(fx0, fy1, w_sw)
    tap(fx1, fy1, w_se)
    return out


def differentiable_warping(src_fea: Tensor, src_proj: Tensor, ref_proj: Tensor,
                           depth_samples: Tensor, return_mask: bool = False):
    """module.py:68-125 (same signature).  Gradient flows to ``src_fea`` only."""
    h1, w1 = src_fea.shape[2:]
    with torch.no_grad():
        proj = compose_projection(src_proj, ref_proj)
        ix, iy, valid = warp_source_coords(proj, depth_samples, h1, w1)
    warped = bilinear_gather(src_fea, ix, iy)
    return (warped, valid) if return_mask else warped


def group_correlation(warped: Tensor, ref: Tensor, groups: int = GROUPS) -> Tensor:
    """itermvs.py:50-51 / :103-104: mean over each contiguous channel block of
    warped[B,C,N,H,W] * ref[B,C,H,W] -> [B,G,N,H,W]."""
    b, c, n, h, w = warped.shape
    prod = warped.view(b, groups, c // groups, n, h, w) * ref.view(b, groups, c // groups, 1, h, w)
    return prod.mean(dim=2)


# --------------------------------------------------------------------------
# small conv stacks (weights looked up by state_dict name)
# --------------------------------------------------------------------------
def _n_first(x: Tensor) -> Tensor:
    """[B,G,N,H,W] -> [B*N,G,H,W] (itermvs.py:343-345, 367-369)."""
    b, g, n, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b * n, g, h, w)


def pixel_view_weight(wts: Weights, corr: Tensor,
                      prefix: str = "iter_mvs.evaluation.pixel_view_weight.") -> Tensor:
    """itermvs.py:333-350: [B,G,N,H,W] -> [B,1,H,W]."""
    b, g, n, h, w = corr.shape
    x = _n_first(corr)
    x = F.relu(F.conv2d(x, wts[prefix + "conv.0.conv.weight"], padding=1))
    x = F.conv2d(x, wts[prefix + "conv.1.weight"], wts[prefix + "conv.1.bias"])
    x = torch.softmax(x.view(b, n, h, w), dim=1)
    return x.max(dim=1, keepdim=True)[0]


def corr_net(wts: Weights, corr: Tensor, level: int) -> Tensor:
    """itermvs.py:352-381: tiny U-Net, [B,G,N,H,W] -> [B,N,H,W].
    ``level`` in 1..3 selects ``corr_conv1.{level-1}``."""
    p = f"iter_mvs.evaluation.corr_conv1.{level - 1}."
    b, g, n, h, w = corr.shape
    x = _n_first(corr)
    c0 = F.relu(F.conv2d(x, wts[p + "conv0.conv.weight"], padding=1))
    c1 = F.relu(F.conv2d(c0, wts[p + "conv1.conv.weight"], stride=2, padding=1))
    c2 = F.relu(F.conv2d(c1, wts[p + "conv2.conv.weight"], stride=2, padding=1))
    u1 = c1 + F.conv_transpose2d(c2, wts[p + "conv3.weight"], stride=2, padding=1, output_padding=1)
    u0 = c0 + F.conv_transpose2d(u1, wts[p + "conv4.weight"], stride=2, padding=1, output_padding=1)
    y = F.conv2d(u0, wts[p + "conv5.weight"], wts[p + "conv5.bias"], padding=1)
    return y.view(b, n, h, w)


# --------------------------------------------------------------------------
# a2: Evaluation, init branch                          itermvs.py:36-82
# --------------------------------------------------------------------------
def evaluation_init(wts: Weights, ref_l3: Tensor, src_l3: Sequence[Tensor],
                    ref_proj_l3: Tensor, src_projs_l3: Sequence[Tensor],
                    depth_sample: Tensor, inv_min: Tensor, inv_max: Tensor
                    ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Returns (view_weights [B,S,2H,2W], score [B,N,H,W], depth [B,1,2H,2W],
    aggregated correlation before CorrNet [B,G,N,H,W])."""
    n = depth_sample.shape[1]
    acc = 0
    wsum = 1e-5
    weights_up: List[Tensor] = []
    for fea, proj in zip(src_l3, src_projs_l3):
        warped = differentiable_warping(fea, proj, ref_proj_l3, depth_sample)
        corr = group_correlation(warped, ref_l3)
        vw = pixel_view_weight(wts, corr)                       # [B,1,H,W]
        weights_up.append(resize_bilinear(vw, 2.0))             # :56-57
        acc = acc + corr * vw.unsqueeze(1)
        wsum = wsum + vw.unsqueeze(1)
    agg = acc / wsum
    score = corr_net(wts, agg, 3)                               # :70 (corr_conv1[-1])
    view_weights = torch.cat(weights_up, dim=1)
    prob = torch.softmax(score, dim=1)
    k = torch.arange(n, dtype=torch.float32, device=score.device).view(1, n, 1, 1)
    idx = (k * prob).sum(dim=1, keepdim=True)
    depth = depth_unnormalization(idx / (n - 1.0), inv_min, inv_max)
    depth = resize_bilinear(depth, 2.0)                         # :80-81
    return view_weights, score, depth, agg


# --------------------------------------------------------------------------
# a3: Evaluation, iteration branch                     itermvs.py:84-126
# --------------------------------------------------------------------------
def ref_feature_quarter(ref_feature: Dict[int, Tensor]) -> Dict[int, Tensor]:
    """itermvs.py:95-98: reference features resampled onto the 1/4 grid
    (level 1: x0.5 == 2x2 box mean, level 2: as is, level 3: x2 bilinear)."""
    return {1: resize_bilinear(ref_feature[1], 0.5), 2: ref_feature[2],
            3: resize_bilinear(ref_feature[3], 2.0)}


def evaluation_iter(wts: Weights, ref_feature: Dict[int, Tensor],
                    src_features: Dict[int, Sequence[Tensor]],
                    ref_proj: Dict[int, Tensor], src_projs: Dict[int, Sequence[Tensor]],
                    depth_sample: Dict[int, Tensor], view_weights: Tensor,
                    return_aggregates: bool = False):
    """Returns scores [B,10,H,W]; optionally also the per-level aggregated
    group correlations [B,G,N_l,H,W] fed to CorrNet."""
    ref_q = ref_feature_quarter(ref_feature)
    scores, aggs = [], []
    for lvl in (1, 2, 3):
        acc = 0
        wsum = 1e-5
        for i, (fea, proj) in enumerate(zip(src_features[lvl], src_projs[lvl])):
            warped = differentiable_warping(fea, proj, ref_proj[lvl], depth_sample[lvl])
            corr = group_correlation(warped, ref_q[lvl])
            b, _, _, h, w = corr.shape
            vw = view_weights[:, i].view(b, 1, 1, h, w)
            acc = acc + corr * vw
            wsum = wsum + vw
        agg = acc / wsum
        aggs.append(agg)
        scores.append(corr_net(wts, agg, lvl))
    out = torch.cat(scores, dim=1)
    return (out, aggs) if return_aggregates else out


def iteration_depth_samples(nd: Tensor, inv_min: Tensor, inv_max: Tensor) -> Dict[int, Tensor]:
    """itermvs.py:290-293: hypotheses around the current normalised depth."""
    out = {}
    for lvl in (1, 2, 3):
        off = torch.tensor(CORR_INTERVAL[lvl], dtype=torch.float32, device=nd.device).view(1, -1, 1, 1) * INTERVAL_SCALE
        out[lvl] = depth_unnormalization(torch.clamp(nd + off, min=0, max=1), inv_min, inv_max)
    return out


# --------------------------------------------------------------------------
# a6/a7: Update (ConvGRU + heads + window regression)  itermvs.py:129-220
# --------------------------------------------------------------------------
def conv_gru(wts: Weights, h: Tensor, x: Tensor, prefix: str = "iter_mvs.update.gru.") -> Tensor:
    """module.py:52-66: 3x3 dilation-2 gates."""
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(F.conv2d(hx, wts[prefix + "convz.weight"], wts[prefix + "convz.bias"], padding=2, dilation=2))
    r = torch.sigmoid(F.conv2d(hx, wts[prefix + "convr.weight"], wts[prefix + "convr.bias"], padding=2, dilation=2))
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], dim=1), wts[prefix + "convq.weight"],
                            wts[prefix + "convq.bias"], padding=2, dilation=2))
    return (1 - z) * h + z * q


def depth_head_logits(wts: Weights, hidden: Tensor, prefix: str = "iter_mvs.update.depth_head.") -> Tensor:
    """itermvs.py:139-145: 3x3 dil-2 32->32, 1x1 32->64, 1x1 64->256."""
    x = F.relu(F.conv2d(hidden, wts[prefix + "0.weight"], padding=2, dilation=2))
    x = F.relu(F.conv2d(x, wts[prefix + "2.weight"]))
    return F.conv2d(x, wts[prefix + "4.weight"], wts[prefix + "4.bias"])


def confidence_logit(wts: Weights, hidden: Tensor, prefix: str = "iter_mvs.update.confidence_head.") -> Tensor:
    """itermvs.py:147-151."""
    x = F.relu(F.conv2d(hidden, wts[prefix + "0.weight"], padding=2, dilation=2))
    return F.conv2d(x, wts[prefix + "2.weight"], wts[prefix + "2.bias"])


def window_regression(prob: Tensor, radius: int = WINDOW_RADIUS) -> Tuple[Tensor, Tensor]:
    """itermvs.py:173-189 / 203-219: first-max argmax, clamped +-radius window
    (border duplicates double-counted), expectation / (1e-6 + mass) / (bins-1).
    prob [B,K,H,W] -> (normalised depth [B,1,H,W], argmax index int64 [B,1,H,W])."""
    bins = prob.shape[1]
    with torch.no_grad():
        best = torch.argmax(prob, dim=1, keepdim=True)
        offs = torch.arange(-radius, radius + 1, device=prob.device).view(1, -1, 1, 1)
        win = torch.clamp(best + offs, 0, bins - 1)             # int64 [B,9,H,W]
    num = 0
    den = 1e-6
    for i in range(2 * radius + 1):
        pi = torch.gather(prob, 1, win[:, i:i + 1])
        num = num + win[:, i:i + 1] * pi
        den = den + pi
    return (num / den) / (bins - 1.0), best


def hidden_init(wts: Weights, score: Tensor, prefix: str = "iter_mvs.update.hidden_init_head.") -> Tensor:
    """itermvs.py:159-164."""
    x = F.relu(F.conv2d(score, wts[prefix + "0.weight"], padding=1))
    x = F.conv2d(x, wts[prefix + "2.weight"], wts[prefix + "2.bias"])
    return torch.tanh(resize_bilinear(x, 2.0))


def depth_init(wts: Weights, hidden: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """itermvs.py:171-190 -> (normalised depth, probability, argmax index)."""
    prob = torch.softmax(depth_head_logits(wts, hidden), dim=1)
    nd, best = window_regression(prob)
    return nd, prob, best


def update_step(wts: Weights, hidden: Tensor, nd: Tensor, score: Tensor, want_conf: bool):
    """itermvs.py:192-220 -> (hidden, nd, prob, conf (sigmoid) | None,
    conf logit | None, argmax index)."""
    hidden = conv_gru(wts, hidden, torch.cat([nd, score], dim=1))
    conf0 = conf = None
    if want_conf:
        conf0 = confidence_logit(wts, hidden)
        conf = torch.sigmoid(conf0)
    prob = torch.softmax(depth_head_logits(wts, hidden), dim=1)
    nd_new, best = window_regression(prob)
    return hidden, nd_new, prob, conf, conf0, best


# --------------------------------------------------------------------------
# a10: convex upsampling                                 module.py:127-140
# --------------------------------------------------------------------------
def convex_upsample(x: Tensor, weight: Tensor, scale: int = 4) -> Tensor:
    """x [B,1,H,W]; weight [B,1,9,s,s,H,W] already soft-maxed over dim 2.
    out[b,0,s*y+i,s*x+j] = sum_k weight[b,0,k,i,j,y,x] * x_pad[y+k//3-1, x+k%3-1]
    with replicate padding (restates ReplicationPad2d + F.unfold)."""
    b, _, h, w = x.shape
    yy = torch.arange(h, device=x.device).view(h, 1)
    xx = torch.arange(w, device=x.device).view(1, w)
    out = torch.zeros(b, 1, scale, scale, h, w, dtype=x.dtype, device=x.device)
    for k in range(9):
        ny = torch.clamp(yy + (k // 3 - 1), 0, h - 1)
        nx = torch.clamp(xx + (k % 3 - 1), 0, w - 1)
        out = out + weight[:, :, k] * x[:, :, ny, nx].view(b, 1, 1, 1, h, w)
    return out.permute(0, 1, 4, 2, 5, 3).reshape(b, 1, scale * h, scale * w)


def upsample_weights(wts: Weights, ref_l2: Tensor, prefix: str = "iter_mvs.upsample.") -> Tensor:
    """itermvs.py:262-264 -> [B,1,9,4,4,H,W] (softmax over the 9 taps)."""
    b, _, h, w = ref_l2.shape
    x = F.relu(F.conv2d(ref_l2, wts[prefix + "0.weight"], padding=1))
    x = F.conv2d(x, wts[prefix + "2.weight"])
    return torch.softmax(x.view(b, 1, 9, 4, 4, h, w), dim=2)


# --------------------------------------------------------------------------
# a11: FeatureNet                                          net.py:7-66
# --------------------------------------------------------------------------
def _bn(wts: Weights, p: str, x: Tensor, training: bool) -> Tensor:
    return F.batch_norm(x, wts[p + "running_mean"], wts[p + "running_var"], wts[p + "weight"],
                        wts[p + "bias"], training=training, momentum=0.1, eps=1e-5)


def _cbr(wts: Weights, p: str, x: Tensor, stride: int, relu: bool, training: bool) -> Tensor:
    y = _bn(wts, p + "bn.", F.conv2d(x, wts[p + "conv.weight"], stride=stride, padding=1), training)
    return F.relu(y) if relu else y


def _res_block(wts: Weights, p: str, x: Tensor, stride: int, training: bool) -> Tensor:
    """module.py:33-50."""
    y = _cbr(wts, p + "conv1.", x, stride, True, training)
    y = _cbr(wts, p + "conv2.", y, 1, False, training)
    if stride != 1:
        x = _cbr(wts, p + "downsample.", x, stride, False, training)
    return F.relu(x + y)


def feature_net(wts: Weights, images: Tensor, training: bool = False) -> Dict[int, Tensor]:
    """net.py:36-65.  images [M,3,H,W] (M = B*V) -> {1:[M,16,H/2,W/2],
    2:[M,32,H/4,W/4], 3:[M,48,H/8,W/8]}.  With ``training`` the batch-norm uses
    batch statistics over all M images (net.py:38-51); in eval the per-view
    python loop of net.py:56-65 is equivalent to batching."""
    p = "feature_net."
    f0 = _cbr(wts, p + "conv1.", images, 1, True, training)
    f1 = _res_block(wts, p + "layer1.1.", _res_block(wts, p + "layer1.0.", f0, 2, training), 1, training)
    f2 = _res_block(wts, p + "layer2.1.", _res_block(wts, p + "layer2.0.", f1, 2, training), 1, training)
    f3 = _res_block(wts, p + "layer3.1.", _res_block(wts, p + "layer3.0.", f2, 2, training), 1, training)
    out3 = F.conv2d(f3, wts[p + "output3.weight"], wts[p + "output3.bias"], padding=1)
    mid = resize_bilinear(f3, 2.0) + F.conv2d(f2, wts[p + "inner2.weight"], wts[p + "inner2.bias"])
    out2 = F.conv2d(mid, wts[p + "output2.weight"], wts[p + "output2.bias"], padding=1)
    mid = resize_bilinear(mid, 2.0) + F.conv2d(f1, wts[p + "inner1.weight"], wts[p + "inner1.bias"])
    out1 = F.conv2d(mid, wts[p + "output1.weight"], wts[p + "output1.bias"], padding=1)
    return {1: out1, 2: out2, 3: out3}


# --------------------------------------------------------------------------
# a8 + a12: IterMVS driver and Pipeline          itermvs.py:253-329, net.py:78-128
# --------------------------------------------------------------------------
def pipeline_forward(wts: Weights, imgs: Dict[str, Tensor], proj_matrices: Dict[str, Tensor],
                     depth_min: Tensor, depth_max: Tensor, iteration: int = 4,
                     test: bool = True, training: bool = False, trace: Optional[dict] = None,
                     feature_storage: Optional[torch.dtype] = None):
    """net.py:78-128.  ``imgs['level_0']`` [B,V,3,H,W]; ``proj_matrices['level_l']``
    [B,V,4,4] for l=1..3.  Returns the same dict as the reference (test: keys
    depths_upsampled / confidence_upsampled; train: depths{combine,probability,
    initial}, depths_upsampled[list], confidences[list], confidence_upsampled).
    ``trace`` (optional dict) receives intermediate tensors for kernel tests.
    ``feature_storage`` (torch.bfloat16 / torch.float16): model of the build's 16-bit feature STORAGE (BASELINE cfg 4 /
    cfg 5; not a reference feature): the three pyramids are rounded to that type (round to nearest even) before the
    matching stages read them, arithmetic stays fp32, the rounding is straight-through for autograd, and the
    up-sampling head keeps reading the unrounded level-2 features -- what itermvs_amd does with ``feature_dtype``."""
    x = imgs["level_0"]
    b, v, _, hh, ww = x.shape
    feats = feature_net(wts, x.reshape(b * v, 3, hh, ww), training)
    ref2_unrounded = feats[2].view(b, v, *feats[2].shape[1:])[:, 0]
    gathered = feats
    if feature_storage is not None:
        gathered = {l: f + (f.to(feature_storage).float() - f).detach() for l, f in feats.items()}
    per_view = {l: f.view(b, v, *f.shape[1:]) for l, f in gathered.items()}
    ref_f = {l: per_view[l][:, 0] for l in (1, 2, 3)}
    src_f = {l: [per_view[l][:, i] for i in range(1, v)] for l in (1, 2, 3)}
    projs = {l: proj_matrices[f"level_{l}"].float() for l in (1, 2, 3)}
    ref_p = {l: projs[l][:, 0] for l in (1, 2, 3)}
    src_p = {l: [projs[l][:, i] for i in range(1, v)] for l in (1, 2, 3)}
    depth_min = depth_min.float()
    depth_max = depth_max.float()

    h, w = ref_f[2].shape[2:]
    up_w = upsample_weights(wts, ref2_unrounded)
    inv_min = (1.0 / depth_min).view(b, 1, 1, 1)
    inv_max = (1.0 / depth_max).view(b, 1, 1, 1)

    samples0 = initial_depth_samples(inv_min, inv_max, h // 2, w // 2)
    view_w, score, depth0, agg0 = evaluation_init(wts, ref_f[3], src_f[3], ref_p[3], src_p[3],
                                                  samples0, inv_min, inv_max)
    hidden = hidden_init(wts, score)
    nd, prob, best = depth_init(wts, hidden)
    if trace is not None:
        trace.update(feats=feats, feats_gathered=gathered, up_w=up_w, view_weights=view_w, init_score=score, init_agg=agg0,
                     hidden0=hidden, nd0=nd, best0=best, iters=[])

    depths = {"combine": [], "probability": [], "initial": []}
    confidences: List[Tensor] = []
    depths_up: List[Tensor] = []
    conf_up = None
    if not test:
        depths["initial"].append(depth0)
        conf0 = confidence_logit(wts, hidden)
        depths["combine"].append(depth_unnormalization(nd, inv_min, inv_max))
        depths["probability"].append(prob)
        confidences.append(conf0)
        nd = nd.detach()

    depth_lo = conf = depth_hi = None
    for it in range(iteration):
        samples = iteration_depth_samples(nd, inv_min, inv_max)
        score, aggs = evaluation_iter(wts, ref_f, src_f, ref_p, src_p, samples, view_w.detach(),
                                      return_aggregates=True)
        last = it == iteration - 1
        if test and last:
            depth_lo = depth_unnormalization(nd, inv_min, inv_max)          # itermvs.py:319
        nd_in = nd
        hidden, nd, prob, conf, conf_logit, best = update_step(wts, hidden, nd, score, want_conf=(not test) or last)
        if trace is not None:
            trace["iters"].append(dict(nd_in=nd_in, samples=samples, aggs=aggs, score=score,
                                       hidden=hidden, nd=nd, best=best, conf=conf))
        if not test:
            depths["combine"].append(depth_unnormalization(nd, inv_min, inv_max))
            depths["probability"].append(prob)
            confidences.append(conf_logit)
        if last:
            depth_hi = depth_unnormalization(convex_upsample(nd, up_w), inv_min, inv_max)
            conf_up = resize_bilinear(conf, 4.0)
            if not test:
                depths_up.append(depth_hi)
        if not test:
            nd = nd.detach()

    if test:
        if trace is not None:
            trace.update(depth=depth_lo, confidence=conf)
        return {"depths_upsampled": depth_hi, "confidence_upsampled": conf_up}
    return {"depths": depths, "depths_upsampled": depths_up, "confidences": confidences,
            "confidence_upsampled": conf_up}


# --------------------------------------------------------------------------
# a13: training loss                                      net.py:131-190
# --------------------------------------------------------------------------
def full_loss(depths, depths_upsampled, confidences, depths_gt, mask, depth_min, depth_max,
              regress: bool = True) -> Tensor:
    """net.py:131-190 (same signature)."""
    radius, bins = WINDOW_RADIUS, PROB_BINS
    probs = depths["probability"]
    k = probs[0].size(1)
    m0 = mask["level_0"] > 0.5
    m2 = mask["level_2"] > 0.5
    gt0 = depths_gt["level_0"]
    gt2 = depths_gt["level_2"]
    b = gt2.shape[0]
    inv_min = (1.0 / depth_min).view(b, 1, 1, 1)
    inv_max = (1.0 / depth_max).view(b, 1, 1, 1)
    ngt = depth_normalization(gt2, inv_min, inv_max)
    gt_bin = torch.floor(torch.clamp(ngt, 0, 1) * (k - 1) * m2.float()).long()
    onehot = torch.zeros_like(probs[0]).scatter_(1, gt_bin, 1)

    npred = len(depths["combine"])
    nd = depth_normalization(depths["initial"][0], inv_min, inv_max)
    loss = (0.8 ** npred) * bins * F.l1_loss(nd[m2], ngt[m2], reduction="mean")
    for it in range(npred):
        coff = 0.8 ** (npred - it - 1)
        p = torch.clamp(probs[it], min=1e-5)
        ce = -(onehot * torch.log(p)).sum(dim=1, keepdim=True)
        loss = loss + coff * ce[m2].mean()
        if regress:
            with torch.no_grad():
                best = torch.argmax(p, dim=1, keepdim=True).float()
                near = (gt_bin >= best - radius) & (gt_bin <= best + radius)
            nd = depth_normalization(depths["combine"][it], inv_min, inv_max)
            sel = m2 & near
            if sel.sum() > 0:
                loss = loss + coff * bins * F.l1_loss(nd[sel], ngt[sel], reduction="mean")
            cgt = (torch.abs(nd[m2].detach() - ngt[m2]) < 0.002).float()
            loss = loss + coff * F.binary_cross_entropy_with_logits(confidences[it][m2], cgt)
    ngt0 = depth_normalization(gt0, inv_min, inv_max)
    nd0 = depth_normalization(depths_upsampled[0], inv_min, inv_max)
    return loss + bins * F.l1_loss(nd0[m0], ngt0[m0], reduction="mean")
