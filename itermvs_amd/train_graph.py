"""Differentiable (training) form of the pipeline and ``full_loss`` -- models/net.py:36-51,
78-190 and the ``not self.test`` branches of models/itermvs.py:253-329.

Training needs gradients w.r.t. the feature maps and every weight, so the dense layers keep the
reference's tensor-level structure on PyTorch-ROCm autograd.  The matching part -- homography warp,
bilinear gather, group-wise correlation and (iteration branch) the view-weighted mean, i.e. each
``Evaluation`` call up to its CorrNet -- is ONE ``torch.autograd.Function`` per call backed by the fused
HIP kernels: forward ``itermvs_corr_init`` / ``itermvs_corr_iter`` (the inference kernels), backward
``itermvs_corr_init_backward`` / ``itermvs_corr_iter_backward`` (warped values recomputed, gradient
scatter-added into the source features with fp32 atomics, gathered into the reference features).  No
``[B,C,N,H,W]`` warped volume and no per-view cost volume of the iteration branch exists in training
either.  Like the reference (module.py:77, itermvs.py:295) no gradient flows to depth hypotheses,
cameras or (iteration branch) view weights.  Everything runs on the GPU; CPU tensors are rejected
upstream (net.py counterpart).
"""
from __future__ import annotations

from typing import Dict, List, Mapping

import torch
import torch.nn.functional as F

from . import ops
from .engine import HIDDEN, INIT_SAMPLES, sample_offsets

Tensor = torch.Tensor
G = 8
BINS = 256
RADIUS = 4


def _unnorm(nd: Tensor, inv_min: Tensor, inv_max: Tensor) -> Tensor:        # module.py:148-152
    return 1.0 / (inv_max + nd * (inv_min - inv_max))


def _norm(depth: Tensor, inv_min: Tensor, inv_max: Tensor) -> Tensor:       # module.py:142-146
    return (1.0 / (depth + 1e-5) - inv_max) / (inv_min - inv_max)


class _Net:
    """Stateless functional view over the weight dict (state_dict names)."""

    def __init__(self, w: Mapping[str, Tensor], bn_training: bool):
        self.w = w
        self.bn_training = bn_training

    # -- FeatureNet (net.py:36-51, batch of all B*V views, BatchNorm over the whole batch) -----
    def cbr(self, x, p, stride=1, relu=True):
        w = self.w
        y = F.conv2d(x, w[p + "conv.weight"], stride=stride, padding=1)
        if self.bn_training:        # batch statistics + ReLU in one HIP op (csrc/bn.hip), running stats updated in place
            y = ops.bn_relu_train(y, w[p + "bn.weight"], w[p + "bn.bias"], w[p + "bn.running_mean"], w[p + "bn.running_var"],
                                  eps=1e-5, momentum=0.1, relu=relu)
            w[p + "bn.num_batches_tracked"].add_(1)
            return y
        y = F.batch_norm(y, w[p + "bn.running_mean"], w[p + "bn.running_var"], w[p + "bn.weight"], w[p + "bn.bias"],
                         training=False, momentum=0.1, eps=1e-5)
        return F.relu(y) if relu else y

    def block(self, x, p, stride):
        y = self.cbr(self.cbr(x, p + "conv1.", stride), p + "conv2.", 1, relu=False)
        if stride != 1:
            x = self.cbr(x, p + "downsample.", stride, relu=False)
        return F.relu(x + y)

    def features(self, x):
        w, p = self.w, "feature_net."
        f = self.cbr(x, p + "conv1.")
        fs = []
        for l in (1, 2, 3):
            f = self.block(self.block(f, f"{p}layer{l}.0.", 2), f"{p}layer{l}.1.", 1)
            fs.append(f)
        f1, f2, f3 = fs
        up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear")
        o3 = F.conv2d(f3, w[p + "output3.weight"], w[p + "output3.bias"], padding=1)
        mid = up(f3) + F.conv2d(f2, w[p + "inner2.weight"], w[p + "inner2.bias"])
        o2 = F.conv2d(mid, w[p + "output2.weight"], w[p + "output2.bias"], padding=1)
        mid = up(mid) + F.conv2d(f1, w[p + "inner1.weight"], w[p + "inner1.bias"])
        o1 = F.conv2d(mid, w[p + "output1.weight"], w[p + "output1.bias"], padding=1)
        return {1: o1, 2: o2, 3: o3}

    # -- Evaluation pieces (itermvs.py:333-381); correlations arrive as [B,N,G,H,W], the layout the fused kernels
    #    write and the reference reaches with its permute (itermvs.py:343-345, 367-369) --------------------------
    def view_weight(self, corr):
        w, p = self.w, "iter_mvs.evaluation.pixel_view_weight."
        b, n, g, h, wd = corr.shape
        x = F.relu(F.conv2d(corr.reshape(b * n, g, h, wd), w[p + "conv.0.conv.weight"], padding=1))
        x = F.conv2d(x, w[p + "conv.1.weight"], w[p + "conv.1.bias"]).view(b, n, h, wd)
        return torch.softmax(x, dim=1).max(dim=1, keepdim=True)[0]

    def corr_net(self, corr, level):
        w, p = self.w, f"iter_mvs.evaluation.corr_conv1.{level - 1}."
        b, n, g, h, wd = corr.shape
        c0 = F.relu(F.conv2d(corr.reshape(b * n, g, h, wd), w[p + "conv0.conv.weight"], padding=1))
        c1 = F.relu(F.conv2d(c0, w[p + "conv1.conv.weight"], stride=2, padding=1))
        c2 = F.relu(F.conv2d(c1, w[p + "conv2.conv.weight"], stride=2, padding=1))
        u1 = c1 + F.conv_transpose2d(c2, w[p + "conv3.weight"], stride=2, padding=1, output_padding=1)
        u0 = c0 + F.conv_transpose2d(u1, w[p + "conv4.weight"], stride=2, padding=1, output_padding=1)
        return F.conv2d(u0, w[p + "conv5.weight"], w[p + "conv5.bias"], padding=1).view(b, n, h, wd)

    # -- Update pieces (itermvs.py:129-220, module.py:52-66) ------------------------------------
    def gru(self, h, x):
        w, p = self.w, "iter_mvs.update.gru."
        hx = torch.cat([h, x], 1)
        conv = lambda t, n: F.conv2d(t, w[p + n + ".weight"], w[p + n + ".bias"], padding=2, dilation=2)
        z = torch.sigmoid(conv(hx, "convz"))
        r = torch.sigmoid(conv(hx, "convr"))
        q = torch.tanh(conv(torch.cat([r * h, x], 1), "convq"))
        return (1 - z) * h + z * q

    def probability(self, hidden):
        w, p = self.w, "iter_mvs.update.depth_head."
        x = F.relu(F.conv2d(hidden, w[p + "0.weight"], padding=2, dilation=2))
        x = F.relu(F.conv2d(x, w[p + "2.weight"]))
        return torch.softmax(F.conv2d(x, w[p + "4.weight"], w[p + "4.bias"]), dim=1)

    def conf_logit(self, hidden):
        w, p = self.w, "iter_mvs.update.confidence_head."
        x = F.relu(F.conv2d(hidden, w[p + "0.weight"], padding=2, dilation=2))
        return F.conv2d(x, w[p + "2.weight"], w[p + "2.bias"])

    def hidden_init(self, score):
        w, p = self.w, "iter_mvs.update.hidden_init_head."
        x = F.conv2d(F.relu(F.conv2d(score, w[p + "0.weight"], padding=1)), w[p + "2.weight"], w[p + "2.bias"])
        return torch.tanh(F.interpolate(x, scale_factor=2, mode="bilinear"))


def _regress(prob: Tensor) -> Tensor:
    """itermvs.py:173-189: differentiable through the gathered probabilities only."""
    with torch.no_grad():
        best = torch.argmax(prob, dim=1, keepdim=True)
        win = torch.clamp(best + torch.arange(-RADIUS, RADIUS + 1, device=prob.device).view(1, -1, 1, 1), 0, BINS - 1)
    picked = torch.gather(prob, 1, win)                       # [B,9,H,W]; duplicates at the borders are kept
    num = 0
    den = 1e-6
    for i in range(2 * RADIUS + 1):
        num = num + win[:, i:i + 1] * picked[:, i:i + 1]
        den = den + picked[:, i:i + 1]
    return (num / den) / (BINS - 1.0)


def _convex_upsample(nd: Tensor, weight: Tensor) -> Tensor:
    """module.py:127-140 with autograd (weight [B,1,9,4,4,H,W] soft-maxed)."""
    b, _, h, w = nd.shape
    patches = F.unfold(F.pad(nd, (1, 1, 1, 1), mode="replicate"), [3, 3]).view(b, 1, 9, 1, 1, h, w)
    up = (patches * weight).sum(dim=2)
    return up.permute(0, 1, 4, 2, 5, 3).reshape(b, 1, 4 * h, 4 * w)


def train_forward(w: Mapping[str, Tensor], imgs: Tensor, projs: Dict[int, Tensor], depth_min: Tensor,
                  depth_max: Tensor, iteration: int, bn_training: bool = True, feature_dtype: str = "fp32",
                  nan_flag: Tensor = None):
    """``Pipeline(test=False).forward``: returns the reference's training dict (net.py:115-120).
    ``feature_dtype`` "bf16" / "fp16" (BASELINE cfg 4: bf16): the three pyramids are STORED in 16 bits for the fused
    correlation kernels (forward and backward gather half the bytes, fp32 arithmetic), exactly like test mode's
    ``Pipeline.feature_dtype``; the rounding is straight-through for autograd (fp32 gradients, fp32 weights).
    ``nan_flag`` (device int32[1]): the deferred form of the reference's NaN asserts on the composed projections
    (module.py:83,87) -- OR-ed with 1 instead of being read back and asserted here, so that the forward issues no
    device-to-host synchronisation and can be captured (train_step.CapturedTrainStep checks the flag after its replays)."""
    deferred_nan = nan_flag is not None
    if feature_dtype not in ops.FEATURE_DTYPES:
        raise ValueError(f"feature_dtype must be one of {sorted(ops.FEATURE_DTYPES)}, got {feature_dtype!r}")
    store_dt = ops.FEATURE_DTYPES[feature_dtype]
    net = _Net(w, bn_training)
    b, v, _, hh, ww = imgs.shape
    s = v - 1
    feats = net.features(imgs.reshape(b * v, 3, hh, ww))
    # channels-last copies of the pyramid: what the fused correlation kernels gather from (autograd sees a layout change)
    cl = {l: f.contiguous(memory_format=torch.channels_last) for l, f in feats.items()}
    stored = None
    if store_dt != torch.float32:
        stored = {l: cl[l].detach().to(store_dt) for l in cl}                          # what the kernels read
        cl = {l: cl[l] + (stored[l].float() - cl[l]).detach() for l in cl}             # same values for the torch side
    # every Evaluation call below sends a gradient to these three tensors: ONE shared zero-filled accumulator per level
    # (ops.FeatureGradPool) instead of one dense tensor per call summed by autograd
    pool = ops.FeatureGradPool()
    cl = ops.feature_grad_sink(pool, cl)
    ref = {l: cl[l].view(b, v, *cl[l].shape[1:])[:, 0] for l in (1, 2, 3)}
    if nan_flag is None:
        nan_flag = torch.zeros((1,), device=imgs.device, dtype=torch.int32)
    with torch.no_grad():
        proj = ops.compose_proj(torch.stack([projs[1], projs[2], projs[3]]).reshape(3 * b, v, 4, 4), nan_flag).view(3, b, s, 12)
    h, wd = feats[2].shape[2:]
    inv_min_b, inv_max_b = (1.0 / depth_min).contiguous(), (1.0 / depth_max).contiguous()
    inv_min, inv_max = inv_min_b.view(b, 1, 1, 1), inv_max_b.view(b, 1, 1, 1)

    u = "iter_mvs.upsample."
    # (the up-sampling head reads the UNROUNDED level-2 reference features, like test mode's planar fp32 copy)
    ref2_fp32 = feats[2].view(b, v, *feats[2].shape[1:])[:, 0]
    up_w = F.conv2d(F.relu(F.conv2d(ref2_fp32, w[u + "0.weight"], padding=1)), w[u + "2.weight"])
    up_w = torch.softmax(up_w.view(b, 1, 9, 4, 4, h, wd), dim=2)

    # ---- initialisation: itermvs.py:36-82 ------------------------------------------------------
    k = torch.arange(INIT_SAMPLES, device=imgs.device, dtype=torch.float32).view(1, -1, 1, 1)
    corr_views = ops.corr_init_train(cl[3], b, v, proj[2], inv_min_b, inv_max_b, INIT_SAMPLES,
                                     stored=None if stored is None else stored[3], pool=pool)      # [B,S,N,8,h3,w3]
    acc, wsum, vws = 0, 1e-5, []
    for i in range(s):
        corr = corr_views[:, i]                                                                 # [B,N,8,h3,w3]
        vw = net.view_weight(corr)
        vws.append(F.interpolate(vw, scale_factor=2, mode="bilinear"))
        acc = acc + corr * vw.unsqueeze(1)
        wsum = wsum + vw.unsqueeze(1)
    score = net.corr_net(acc / wsum, 3)
    view_w = torch.cat(vws, 1)
    prob0 = torch.softmax(score, dim=1)
    depth_init = _unnorm((k * prob0).sum(1, keepdim=True) / (INIT_SAMPLES - 1.0), inv_min, inv_max)
    depth_init = F.interpolate(depth_init, scale_factor=2, mode="bilinear")

    hidden = net.hidden_init(score)
    prob = net.probability(hidden)
    nd = _regress(prob)
    depths = {"combine": [_unnorm(nd, inv_min, inv_max)], "probability": [prob], "initial": [depth_init]}
    confidences: List[Tensor] = [net.conf_logit(hidden)]
    depths_up: List[Tensor] = []
    conf_up = None
    nd = nd.detach()
    # itermvs.py:95-98: reference features on the 1/4 grid, packed [B,H,W,96] like itermvs_ref_quarter does at inference
    ref_q = torch.cat([F.interpolate(ref[1], scale_factor=0.5, mode="bilinear"), ref[2],
                       F.interpolate(ref[3], scale_factor=2, mode="bilinear")], 1).permute(0, 2, 3, 1).contiguous()
    offsets = sample_offsets()
    view_w_const = view_w.detach()                                                       # itermvs.py:295

    # ---- iterations: itermvs.py:288-314 ----------------------------------------------------------
    for it in range(iteration):
        # hypotheses clamp(nd + offsets) -> depth (itermvs.py:290-293) are built inside the kernel, as at inference
        aggs = ops.corr_iter_train(cl, b, v, ref_q, proj, view_w_const, inv_min_b, inv_max_b, nd, offsets, stored=stored, pool=pool)
        scores = [net.corr_net(aggs[i], l) for i, l in enumerate((1, 2, 3))]
        hidden = net.gru(hidden, torch.cat([nd] + scores, 1))
        conf0 = net.conf_logit(hidden)
        prob = net.probability(hidden)
        nd = _regress(prob)
        depths["combine"].append(_unnorm(nd, inv_min, inv_max))
        depths["probability"].append(prob)
        confidences.append(conf0)
        if it == iteration - 1:
            depths_up.append(_unnorm(_convex_upsample(nd, up_w), inv_min, inv_max))
            conf_up = F.interpolate(torch.sigmoid(conf0), scale_factor=4, mode="bilinear")
        nd = nd.detach()
    # module.py:83,87: the reference asserts inside every warp; here once per forward, after everything is enqueued
    if not deferred_nan:
        assert int(nan_flag.item()) == 0, "nan in proj (singular or non-finite camera matrix, module.py:83,87)"
    return {"depths": depths, "depths_upsampled": depths_up, "confidences": confidences,
            "confidence_upsampled": conf_up}


def _masked_mean(values: Tensor, m: Tensor, empty_is_zero: bool = False) -> Tensor:
    """mean of ``values`` over the True entries of ``m`` without boolean indexing: fixed shapes, no device-to-host
    synchronisation (``x[mask]`` sizes its result on the host), so the loss can be captured into a hipGraph.  An empty
    selection gives NaN like ``x[mask].mean()`` unless ``empty_is_zero`` (the reference's ``if torch.sum(mask_new) > 0``)."""
    mf = m.to(values.dtype)
    cnt = mf.sum()
    total = (torch.where(m, values, torch.zeros_like(values))).sum()
    return total / (torch.clamp(cnt, min=1.0) if empty_is_zero else cnt)


def full_loss(depths, depths_upsampled, confidences, depths_gt, mask, depth_min, depth_max, regress=True):
    """models/net.py:131-190: cross-entropy on the 256-bin probabilities (one-hot GT bin), L1 on the
    normalised initial / windowed per-iteration / up-sampled depths, BCE on the confidence logits;
    iteration k of n weighted by 0.8^(n-k-1).  Every ``x[mask].mean()`` of the reference is a masked sum over a count here
    (``_masked_mean``): same value up to the order of the summation, no host synchronisation -- the whole training step
    (``train_step.CapturedTrainStep``) replays as one hipGraph."""
    probs = depths["probability"]
    bins = probs[0].size(1)
    m_full = mask["level_0"] > 0.5
    m_q = mask["level_2"] > 0.5
    gt_full, gt_q = depths_gt["level_0"], depths_gt["level_2"]
    b = gt_q.shape[0]
    inv_min = (1.0 / depth_min).view(b, 1, 1, 1)
    inv_max = (1.0 / depth_max).view(b, 1, 1, 1)
    ngt = _norm(gt_q, inv_min, inv_max)
    gt_bin = torch.floor(torch.clamp(ngt, 0, 1) * (bins - 1) * m_q.float()).long()
    target = torch.zeros_like(probs[0]).scatter_(1, gt_bin, 1)

    n = len(depths["combine"])
    total = (0.8 ** n) * BINS * _masked_mean((_norm(depths["initial"][0], inv_min, inv_max) - ngt).abs(), m_q)
    for k in range(n):
        coff = 0.8 ** (n - k - 1)
        p = torch.clamp(probs[k], min=1e-5)
        total = total + coff * _masked_mean(-(target * torch.log(p)).sum(1, keepdim=True), m_q)
        if regress:
            with torch.no_grad():
                best = torch.argmax(p, dim=1, keepdim=True).float()
                near = (gt_bin >= best - RADIUS) & (gt_bin <= best + RADIUS)
            nd = _norm(depths["combine"][k], inv_min, inv_max)
            total = total + coff * BINS * _masked_mean((nd - ngt).abs(), m_q & near, empty_is_zero=True)   # net.py:175-176
            conf_gt = ((nd.detach() - ngt).abs() < 0.002).float()
            total = total + coff * _masked_mean(F.binary_cross_entropy_with_logits(confidences[k], conf_gt, reduction="none"), m_q)
    ngt_full = _norm(gt_full, inv_min, inv_max)
    return total + BINS * _masked_mean((_norm(depths_upsampled[0], inv_min, inv_max) - ngt_full).abs(), m_full)
