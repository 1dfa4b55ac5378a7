"""Differentiable (training) form of the pipeline and ``full_loss`` -- models/net.py:36-51,
78-190 and the ``not self.test`` branches of models/itermvs.py:253-329.

Training needs gradients w.r.t. the feature maps and every weight, so this path keeps the
reference's tensor-level structure and runs it on PyTorch-ROCm autograd, with the homography
warp -- the only non-dense operator on the path -- provided by the hand-written HIP kernels
(``itermvs_warp`` forward, ``itermvs_warp_backward`` scatter-add) wrapped in a
``torch.autograd.Function`` (ops.warp), and the camera composition by ``itermvs_compose_proj``.
Like the reference (module.py:77) no gradient flows to depth hypotheses or cameras.
Everything runs on the GPU; CPU tensors are rejected upstream (net.py counterpart).
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
        y = F.batch_norm(y, w[p + "bn.running_mean"], w[p + "bn.running_var"], w[p + "bn.weight"], w[p + "bn.bias"],
                         training=self.bn_training, momentum=0.1, eps=1e-5)
        if self.bn_training:
            w[p + "bn.num_batches_tracked"].add_(1)
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

    # -- Evaluation pieces (itermvs.py:333-381) -------------------------------------------------
    @staticmethod
    def _fold_n(x):                                    # [B,G,N,H,W] -> [B*N,G,H,W]
        b, g, n, h, w = x.shape
        return x.permute(0, 2, 1, 3, 4).reshape(b * n, g, h, w)

    def view_weight(self, corr):
        w, p = self.w, "iter_mvs.evaluation.pixel_view_weight."
        b, _, n, h, wd = corr.shape
        x = F.relu(F.conv2d(self._fold_n(corr), w[p + "conv.0.conv.weight"], padding=1))
        x = F.conv2d(x, w[p + "conv.1.weight"], w[p + "conv.1.bias"]).view(b, n, h, wd)
        return torch.softmax(x, dim=1).max(dim=1, keepdim=True)[0]

    def corr_net(self, corr, level):
        w, p = self.w, f"iter_mvs.evaluation.corr_conv1.{level - 1}."
        b, _, n, h, wd = corr.shape
        c0 = F.relu(F.conv2d(self._fold_n(corr), w[p + "conv0.conv.weight"], padding=1))
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


def _group_corr(warped: Tensor, ref: Tensor) -> Tensor:      # itermvs.py:50-51
    b, c, n, h, w = warped.shape
    return (warped.view(b, G, c // G, n, h, w) * ref.view(b, G, c // G, 1, h, w)).mean(dim=2)


def train_forward(w: Mapping[str, Tensor], imgs: Tensor, projs: Dict[int, Tensor], depth_min: Tensor,
                  depth_max: Tensor, iteration: int, bn_training: bool = True):
    """``Pipeline(test=False).forward``: returns the reference's training dict (net.py:115-120)."""
    net = _Net(w, bn_training)
    b, v, _, hh, ww = imgs.shape
    s = v - 1
    feats = net.features(imgs.reshape(b * v, 3, hh, ww))
    pv = {l: f.view(b, v, *f.shape[1:]) for l, f in feats.items()}
    nan_flag = torch.zeros((1,), device=imgs.device, dtype=torch.int32)
    with torch.no_grad():
        proj = ops.compose_proj(torch.stack([projs[1], projs[2], projs[3]]).reshape(3 * b, v, 4, 4), nan_flag).view(3, b, s, 12)
    h, wd = feats[2].shape[2:]
    inv_min = (1.0 / depth_min).view(b, 1, 1, 1)
    inv_max = (1.0 / depth_max).view(b, 1, 1, 1)

    u = "iter_mvs.upsample."
    up_w = F.conv2d(F.relu(F.conv2d(pv[2][:, 0], w[u + "0.weight"], padding=1)), w[u + "2.weight"])
    up_w = torch.softmax(up_w.view(b, 1, 9, 4, 4, h, wd), dim=2)

    # ---- initialisation: itermvs.py:36-82 ------------------------------------------------------
    k = torch.arange(INIT_SAMPLES, device=imgs.device, dtype=torch.float32).view(1, -1, 1, 1)
    samples = 1.0 / (inv_max + (k.expand(b, INIT_SAMPLES, h // 2, wd // 2) / (INIT_SAMPLES - 1)) * (inv_min - inv_max))
    acc, wsum, vws = 0, 1e-5, []
    for i in range(s):
        corr = _group_corr(ops.warp(pv[3][:, i + 1], proj[2][:, i], samples), pv[3][:, 0])
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
    ref_q = {1: F.interpolate(pv[1][:, 0], scale_factor=0.5, mode="bilinear"), 2: pv[2][:, 0],
             3: F.interpolate(pv[3][:, 0], scale_factor=2, mode="bilinear")}            # itermvs.py:95-98
    offs = {l: torch.tensor(o, device=imgs.device).view(1, -1, 1, 1) for l, o in sample_offsets().items()}
    view_w_const = view_w.detach()                                                       # itermvs.py:295

    # ---- iterations: itermvs.py:288-314 ----------------------------------------------------------
    for it in range(iteration):
        scores = []
        for l in (1, 2, 3):
            d = _unnorm(torch.clamp(nd + offs[l], 0, 1), inv_min, inv_max)
            acc, wsum = 0, 1e-5
            for i in range(s):
                corr = _group_corr(ops.warp(pv[l][:, i + 1], proj[l - 1][:, i], d), ref_q[l])
                vw = view_w_const[:, i].view(b, 1, 1, h, wd)
                acc = acc + corr * vw
                wsum = wsum + vw
            scores.append(net.corr_net(acc / wsum, l))
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
    assert int(nan_flag.item()) == 0, "nan in proj (singular or non-finite camera matrix, module.py:83,87)"
    return {"depths": depths, "depths_upsampled": depths_up, "confidences": confidences,
            "confidence_upsampled": conf_up}


def full_loss(depths, depths_upsampled, confidences, depths_gt, mask, depth_min, depth_max, regress=True):
    """models/net.py:131-190: cross-entropy on the 256-bin probabilities (one-hot GT bin), L1 on the
    normalised initial / windowed per-iteration / up-sampled depths, BCE on the confidence logits;
    iteration k of n weighted by 0.8^(n-k-1)."""
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
    total = (0.8 ** n) * BINS * F.l1_loss(_norm(depths["initial"][0], inv_min, inv_max)[m_q], ngt[m_q])
    for k in range(n):
        coff = 0.8 ** (n - k - 1)
        p = torch.clamp(probs[k], min=1e-5)
        total = total + coff * (-(target * torch.log(p)).sum(1, keepdim=True))[m_q].mean()
        if regress:
            with torch.no_grad():
                best = torch.argmax(p, dim=1, keepdim=True).float()
                near = (gt_bin >= best - RADIUS) & (gt_bin <= best + RADIUS)
            nd = _norm(depths["combine"][k], inv_min, inv_max)
            sel = m_q & near
            if sel.sum() > 0:
                total = total + coff * BINS * F.l1_loss(nd[sel], ngt[sel])
            conf_gt = (torch.abs(nd[m_q].detach() - ngt[m_q]) < 0.002).float()
            total = total + coff * F.binary_cross_entropy_with_logits(confidences[k][m_q], conf_gt)
    ngt_full = _norm(gt_full, inv_min, inv_max)
    return total + BINS * F.l1_loss(_norm(depths_upsampled[0], inv_min, inv_max)[m_full], ngt_full[m_full])
