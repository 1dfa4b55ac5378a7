"""Test infrastructure: the inference engine with every convolution on PyTorch-ROCm (MIOpen / ATen) instead of the
hand-written MFMA kernels.  An independent second implementation of the dense layers on the same GPU, used by the
cross-backend parity tests (the product engine has exactly one back-end); the gather / regression / up-sampling
kernels are the library's in both."""
import torch
import torch.nn.functional as F

from itermvs_amd import ops
from itermvs_amd.engine import HIDDEN, INIT_SAMPLES, InferenceEngine


class MiopenEngine(InferenceEngine):
    def _pack_weights(self) -> None:      # plain torch weights are used as they are
        pass

    def _cbr_t(self, x, name, stride, relu):
        wt, b = self.cbr[name]
        y = F.conv2d(x, wt, b, stride=stride, padding=1)
        return F.relu_(y) if relu else y

    def _res_t(self, x, name, stride):
        y = self._cbr_t(self._cbr_t(x, name + "conv1.", stride, True), name + "conv2.", 1, False)
        if stride != 1:
            x = self._cbr_t(x, name + "downsample.", stride, False)
        return F.relu_(y.add_(x))

    def feature_net(self, x, compose=None):
        w, p = self.w, "feature_net."
        if compose is not None:             # (the product composes the cameras in its stem launch: same kernel body, own launch here)
            mats, flag, depth_range = compose
            self.composed = list(ops.compose_proj(mats, flag, depth_range))
        f0 = self._cbr_t(x, "conv1.", 1, True)
        f1 = self._res_t(self._res_t(f0, "layer1.0.", 2), "layer1.1.", 1)
        f2 = self._res_t(self._res_t(f1, "layer2.0.", 2), "layer2.1.", 1)
        f3 = self._res_t(self._res_t(f2, "layer3.0.", 2), "layer3.1.", 1)
        o3 = F.conv2d(f3, w[p + "output3.weight"], w[p + "output3.bias"], padding=1)
        mid = ops.bilinear_up(f3, 2).add_(F.conv2d(f2, w[p + "inner2.weight"], w[p + "inner2.bias"]))   # net.py:46
        o2 = F.conv2d(mid, w[p + "output2.weight"], w[p + "output2.bias"], padding=1)
        mid = ops.bilinear_up(mid, 2).add_(F.conv2d(f1, w[p + "inner1.weight"], w[p + "inner1.bias"]))  # net.py:49
        o1 = F.conv2d(mid, w[p + "output1.weight"], w[p + "output1.bias"], padding=1)
        self.o2_planar = o2.contiguous()
        # 16-bit feature storage: torch's cast rounds to nearest even like the HIP convolutions' channels-last epilogue
        return {l: ops.channels_last(f).to(self.feature_dtype) for l, f in ((1, o1), (2, o2), (3, o3))}

    def _corr_net(self, x, level):
        w, p = self.w, f"iter_mvs.evaluation.corr_conv1.{level - 1}."
        c0 = F.relu_(F.conv2d(x, w[p + "conv0.conv.weight"], padding=1))
        c1 = F.relu_(F.conv2d(c0, w[p + "conv1.conv.weight"], stride=2, padding=1))
        c2 = F.relu_(F.conv2d(c1, w[p + "conv2.conv.weight"], stride=2, padding=1))
        u1 = F.conv_transpose2d(c2, w[p + "conv3.weight"], stride=2, padding=1, output_padding=1).add_(c1)
        u0 = F.conv_transpose2d(u1, w[p + "conv4.weight"], stride=2, padding=1, output_padding=1).add_(c0)
        return F.conv2d(u0, w[p + "conv5.weight"], w[p + "conv5.bias"], padding=1)

    def corr_nets(self, x, levels, seg_end=(), out=None, out2=None):
        bounds = [0] + list(seg_end) + [x.shape[0]]
        y = torch.cat([self._corr_net(x[bounds[i]:bounds[i + 1]], l) for i, l in enumerate(levels)], 0)
        for o in (out, out2):
            if o is not None:
                o.copy_(y)
        return y

    def depth_head(self, hidden):
        w, p = self.w, "iter_mvs.update.depth_head."
        x = F.relu_(F.conv2d(hidden, w[p + "0.weight"], padding=2, dilation=2))
        x = F.relu_(F.conv2d(x, w[p + "2.weight"]))
        return F.conv2d(x, w[p + "4.weight"], w[p + "4.bias"])

    def confidence(self, hidden, out=None):
        w, p = self.w, "iter_mvs.update.confidence_head."
        x = F.relu_(F.conv2d(hidden, w[p + "0.weight"], padding=2, dilation=2))
        y = torch.sigmoid_(F.conv2d(x, w[p + "2.weight"], w[p + "2.bias"]))
        if out is not None:
            out.copy_(y)
        return y

    def upsample_logits(self, ref2_nchw, ws):
        w, u = self.w, "iter_mvs.upsample."
        return F.conv2d(F.relu_(F.conv2d(ref2_nchw.contiguous(), w[u + "0.weight"], padding=1)), w[u + "2.weight"])

    def stage_init(self, ws, src3, ref3, proj3, inv_min, inv_max, trace=None):
        b, _, h3, w3 = ref3.shape
        s = len(src3)
        w, pv = self.w, "iter_mvs.evaluation.pixel_view_weight."
        corr_v = ops.corr_init(src3, ref3, proj3, inv_min, inv_max, INIT_SAMPLES)
        x = F.relu_(F.conv2d(corr_v.view(b * s * INIT_SAMPLES, 8, h3, w3), w[pv + "conv.0.conv.weight"], padding=1))
        x = F.conv2d(x, w[pv + "conv.1.weight"], w[pv + "conv.1.bias"])
        vw = ops.softmax_max(x.view(b * s, INIT_SAMPLES, h3, w3))
        view_w = ops.bilinear_up(vw, 2).view(b, s, 2 * h3, 2 * w3)
        agg0 = ops.view_aggregate(corr_v, vw.view(b, s, h3, w3))
        score0 = self.stage_score0(agg0)
        self.stage_hidden0(ws, score0)
        if trace is not None:
            trace.update(corr_views=corr_v, view_weights=view_w, init_agg=agg0, init_score=score0, hidden0=ws["hidden"].clone())
        return view_w

    def stage_hidden0(self, ws, score0):
        w, hi = self.w, "iter_mvs.update.hidden_init_head."
        x = F.conv2d(F.relu_(F.conv2d(score0, w[hi + "0.weight"], padding=1)), w[hi + "2.weight"], w[hi + "2.bias"])
        hidden0 = ops.bilinear_up(x, 2, act="tanh")
        ws["hidden"].copy_(hidden0)
        ws["hx"][:, :HIDDEN].copy_(hidden0)

    def stage_head(self, ws, want_logits=False, want_best=False, with_conf=False):
        if with_conf:
            self.confidence(ws["hidden"], ws["conf"])
        logits = self.depth_head(ws["hidden"])
        _, _, best = ops.prob_regress(logits, nd_out=[(ws["hx"], HIDDEN), (ws["hx2"], HIDDEN)], want_best=True)
        return logits, best

    def stage_corrnets(self, ws):
        b = ws["b"]
        h, wd = ws["hx"].shape[2:]
        scores = [self._corr_net(a.view(-1, 8, h, wd), l).view(b, -1, h, wd) for l, a in zip((1, 2, 3), ws["agg"])]
        ops.pack_scores(scores, ws["hx"], ws["hx2"], HIDDEN + 1)

    def stage_gru(self, ws):
        w, g = self.w, "iter_mvs.update.gru."
        hx, hx2 = ws["hx"], ws["hx2"]
        zr = F.conv2d(hx, self.w_zr, self.b_zr, padding=2, dilation=2)
        ops.gru_rh(zr, hx, hx2, HIDDEN)
        q = F.conv2d(hx2, w[g + "convq.weight"], w[g + "convq.bias"], padding=2, dilation=2)
        ops.gru_out(zr, q, hx, ws["hidden"], HIDDEN)
