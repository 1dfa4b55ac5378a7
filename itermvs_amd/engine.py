"""MI355X inference engine for the IterMVS matching hot path (test mode of ``Pipeline``).

Every compute kernel of a depth map is a hand-written HIP kernel behind the C ABI (``itermvs_amd.ops``); PyTorch-ROCm
provides device memory, streams and hipGraph capture only.  Data flow per batch of reference views (all buffers stay
resident in HBM; ``K`` = one launch):

    feature_net      FeatureNet, BN folded, all B*V views in one batch, MFMA convs              net.py:36-65
                       -> channels-last pyramids f1 [BV,H/2,W/2,16], f2 [BV,H/4,W/4,32], f3 [BV,H/8,W/8,48]
       (the stem launch also composes src @ inv(ref) for 3 levels x S views + the inverse depth range: module.py:77-90)
    K  ref_quarter     reference features on the 1/4 grid, packed                                itermvs.py:95-98
    stage_init       K corr_init (per-view group correlation, 32 hypotheses)                     itermvs.py:48-51
                       PixelViewWeight: K conv 3x3 + K pvw_tail (1x1 + softmax + max), K bilinear_up   :333-350,56
                     K view_aggregate, CorrNet (6 K), hidden-init head (2 K) + K bilinear_up2(tanh)    :59-70,159-164
    stage_head       K head_fused: depth head + softmax + arg-max window regression               itermvs.py:139-145,171-190
    per iteration (itermvs.py:288-324):
      stage_corr     K corr_iter: hypotheses + warp + gather + group corr + view mean, 3 levels
      stage_corrnets 3 x CorrNet in 6 K (three weight sets per launch) -> scores into the GRU input buffers
      stage_gru      ConvGRU: K (z, r gates, one launch two results) + K (q, state update)       module.py:52-66
      stage_head     (+ confidence head on the last iteration)
    K  convex_upsample 9-tap softmax + convex x4 + un-normalise                                  module.py:127-152
    K  bilinear_up x4  confidence                                                                itermvs.py:323

The stages read and write the persistent workspace of ``_workspace`` (GRU input buffers ``hx`` / ``hx2``, hidden state,
CorrNet inputs); ``run`` is their composition, the teacher-forced parity tests drive them one by one.
"""
from __future__ import annotations

import gc
import itertools
import weakref
from typing import Dict, List, Mapping, Tuple

import torch

from . import ops

Tensor = torch.Tensor

INIT_SAMPLES = 32          # itermvs.py:237
HIDDEN = 32                # net.py:72
# itermvs.py:229-235: corr_interval * interval_scale, evaluated in fp32 like the reference
_INTERVALS = {1: (-2.0, -2.0 / 3, 2.0 / 3, 2.0), 2: (-8.0, -8.0 / 3, 8.0 / 3, 8.0), 3: (-32.0, 32.0)}


def sample_offsets() -> Dict[int, Tuple[float, ...]]:
    out = {}
    for l, vals in _INTERVALS.items():
        t = torch.tensor(vals, dtype=torch.float32) * (1.0 / 256)
        out[l] = tuple(float(v) for v in t)
    return out


def fold_batchnorm(w: Mapping[str, Tensor], prefix: str, eps: float = 1e-5) -> Tuple[Tensor, Tensor]:
    """conv (no bias) followed by eval-mode BatchNorm -> (weight, bias) of one conv."""
    scale = w[prefix + "bn.weight"] / torch.sqrt(w[prefix + "bn.running_var"] + eps)
    weight = w[prefix + "conv.weight"] * scale.view(-1, 1, 1, 1)
    bias = w[prefix + "bn.bias"] - w[prefix + "bn.running_mean"] * scale
    return weight.contiguous(), bias.contiguous()


class InferenceEngine:
    """Test-mode ``IterMVS.forward`` (itermvs.py:253-329) on hand-written HIP kernels."""

    def __init__(self, weights: Mapping[str, Tensor], iteration: int, feature_dtype: str = "fp32", projection: str = "device_fp64",
                 conv_arithmetic: str = "bf16x3", side_branch: bool = False):
        """``projection``: how ``src_proj @ inverse(ref_proj)`` (module.py:77-90) is composed.  "device_fp64" (default): on the
        GPU in fp64, rounded once, inside the launch that packs the reference features -- no host round trip, within 5e-5
        px of any fp32 evaluation.  "host_fp32": on the host with torch in fp32, operation for operation like the reference
        (``torch.inverse`` per batch item, then ``torch.matmul``) -- for users who need tap indices identical to a
        reference run on the same host.  With the cameras handed over as CPU tensors (where a data loader has them anyway)
        the composition costs no synchronisation and the mode IS capturable: the graph reads the composed [3,B,S,12] matrices
        from a static buffer that each replay refreshes through pinned memory (``run(..., composed=)``,
        ``GraphedRunner(..., composed=)``).  With device-resident cameras it costs a device-to-host copy and a synchronisation per
        forward and cannot be captured.  (The reference itself has no single bit pattern here: its fp32 LAPACK inverse
        differs between CPU BLAS builds and from its own CUDA path.)
        ``feature_dtype``: storage type of the three feature pyramids the correlation kernels gather from -- "fp32"
        (default, the reference's numerics), "bf16" or "fp16" (BASELINE cfg 4 / cfg 5: half the gathered bytes, fp32
        arithmetic; the output convolutions of FeatureNet round their fp32 results to nearest even).
        ``conv_arithmetic``: how the 3x3 convolutions with more than 8 input channels multiply (net.py:36-66, module.py:6-66,
        itermvs.py:139-164): "bf16x3" (default) = both operands split EXACTLY into three bf16 terms, the six largest cross
        products accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (csrc/conv_tile3.hip: error of the size of one fp32
        rounding per product, 2.7x less matrix-pipe time); "fp32" = v_mfma_f32_16x16x4_f32, bit-for-bit an fmaf chain."""
        if conv_arithmetic not in ("bf16x3", "fp32"):
            raise ValueError(f"conv_arithmetic must be 'bf16x3' or 'fp32', got {conv_arithmetic!r}")
        self.split3 = conv_arithmetic == "bf16x3"
        if feature_dtype not in ops.FEATURE_DTYPES:
            raise ValueError(f"feature_dtype must be one of {sorted(ops.FEATURE_DTYPES)}, got {feature_dtype!r}")
        self.feature_dtype = ops.FEATURE_DTYPES[feature_dtype]
        if projection not in ("device_fp64", "host_fp32"):
            raise ValueError(f"projection must be 'device_fp64' or 'host_fp32', got {projection!r}")
        self.projection = projection
        w = {k: v.detach() for k, v in weights.items()}
        dev = w["feature_net.conv1.conv.weight"].device
        if dev.type != "cuda":
            raise RuntimeError("InferenceEngine needs the model on an MI355X (model.cuda()); no CPU fallback")
        self.device = dev
        self.iteration = iteration
        self.w = w
        fn = "feature_net."
        self.cbr: Dict[str, Tuple[Tensor, Tensor]] = {}
        names = ["conv1."] + [f"layer{l}.{b}.{c}." for l in (1, 2, 3) for b, cs in ((0, ("conv1", "conv2", "downsample")),
                                                                                     (1, ("conv1", "conv2"))) for c in cs]
        for n in names:
            self.cbr[n] = fold_batchnorm(w, fn + n)
        g = "iter_mvs.update.gru."
        self.w_zr = torch.cat([w[g + "convz.weight"], w[g + "convr.weight"]], 0).contiguous()
        self.b_zr = torch.cat([w[g + "convz.bias"], w[g + "convr.bias"]], 0).contiguous()
        self.offsets = sample_offsets()
        self._ws: Dict[tuple, dict] = {}
        self._released: list = []                    # workspaces of dropped runners, parked until their last replay is done
        self._ws_owner = None
        self._owner_tokens = itertools.count(1)      # never reused, unlike id(): a new runner cannot alias an old workspace
        # OR-ed with 1 by itermvs_compose_proj when a composed projection is NaN (module.py:83,87 assert on the host);
        # read and cleared by check_projection_finite()
        self.nan_flag = torch.zeros((1,), device=dev, dtype=torch.int32)
        # ref_quarter and the up-sampling weights on a second stream = a parallel branch of the captured graph (see run)
        self.side_branch = side_branch
        self._side = None
        self.profile_iterations = None   # set of iteration indices whose corr_iter launch carries timing events (None = all)
        self.profile_init = True         # whether the corr_init launch carries timing events (bench.py)
        self.pk: Dict[str, object] = {}
        self._pack_weights()

    # -- weights --------------------------------------------------------------------------------
    def _pack_weights(self) -> None:
        """Re-lay every conv weight once (BatchNorm already folded) in the matrix-core formats of ``ops.MfmaWeight``
        (CorrNet's two transposed convolutions included)."""
        w, pk = self.w, self.pk
        # the two layers with almost no contraction to feed a matrix core (3 -> 8 on the full-resolution images,
        # 8 -> 1 at the end of CorrNet) run faster on the one-thread-per-pixel VALU kernel: 31 vs 39 us, 6.7 vs 10.2 us
        pack = lambda wt: ops.pack_conv_weight(wt) if (wt.shape[1] <= 4 or wt.shape[0] == 1) and wt.shape[2] == 3 else ops.MfmaWeight(wt, split3=self.split3)
        for n, (wt, _) in self.cbr.items():
            pk["feature_net." + n] = pack(wt)
        for k, v in w.items():
            if not k.endswith("weight") or v.dim() != 4 or ".bn." in k or k.startswith("feature_net.") and ".conv." in k:
                continue
            if k.endswith("conv3.weight") or k.endswith("conv4.weight"):      # ConvTranspose2d (itermvs.py:359-363)
                pk[k] = ops.MfmaWeight(v, transposed=True)
            else:
                pk[k] = pack(v)
        pv = "iter_mvs.evaluation.pixel_view_weight.conv.1."
        cf = "iter_mvs.update.confidence_head.2."
        self.conf_dot = torch.cat([w[cf + "weight"].reshape(-1), w[cf + "bias"].reshape(-1)]).float().contiguous()
        self.pvw_dot = torch.cat([w[pv + "weight"].reshape(-1), w[pv + "bias"].reshape(-1)]).float().contiguous()
        self.stem_w = ops.pack_stem_weights(*self.cbr["conv1."], *self.cbr["layer1.0.conv1."], *self.cbr["layer1.0.downsample."])
        self.corrnet_w = {l: ops.pack_corrnet_weights(w, f"iter_mvs.evaluation.corr_conv1.{l - 1}.", split3=self.split3) for l in (1, 2, 3)}
        dh = "iter_mvs.update.depth_head."
        self.head_w1, self.head_w2 = ops.pack_head_weights(w[dh + "2.weight"], w[dh + "4.weight"])
        if self.split3:     # the 64 -> 256 layer of the fused head on the bf16 matrix instruction (bf16x3 arithmetic, head.hip W2B)
            self.head_w2_fused = ops.pack_head_w2_split3(w[dh + "4.weight"])
            self.head_w0_fused = ops.pack_head_w0_split3(w[dh + "0.weight"])     # ... and its dilated 3x3 layer (launches without the confidence rider)
        else:
            self.head_w2_fused = self.head_w2
            self.head_w0_fused = None
        # (z / r gates, 43 -> 64 dilated at 1/4 resolution: the bf16x3 form measured 18.7 us against 18.2 us -- four channel
        #  blocks stage and split each tile four times; the q convolution, two blocks, gains: 11.1 vs 13.1 us)
        self.pk_zr = ops.MfmaWeight(self.w_zr, split3=False)
        self.gru_coop = self.split3      # both GRU convolutions as cooperative bf16x3 kernels (gru.hip)
        if self.gru_coop:
            g = "iter_mvs.update.gru."
            self.gru_zr = ops.pack_gru_conv_split3(self.w_zr)
            self.gru_q = ops.pack_gru_conv_split3(w[g + "convq.weight"])
            self.gru_bq = w[g + "convq.bias"].contiguous()
        # the two conv3x3 32 -> 64 + ReLU + conv1x1 heads, one launch each (csrc/stack2.hip)
        hi, up = "iter_mvs.update.hidden_init_head.", "iter_mvs.upsample."
        self.pk_hi0, self.pk_up0 = ops.MfmaWeight(w[hi + "0.weight"], split3=False), ops.MfmaWeight(w[up + "0.weight"], split3=False)
        self.hi1, self.hi1_bias = ops.pack_conv1x1_operand(w[hi + "2.weight"], w[hi + "2.bias"])
        self.up1, _ = ops.pack_conv1x1_operand(w[up + "2.weight"])
        if self.split3:          # both small heads in the bf16x3 arithmetic (stack2.hip W3)
            self.pk_hi0, self.hi1, self.hi1_bias = ops.pack_conv3x3_conv1x1_split3(w[hi + "0.weight"], w[hi + "2.weight"], w[hi + "2.bias"])
            self.pk_up0, self.up1, _ = ops.pack_conv3x3_conv1x1_split3(w[up + "0.weight"], w[up + "2.weight"])
        # the confidence head's 3x3 layer in the fp32 tile format (it rides in the depth head's last launch, csrc/head.hip)
        self.pk_conf = ops.MfmaWeight(w["iter_mvs.update.confidence_head.0.weight"], split3=False)

    def _side_stream(self) -> "torch.cuda.Stream":
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _conv(self, x: Tensor, name: str, bias: bool = False, **kw) -> Tensor:
        """one layer by state-dict name (``name`` + "weight"/"bias")"""
        return ops.conv2d(x, self.pk[name + "weight"], self.w[name + "bias"] if bias else None, **kw)

    # -- FeatureNet -----------------------------------------------------------------------------
    def _cbr(self, x: Tensor, name: str, stride: int, act: str, add: Tensor = None) -> Tensor:
        return ops.conv2d(x, self.pk["feature_net." + name], self.cbr[name][1], stride=stride, act=act, add=add)

    def _res(self, x: Tensor, name: str, stride: int) -> Tensor:
        if stride == 1:
            y = self._cbr(x, name + "conv1.", 1, "relu")
            return self._cbr(y, name + "conv2.", 1, "relu", add=x)      # relu(x + y), module.py:50
        # stride-2 block: conv1 (+ReLU) and the down-sampling shortcut read the same input -> one launch, two results
        key = "feature_net." + name + "conv1+downsample"
        if key not in self.pk:
            (w1, b1), (wd, bd) = self.cbr[name + "conv1."], self.cbr[name + "downsample."]
            self.pk[key] = (ops.MfmaWeight(torch.cat([w1, wd]), split3=self.split3), torch.cat([b1, bd]).contiguous(), w1.shape[0])
        wt, bias, c = self.pk[key]
        n, _, hh, ww = x.shape
        sc = torch.empty((n, c, (hh - 1) // 2 + 1, (ww - 1) // 2 + 1), device=x.device)
        y = ops.conv2d(x, wt, bias, stride=2, act="relu", split=(c, "none", sc))
        return self._cbr(y, name + "conv2.", 1, "relu", add=sc)

    def feature_net(self, x: Tensor, compose=None) -> Dict[int, Tensor]:
        """net.py:36-65 with BN folded; x [M,3,H,W] -> channels-last pyramids {1,2,3} (the layout the correlation kernels
        gather from); level 2 also keeps a planar copy (``o2_planar``) for the up-sampling head.
        ``compose`` (see ops.stem): the camera composition rides in the stem launch; its results land in ``self.composed``."""
        p = "feature_net."
        m, _, hh, ww = x.shape
        dev = x.device
        cl = lambda c, s: torch.empty((m, c, hh // s, ww // s), device=dev, dtype=self.feature_dtype,
                                      memory_format=torch.channels_last)
        o1, o2, o3 = cl(16, 2), cl(32, 4), cl(48, 8)
        self.o2_planar = torch.empty((m, 32, hh // 4, ww // 4), device=dev)
        # conv1 + layer1[0].conv1 / .downsample in one launch, fea0 never leaves LDS (+ the camera composition, module.py:77-90)
        # (bf16x3 arithmetic: the stem's two results go to the chain kernel as channel quads -- 16-byte stores and loads)
        if compose is not None:
            y, sc, *self.composed = ops.stem(x, *self.stem_w, compose=compose, quads=self.split3)
        else:
            y, sc = ops.stem(x, *self.stem_w, quads=self.split3)
        if self.split3:
            # layer1[0].conv2 (+ shortcut) and both layers of layer1[1] in one launch, the two intermediate maps in LDS (res_chain.hip)
            names = ("layer1.0.conv2.", "layer1.1.conv1.", "layer1.1.conv2.")
            f1 = ops.res_chain16(y, sc, [self.pk[p + k] for k in names], [self.cbr[k][1] for k in names], quads=True)
        else:
            f1 = self._res(self._cbr(y, "layer1.0.conv2.", 1, "relu", add=sc), "layer1.1.", 1)
        f2 = self._res(self._res(f1, "layer2.0.", 2), "layer2.1.", 1)
        f3 = self._res(self._res(f2, "layer3.0.", 2), "layer3.1.", 1)
        self._conv(f3, p + "output3.", bias=True, channels_last_out=True, out=o3)
        mid = self._conv(f2, p + "inner2.", bias=True, ksize=1, pad=0, add=f3, add_up2=True)   # net.py:46 (fused F.interpolate)
        self._conv(mid, p + "output2.", bias=True, channels_last_out=True, out=o2, out2=self.o2_planar)
        if self.split3:
            # net.py:49-50 in one launch: the 48-channel half-resolution map (78 MB at cfg 1) stays in LDS (lat_conv.hip)
            ops.lateral_conv3x3(f1, mid, self.pk[p + "inner1.weight"], self.w[p + "inner1.bias"], self.pk[p + "output1.weight"],
                                self.w[p + "output1.bias"], out=o1, channels_last_out=True)
        else:
            mid = self._conv(f1, p + "inner1.", bias=True, ksize=1, pad=0, add=mid, add_up2=True)  # net.py:49
            self._conv(mid, p + "output1.", bias=True, channels_last_out=True, out=o1)
        return {1: o1, 2: o2, 3: o3}

    # -- small stacks ---------------------------------------------------------------------------
    def corr_nets(self, x: Tensor, levels, seg_end=(), out: Tensor = None, out2: Tensor = None) -> Tensor:
        """itermvs.py:352-381 for one or three levels: x [M,8,h,w] whose batch items [0,seg_end[0]) / [seg_end[0],seg_end[1]) /
        rest belong to levels[0..2] -> [M,1,h,w].  ONE launch (itermvs_corrnet: the whole U-Net per 32 x 32 tile in LDS)."""
        return ops.corrnet(x, [self.corrnet_w[l] for l in levels], seg_end, out=out, out2=out2)

    def depth_head(self, hidden: Tensor) -> Tensor:
        """itermvs.py:139-145 layer by layer -> logits [B,256,h,w] (traced / teacher-forced runs only)"""
        p = "iter_mvs.update.depth_head."
        x = self._conv(hidden, p + "0.", pad=2, dilation=2, act="relu")
        x = self._conv(x, p + "2.", ksize=1, pad=0, act="relu")
        return self._conv(x, p + "4.", bias=True, ksize=1, pad=0)

    def confidence(self, hidden: Tensor, out: Tensor = None) -> Tensor:
        """itermvs.py:147-151 + sigmoid (:198)"""
        p = "iter_mvs.update.confidence_head."
        # the 1x1 layer to one channel and the sigmoid are the epilogue of the dilated 3x3 layer (one launch, the 32-channel
        # tensor is never stored)
        return self._conv(hidden, p + "0.", pad=2, dilation=2, act="relu_dot_sigmoid", aux1=self.conf_dot, out=out)

    def upsample_logits(self, ref2_nchw: Tensor, ws: dict) -> Tensor:
        """itermvs.py:262-263 (the softmax over the 9 taps is part of convex_upsample)"""
        return ops.conv3x3_conv1x1(ref2_nchw, self.pk_up0, self.up1, None, 144, out=ws["up_logits"])

    # -- workspace ------------------------------------------------------------------------------
    def _workspace(self, b: int, h: int, w: int) -> dict:
        # one workspace per (shape, owner): eager calls share the engine's; every GraphedRunner captures onto its OWN
        # (``_ws_owner`` is set while a runner warms up / captures), so runners replayed on different streams never
        # share GRU buffers
        self._purge_released()
        key = (b, h, w, self._ws_owner)
        ws = self._ws.get(key)
        if ws is None:
            dev = self.device
            nx = 1 + sum(len(v) for v in self.offsets.values())          # GRU input channels: depth + 10 scores
            ws = {
                "hx": torch.zeros((b, HIDDEN + nx, h, w), device=dev),     # [h | nd | scores]  (module.py:60)
                "hx2": torch.zeros((b, HIDDEN + nx, h, w), device=dev),    # [r*h | nd | scores] (module.py:64)
                "hidden": torch.empty((b, HIDDEN, h, w), device=dev),
                "agg_all": torch.empty((b * (nx - 1), 8, h, w), device=dev),   # the three levels' CorrNet inputs, back to back
                "zbuf": torch.empty((b, HIDDEN, h, w), device=dev),
                "up_logits": torch.empty((b, 144, h, w), device=dev),
                "conf": torch.empty((b, 1, h, w), device=dev),
            }
            o, views = 0, []
            for l in (1, 2, 3):
                n = b * len(self.offsets[l])
                views.append(ws["agg_all"][o:o + n].view(b, len(self.offsets[l]), 8, h, w))
                o += n
            ws["agg"] = views
            ws["b"] = b
            self._ws[key] = ws
        return ws

    def _release_owner(self, token: int, streams: dict = None) -> None:
        """drop the private workspaces of a GraphedRunner that went away (its finalizer calls this -- from the garbage collector,
        i.e. at ANY point, in any thread, also in the middle of another runner's stream capture, where every synchronising or
        recording HIP call is illegal and invalidates that capture).  The buffers were allocated on the runner's capture stream
        while its graph replays on the callers' streams, so a replay may still be writing them: they are parked, and handed back
        to the caching allocator by ``_purge_released`` once events recorded AFTER the drop ON EVERY STREAM THE RUNNER EVER
        REPLAYED ON (``streams``: the runner's own record, handed over by value) have completed -- the collector's current
        stream says nothing about where the replays ran."""
        keys = [k for k in self._ws if k[3] == token]
        if not keys:
            return
        parked = [self._ws.pop(k) for k in keys]
        streams = dict(streams or {})
        self._released.append((parked, self._guard_events(streams) or streams))    # a dict = "not guarded yet"

    def _guard_events(self, streams: dict):
        """events recorded now on each of ``streams`` (plus the current one); None while a capture is in progress or at shutdown"""
        try:
            if torch.cuda.is_current_stream_capturing():
                return None
            todo = dict(streams or {})
            cur = torch.cuda.current_stream(self.device)
            todo.setdefault(cur.cuda_stream, cur)
            evs = []
            for st in todo.values():
                ev = torch.cuda.Event()
                ev.record(st)                    # everything enqueued so far on a stream that replayed
                evs.append(ev)
            return evs
        except Exception:                        # interpreter shutdown: the context may already be gone
            return None

    def _purge_released(self) -> None:
        """free parked workspaces whose guarding event has completed (called where no capture is in progress)"""
        if not self._released or torch.cuda.is_current_stream_capturing():
            return
        keep = []
        for parked, guard in self._released:
            if isinstance(guard, dict):          # dropped during a capture: guard it now, free it at a later purge
                keep.append((parked, self._guard_events(guard) or guard))
            elif not all(ev.query() for ev in guard):
                keep.append((parked, guard))
        self._released = keep

    # -- stages (each reads / writes the workspace; see the module docstring) ---------------------
    def stage_init(self, ws: dict, src3: List[Tensor], ref3: Tensor, proj3: Tensor, inv_min: Tensor, inv_max: Tensor,
                   trace: dict = None) -> Tensor:
        """itermvs.py:36-70 + :159-164: view weights [B,S,h,w] (returned) and the initial hidden state (``hidden``, ``hx``)"""
        b, _, h3, w3 = ref3.shape
        s = len(src3)
        w = self.w
        # [B,S,32,8,h3,w3]; bf16x3 arithmetic: stored groups last -- PixelViewWeight's 3x3 layer then stages a pixel's 8 group
        # correlations with two 16-byte loads instead of eight dwords (its staging was 21 of its 24 us)
        corr_v = ops.corr_init(src3, ref3, proj3, inv_min, inv_max, INIT_SAMPLES, timed=self.profile_init, groups_last=self.split3)
        pv = "iter_mvs.evaluation.pixel_view_weight."
        # PixelViewWeight (itermvs.py:333-350): 3x3 layer with ReLU and the 1x1 layer to one channel in its epilogue (the
        # 16-channel tensor is never stored), then softmax over the 32 hypotheses and its maximum
        logit = self._conv(corr_v.reshape(b * s * INIT_SAMPLES, 8, h3, w3), pv + "conv.0.conv.", act="relu_dot", aux1=self.pvw_dot)
        vw = ops.softmax_max(logit.view(b * s, INIT_SAMPLES, h3, w3))
        # [B,32,8,h3,w3]; view weights x2 (itermvs.py:56-57,71), stored pixel-major [B,h,w,S] for the iteration kernel
        agg0, view_w = ops.view_aggregate_up(corr_v, vw.view(b, s, h3, w3), interleaved=True)
        score0 = self.stage_score0(agg0)
        self.stage_hidden0(ws, score0)
        if trace is not None:
            trace.update(corr_views=corr_v, view_weights=view_w, init_agg=agg0, init_score=score0,
                         hidden0=ws["hidden"].clone())
        return view_w

    def stage_score0(self, agg0: Tensor) -> Tensor:
        """itermvs.py:70: CorrNet[2] on the aggregated initial correlation [B,32,8,h3,w3] -> [B,32,h3,w3]"""
        b, n, _, h3, w3 = agg0.shape
        return self.corr_nets(agg0.view(b * n, 8, h3, w3), [3]).view(b, n, h3, w3)

    def stage_hidden0(self, ws: dict, score0: Tensor) -> None:
        """itermvs.py:159-164: hidden-init head + x2 bilinear + tanh, written to ``hidden`` and ``hx[:, :32]`` in one launch"""
        x = ops.conv3x3_conv1x1(score0, self.pk_hi0, self.hi1, self.hi1_bias, HIDDEN)
        ops.bilinear_up_into(x, 2, ws["hidden"], ws["hx"][:, :HIDDEN], act="tanh")

    def stage_head(self, ws: dict, want_logits: bool = False, want_best: bool = False, with_conf: bool = False):
        """depth head + softmax regression (itermvs.py:139-145, 171-190 / 201-219) on ``hidden``; the normalised depth goes
        to channel 32 of both GRU input buffers.  Default: ONE launch (itermvs_head_fused), the 256-bin logits never reach
        memory; ``want_logits`` evaluates the head layer by layer.  ``with_conf`` (last GRU iteration, itermvs.py:197-199): the
        confidence head reads the same hidden state -- it rides in the same launch and fills ``ws["conf"]``.
        Returns (logits | None, arg-max bins | None)."""
        hidden, nd_out = ws["hidden"], [(ws["hx"], HIDDEN), (ws["hx2"], HIDDEN)]
        if not want_logits:
            p = "iter_mvs.update.depth_head."
            conf = (self.pk_conf, self.conf_dot, ws["conf"]) if with_conf else None
            w0 = self.head_w0_fused if (self.head_w0_fused is not None and conf is None) else self.pk[p + "0.weight"]
            _, best = ops.head_fused(hidden, w0, self.head_w1, self.head_w2_fused, self.w[p + "4.bias"],
                                     nd_out=nd_out, want_best=want_best, conf=conf)
            return None, best
        if with_conf:
            self.confidence(hidden, ws["conf"])
        logits = self.depth_head(hidden)
        _, _, best = ops.prob_regress(logits, nd_out=nd_out, want_best=True)
        return logits, best

    def stage_corr(self, ws: dict, src, ref_q: Tensor, proj: Tensor, view_w: Tensor, inv_min: Tensor, inv_max: Tensor,
                   timed: bool = True) -> List[Tensor]:
        """itermvs.py:290-293 + :84-120: hypotheses around ``hx[:, 32]`` -> the three CorrNet inputs (``agg``)"""
        return ops.corr_iter(src, ref_q, proj, view_w, inv_min, inv_max, norm_depth=ws["hx"][:, HIDDEN:HIDDEN + 1],
                             offsets=self.offsets, out=ws["agg"], timed=timed)

    def stage_corrnets(self, ws: dict) -> None:
        """itermvs.py:121-124: the three CorrNets, one launch per layer over all 10*B maps; scores -> channels 33..42 of
        ``hx`` and ``hx2`` (for B = 1 written there directly by the last layer)"""
        b = ws["b"]
        hx, hx2 = ws["hx"], ws["hx2"]
        h, wd = hx.shape[2:]
        n1, n2 = b * len(self.offsets[1]), b * (len(self.offsets[1]) + len(self.offsets[2]))
        if b == 1:
            self.corr_nets(ws["agg_all"], (1, 2, 3), (n1, n2), out=hx[0, HIDDEN + 1:].unsqueeze(1), out2=hx2[0, HIDDEN + 1:].unsqueeze(1))
        else:
            sc = self.corr_nets(ws["agg_all"], (1, 2, 3), (n1, n2))
            scores = [sc[:n1].view(b, -1, h, wd), sc[n1:n2].view(b, -1, h, wd), sc[n2:].view(b, -1, h, wd)]
            ops.pack_scores(scores, hx, hx2, HIDDEN + 1)

    def stage_gru(self, ws: dict) -> None:
        """ConvGRU (module.py:59-66) with the gate math in the conv epilogues: the update and reset gates read the same
        input -> one launch, two results (z -> ``zbuf``, r*h -> ``hx2``); then q and the state update (-> ``hx``, ``hidden``)"""
        hx, hx2, zbuf = ws["hx"], ws["hx2"], ws["zbuf"]
        if self.gru_coop and hx.shape[1] * hx.shape[2] * hx.shape[3] * 4 <= 1 << 29:      # (its 32-bit buffer offsets: maps up to 3.1 M pixels)
            ops.gru_conv(hx, self.gru_zr, self.b_zr, hx[:, :HIDDEN], zbuf, out2=hx2[:, :HIDDEN])
            ops.gru_conv(hx2, self.gru_q, self.gru_bq, hx[:, :HIDDEN], hx[:, :HIDDEN], out2=ws["hidden"], z=zbuf)
            return
        ops.conv2d(hx, self.pk_zr, self.b_zr, pad=2, dilation=2, act="sigmoid", out=zbuf, aux1=hx[:, :HIDDEN],
                   split=(HIDDEN, "gru_rh", hx2[:, :HIDDEN]))
        self._conv(hx2, "iter_mvs.update.gru.convq.", bias=True, pad=2, dilation=2, act="gru_out", aux1=hx[:, :HIDDEN],
                   aux2=zbuf, out=hx[:, :HIDDEN], out2=ws["hidden"])

    # -- one batch of reference views -----------------------------------------------------------
    def run(self, imgs: Tensor, projs, depth_min: Tensor, depth_max: Tensor, trace: dict = None,
            composed: Tensor = None) -> Tuple[Tensor, Tensor]:
        """imgs [B,V,3,H,W]; projs {1,2,3: [B,V,4,4]} like the reference's sample dict, or the same stacked [3,B,V,4,4];
        depth_min/max [B] -> (depth [B,1,H,W], confidence [B,1,H,W]) like itermvs.py:326-327 / net.py:125-128.
        ``trace`` (dict) collects the intermediate tensors and evaluates the depth head layer by layer (logits kept).
        ``composed``: the projections already composed ([3,B,S,12] on the device, ``compose_host``); ``projs`` is then unused."""
        b, v, _, hh, ww = imgs.shape
        s = v - 1
        # camera composition (+ inverse depth range): a few threads of the FIRST launch (it depends on the cameras only)
        pstack = None if composed is not None else (projs if torch.is_tensor(projs) else torch.stack([projs[1], projs[2], projs[3]]))
        in_stem = composed is None and self.projection != "host_fp32"
        feats = self.feature_net(imgs.reshape(b * v, 3, hh, ww).contiguous(),
                                 compose=(pstack.reshape(3 * b, v, 4, 4), self.nan_flag, (depth_min, depth_max)) if in_stem else None)
        per_view = {l: feats[l].view(b, v, *feats[l].shape[1:]) for l in (1, 2, 3)}
        src = {l: [per_view[l][:, i] for i in range(1, v)] for l in (1, 2, 3)}
        ref = {l: per_view[l][:, 0] for l in (1, 2, 3)}
        h, wd = feats[2].shape[2:]
        ws = self._workspace(b, h, wd)
        hx = ws["hx"]

        f2p = self.o2_planar
        ref2_nchw = f2p[:1] if b == 1 else f2p.view(b, v, *f2p.shape[1:])[:, 0].contiguous()
        # Two launches depend on FeatureNet only and are needed late: the reference features on the 1/4 grid (first read by the first
        # corr_iter launch) and the up-sampling weights (read by the final convex up-sampling).  ``side_branch`` (OFF by default): they
        # go to a second stream -- a parallel branch of the captured hipGraph -- and run beside the initialisation stage instead of in
        # front of it; the main stream joins them where their results are first read.  Measured +25..33 us per depth map: kernels that
        # share the machine slow each other down by more than the overlap returns (profiles/r06/r06f_side_branch_ab.txt).
        main = torch.cuda.current_stream(self.device)
        side = self._side_stream() if self.side_branch else None
        ev_rq = None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ref_q = ops.ref_quarter(ref[1], ref[2], ref[3])
                ev_rq = torch.cuda.Event()
                ev_rq.record(side)
                up_logits = self.upsample_logits(ref2_nchw, ws)
            ref_q.record_stream(main)
        else:
            up_logits = self.upsample_logits(ref2_nchw, ws)             # only needed by the final convex up-sampling
            ref_q = ops.ref_quarter(ref[1], ref[2], ref[3])
        if composed is not None:
            proj = composed
            inv_min, inv_max = 1.0 / depth_min, 1.0 / depth_max                    # itermvs.py:267-268 (IEEE division, as on the host)
        elif self.projection == "host_fp32":
            proj, inv_min, inv_max = self.compose_on_host(pstack.reshape(3, b, v, 4, 4), depth_min, depth_max)
        else:
            proj, inv_min, inv_max = self.composed
        proj = proj.view(3, b, s, 12)

        view_w = self.stage_init(ws, src[3], ref[3], proj[2], inv_min, inv_max, trace)          # itermvs.py:270-276
        logits, best = self.stage_head(ws, trace is not None)
        if ev_rq is not None:
            main.wait_event(ev_rq)                                      # the first corr_iter launch reads ref_q
        if trace is not None:
            trace.update(feats=feats, proj=proj, ref_q=ref_q, logits0=logits, nd0=hx[:, HIDDEN:HIDDEN + 1].clone(), best0=best,
                         up_logits=up_logits, iters=[])

        conf = None
        for it in range(self.iteration):                                                         # itermvs.py:288-324
            # (bench.py: only the iterations in ``profile_iterations`` get timing events around this launch)
            timed = self.profile_iterations is None or it in self.profile_iterations
            nd_in = hx[:, HIDDEN:HIDDEN + 1].clone() if trace is not None else None
            aggs = self.stage_corr(ws, src, ref_q, proj, view_w, inv_min, inv_max, timed)
            self.stage_corrnets(ws)
            score = hx[:, HIDDEN + 1:].clone() if trace is not None else None
            self.stage_gru(ws)
            last = it == self.iteration - 1                                                      # itermvs.py:197-199
            logits, best = self.stage_head(ws, trace is not None, with_conf=last)
            if last:
                conf = ws["conf"]
            if trace is not None:
                trace["iters"].append(dict(nd_in=nd_in, aggs=[a.clone() for a in aggs], score=score,
                                           hidden=ws["hidden"].clone(), logits=logits, best=best,
                                           nd=hx[:, HIDDEN:HIDDEN + 1].clone(), conf=conf))

        if side is not None:
            main.wait_stream(side)                                      # the up-sampling weights
        # convex up-sampling of the depth and bilinear up-sampling of the confidence: one launch     itermvs.py:321-324
        return ops.final_upsample(up_logits, hx, inv_min, inv_max, conf, nd_channel=HIDDEN)

    def compose_on_host(self, pstack: Tensor, depth_min: Tensor, depth_max: Tensor):
        """module.py:77-90 on the host in fp32, operation for operation (``projection="host_fp32"``): pstack [3,B,V,4,4] ->
        (proj [3,B,S,12], 1/depth_min, 1/depth_max) on the device.  Synchronises (the cameras come back from the GPU)."""
        proj = self.compose_host(pstack.detach().float().cpu())
        if bool(torch.isnan(proj).any()):
            self.nan_flag.fill_(1)                                                  # surfaces through check_projection_finite
        dev = pstack.device
        inv_min, inv_max = 1.0 / depth_min.detach().float().cpu(), 1.0 / depth_max.detach().float().cpu()      # itermvs.py:267-268
        return proj.to(dev), inv_min.to(dev), inv_max.to(dev)

    @staticmethod
    def compose_host(pm: Tensor) -> Tensor:
        """module.py:77-90 on the host in fp32, operation for operation: CPU cameras [3,B,V,4,4] -> CPU [3,B,S,12] rows of
        ``(src_proj @ inverse(ref_proj))[:3, :4]`` (``torch.inverse`` per batch item like module.py:81,86, then ``torch.matmul``).
        No device work, no synchronisation."""
        pm = pm.detach().float()
        levels = []
        for l in range(3):
            ref_p = pm[l, :, 0]                                                     # [B,4,4]
            inv = torch.stack([torch.inverse(ref_p[i]) for i in range(ref_p.shape[0])])
            views = [torch.matmul(pm[l, :, k], inv)[:, :3, :4].reshape(-1, 12) for k in range(1, pm.shape[2])]
            levels.append(torch.stack(views, 1))
        return torch.stack(levels).contiguous()                                     # [3,B,S,12]

    def check_projection_finite(self) -> None:
        """Deferred form of the reference's NaN asserts (module.py:83,87) for every ``run`` / graph replay enqueued since
        the last check: waits for them (one 4-byte read), clears the flag, raises ``AssertionError`` like the reference."""
        bad = int(self.nan_flag.item())
        if bad:
            self.nan_flag.zero_()
            raise AssertionError("nan in proj (singular or non-finite camera matrix, module.py:83,87)")


def _release_workspace(engine_ref, token: int, streams: dict) -> None:
    engine = engine_ref()
    if engine is not None:
        engine._release_owner(token, streams)


class GraphedRunner:
    """One depth map per replay with no host work: ``InferenceEngine.run`` captured into ONE hipGraph
    (torch.cuda.CUDAGraph) on a private stream, with static input / output buffers.

    A replay is one graph launch instead of ~85 Python-driven kernel launches; several runners on different
    streams keep independent reference views in flight on one GPU (replay happens on the caller's current
    stream; only the capture uses a private stream).  When the library's timing hooks are enabled at capture
    time, the ``itermvs_corr_iter`` / ``itermvs_corr_init`` launches are bracketed by external event-record nodes inside
    the graph (``profile_pairs`` = their range for ``ops.profile_graph_read``), so bench.py times them in the timed region.
    Outputs are static buffers, overwritten by the next replay on the same runner."""

    def __init__(self, engine: InferenceEngine, imgs: Tensor, projs: Dict[int, Tensor], depth_min: Tensor,
                 depth_max: Tensor, stream: "torch.cuda.Stream" = None, composed: Tensor = None):
        """``composed`` (CPU [3,B,S,12], ``InferenceEngine.compose_host``): capture the form that READS composed projections
        from a static device buffer instead of composing the cameras on the device (``projs`` is then ignored): the
        capturable ``projection="host_fp32"``.  Replays refresh the buffer through a small ring of pinned staging buffers."""
        self.engine = engine
        self.stream = stream or torch.cuda.Stream(device=imgs.device)
        self.imgs = imgs.clone()
        self.composed = None
        if composed is not None:
            self.composed = composed.to(imgs.device).contiguous()                       # static [3,B,S,12]
            self._pin = [torch.empty(tuple(composed.shape), dtype=torch.float32).pin_memory() for _ in range(4)]
            self._pin_ev = [None] * len(self._pin)
            self._pin_i = 0
            self.proj_stack, self.projs = None, {}
        else:
            self.proj_stack = torch.stack([projs[1], projs[2], projs[3]]).contiguous()  # static [3,B,V,4,4]
            self.projs = {l: self.proj_stack[l - 1] for l in (1, 2, 3)}
        self.depth_min, self.depth_max = depth_min.clone(), depth_max.clone()
        self.key = (tuple(imgs.shape), tuple(depth_min.shape))
        self.stream.wait_stream(torch.cuda.current_stream(imgs.device))
        # this runner's private workspace (see _workspace), released with the runner: the finalizer holds the engine weakly
        # and the token by value, so a dropped runner frees its GRU / correlation buffers instead of leaving them in
        # engine._ws for the engine's lifetime
        self._ws_token = next(engine._owner_tokens)
        engine._ws_owner = self._ws_token
        # every stream this runner's graph is ever launched on ({handle: Stream}); the finalizer gets the dict itself, so what
        # __call__ / replay() add later is seen when the runner is dropped
        self._replay_streams = {self.stream.cuda_stream: self.stream}
        weakref.finalize(self, _release_workspace, weakref.ref(engine), self._ws_token, self._replay_streams)
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(2):                              # warm-up: allocates the workspaces, primes caches
                    engine.run(self.imgs, self.proj_stack, self.depth_min, self.depth_max, composed=self.composed)
                torch.cuda.synchronize(imgs.device)
                first = ops.profile_graph_count()
                self.graph = torch.cuda.CUDAGraph()
                # finalizers of dropped runners / graphs must not run in the middle of the capture (a destroyed hipGraph or any
                # synchronising call from the collector invalidates it): collect now, keep the collector off until the end
                gc.collect()
                gc_was_on = gc.isenabled()
                gc.disable()
                try:
                    # thread_local: calls other host threads make meanwhile (scan_dataset.Prefetcher's decoder pins the next
                    # sample = hipHostMalloc) are legal and leave this capture alone; in the default global mode such a call
                    # invalidated the capture now and then ("operation not permitted when stream is capturing")
                    self.graph.capture_begin(capture_error_mode="thread_local")
                    self.out = engine.run(self.imgs, self.proj_stack, self.depth_min, self.depth_max, composed=self.composed)
                    self.graph.capture_end()
                finally:
                    if gc_was_on:
                        gc.enable()
                self.profile_pairs = (first, ops.profile_graph_count() - first)
        finally:
            engine._ws_owner = None
        torch.cuda.synchronize(imgs.device)

    @property
    def static_inputs(self):
        """(imgs, projs, depth_min, depth_max) buffers the graph reads: a producer (data loader, H2D copy)
        may fill them in place and pass them back to ``__call__``, which then skips its staging copies."""
        return self.imgs, self.projs, self.depth_min, self.depth_max

    def _stage_composed(self, host: Tensor) -> None:
        """CPU [3,B,S,12] -> the static device buffer, in stream order, without synchronising the stream: a ring of pinned
        buffers, each reused only after the copy that last read it has completed (a host wait on a long-finished event)"""
        i = self._pin_i
        self._pin_i = (i + 1) % len(self._pin)
        if self._pin_ev[i] is not None:
            self._pin_ev[i].synchronize()
        self._pin[i].copy_(host)
        self.composed.copy_(self._pin[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pin_ev[i] = ev

    def __call__(self, imgs: Tensor, projs, depth_min: Tensor, depth_max: Tensor):
        """copy the sample into the static inputs (unless it already IS them) and replay ON THE CURRENT
        STREAM; returns the static (depth, confidence) buffers, valid in stream order like any other torch op.
        A runner captured with ``composed=`` takes the composed projections (CPU [3,B,S,12]) as ``projs``."""
        if self.composed is not None:
            self._stage_composed(projs)
            projs = {}
        dst, src = [], []
        for d, t in [(self.imgs, imgs), (self.depth_min, depth_min), (self.depth_max, depth_max)] + \
                    [(self.projs[l], projs[l]) for l in self.projs]:
            if t is not d:
                dst.append(d)
                src.append(t)
        if dst:
            ok = all(t.is_cuda and t.is_contiguous() and t.dtype == d.dtype and t.shape == d.shape for d, t in zip(dst, src))
            if ok:
                ops.copy_multi(dst, src)            # images + cameras + depth range: ONE launch
            else:
                for d, t in zip(dst, src):
                    d.copy_(t, non_blocking=True)
        return self.replay()

    def replay(self):
        """launch the captured graph on the current stream as it is (static inputs already hold the sample)"""
        cur = torch.cuda.current_stream(self.imgs.device)
        self._replay_streams.setdefault(cur.cuda_stream, cur)
        self.graph.replay()
        return self.out
