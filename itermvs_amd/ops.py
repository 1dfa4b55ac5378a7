"""Tensor-level wrappers over the C ABI (one per entry point of include/itermvs_hip.h).

Every function takes CUDA (ROCm) float32 tensors, enqueues the HIP kernel on torch's current
stream and returns freshly allocated outputs (or writes into ``out=`` buffers).  There is no
CPU path: a CPU tensor raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import MAX_HYP, MAX_SRC, ConvParams, CorrInitParams, CorrIterParams, FMap, LevelSrc, TapParams, check

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Tensor, name: str) -> Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/ROCm tensor - the IterMVS HIP engine has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t


_FEATURE_DTYPES = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
FEATURE_DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def _feat(t: Tensor, name: str) -> int:
    """a feature map (fp32, or fp16 / bf16 STORAGE -- the kernels compute in fp32) -> its itermvs_dtype"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/ROCm tensor - the IterMVS HIP engine has no CPU path")
    if t.dtype not in _FEATURE_DTYPES:
        raise RuntimeError(f"{name}: feature maps are float32, float16 or bfloat16, got {t.dtype}")
    return _FEATURE_DTYPES[t.dtype]


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def fmap(t: Tensor, name: str = "fmap") -> FMap:
    """Describe a [B,C,H,W] tensor (any strides; fp32 or 16-bit feature storage) as an ``itermvs_fmap``."""
    dt = _feat(t, name)
    if t.dim() != 4:
        raise RuntimeError(f"{name}: expected [B,C,H,W]")
    sb, sc, sy, sx = t.stride()
    return FMap(t.data_ptr(), sb, sc, sy, sx, t.shape[1], t.shape[2], t.shape[3], dt)


def level_src(views: Sequence[Tensor], name: str = "src") -> LevelSrc:
    """S source-view maps [B,C,H1,W1] sharing shape and strides -> ``itermvs_level_src``."""
    if not 1 <= len(views) <= MAX_SRC:
        raise RuntimeError(f"{name}: need 1..{MAX_SRC} source views, got {len(views)}")
    v0 = views[0]
    ls = LevelSrc()
    ls.dtype = _feat(v0, name)
    for i, v in enumerate(views):
        if _feat(v, name) != ls.dtype or v.shape != v0.shape or v.stride() != v0.stride():
            raise RuntimeError(f"{name}: all source views must share dtype, shape and strides")
        ls.view[i] = v.data_ptr()
    ls.sb, ls.sc, ls.sy, ls.sx = v0.stride()
    ls.C, ls.H, ls.W = v0.shape[1], v0.shape[2], v0.shape[3]
    return ls


def channels_last(t: Tensor) -> Tensor:
    """Return ``t`` in channels-last memory format (no copy if it already is)."""
    return t.contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------
def compose_proj(mats: Tensor, nan_flag: Optional[Tensor] = None, depth_range=None):
    """module.py:77-90.  mats [n,V,4,4] (view 0 = reference) -> [n,V-1,12] rows of [rot|trans].
    ``depth_range`` = (depth_min [B], depth_max [B]): the same launch also returns (1/depth_min, 1/depth_max)
    (itermvs.py:240-241); the result is then (proj, inv_min, inv_max)."""
    mats = _dev(mats, "mats").contiguous()
    n, v = mats.shape[0], mats.shape[1]
    out = torch.empty((n, v - 1, 12), device=mats.device, dtype=torch.float32)
    dmin = dmax = imin = imax = None
    nb = 0
    if depth_range is not None:
        dmin, dmax = (_dev(t, "depth range").contiguous() for t in depth_range)
        nb = dmin.numel()
        imin, imax = torch.empty_like(dmin), torch.empty_like(dmax)
    check(_lib.load().itermvs_compose_proj(mats.data_ptr(), n, v, out.data_ptr(), _ptr(nan_flag), _ptr(dmin), _ptr(dmax),
                                           nb, _ptr(imin), _ptr(imax), _stream()), "itermvs_compose_proj")
    return out if depth_range is None else (out, imin, imax)


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, proj12, depth, want_mask):
        b, c, h1, w1 = src.shape
        _, n, h, w = depth.shape
        depth = depth.contiguous()
        out = torch.empty((b, c, n, h, w), device=src.device, dtype=torch.float32)
        mask = torch.empty((b, n, h, w), device=src.device, dtype=torch.uint8) if want_mask else None
        fm = fmap(src, "src_fea")
        check(_lib.load().itermvs_warp(C.byref(fm), proj12.data_ptr(), depth.data_ptr(), b, n, h, w, out.data_ptr(),
                                       _ptr(mask), _stream()), "itermvs_warp")
        ctx.save_for_backward(proj12, depth)
        ctx.src_shape = (b, c, h1, w1)
        ctx.mark_non_differentiable(*([mask] if want_mask else []))
        return (out, mask) if want_mask else out

    @staticmethod
    def backward(ctx, gout, *_):
        proj12, depth = ctx.saved_tensors
        b, c, h1, w1 = ctx.src_shape
        _, n, h, w = depth.shape
        gsrc = torch.zeros((b, c, h1, w1), device=gout.device, dtype=torch.float32)
        gout = gout.contiguous()
        check(_lib.load().itermvs_warp_backward(gout.data_ptr(), proj12.data_ptr(), depth.data_ptr(), b, c, n, h, w,
                                                h1, w1, gsrc.data_ptr(), _stream()), "itermvs_warp_backward")
        return gsrc, None, None, None


def warp(src_fea: Tensor, proj12: Tensor, depth_samples: Tensor, return_mask: bool = False):
    """module.py:68-125 on a pre-composed projection [B,12]: -> [B,C,N,H,W] (and bool mask)."""
    _dev(src_fea, "src_fea"); _dev(depth_samples, "depth_samples")
    proj12 = _dev(proj12, "proj").contiguous()
    res = _WarpFn.apply(src_fea, proj12, depth_samples, return_mask)
    if return_mask:
        return res[0], res[1].bool()
    return res


def ref_quarter(ref1: Tensor, ref2: Tensor, ref3: Tensor) -> Tensor:
    """itermvs.py:95-98: reference features of levels 1..3 on the 1/4 grid, [B,H,W,C1+C2+C3]."""
    b, _, h, w = ref2.shape
    cq = ref1.shape[1] + ref2.shape[1] + ref3.shape[1]
    out = torch.empty((b, h, w, cq), device=ref2.device, dtype=torch.float32)
    f1, f2, f3 = fmap(ref1, "ref1"), fmap(ref2, "ref2"), fmap(ref3, "ref3")
    check(_lib.load().itermvs_ref_quarter(C.byref(f1), C.byref(f2), C.byref(f3), b, out.data_ptr(), _stream()),
          "itermvs_ref_quarter")
    return out


def copy_multi(dst: Sequence[Tensor], src: Sequence[Tensor]) -> None:
    """dst[i].copy_(src[i]) for up to 8 dense tensor pairs of equal dtype and shape in ONE launch (itermvs_copy_multi)"""
    n = len(dst)
    if n == 0:
        return
    if n > 8 or len(src) != n:
        raise RuntimeError("copy_multi: 1..8 tensor pairs")
    for d, t in zip(dst, src):
        if not (isinstance(d, torch.Tensor) and isinstance(t, torch.Tensor) and d.is_cuda and t.is_cuda):
            raise RuntimeError("copy_multi: expected CUDA/ROCm tensors - the IterMVS HIP engine has no CPU path")
        if d.dtype != t.dtype or d.shape != t.shape or not d.is_contiguous() or not t.is_contiguous():
            raise RuntimeError("copy_multi: pairs must be dense tensors of equal dtype and shape")
    sp = (C.c_void_p * n)(*[t.data_ptr() for t in src])
    dp = (C.c_void_p * n)(*[d.data_ptr() for d in dst])
    nb = (C.c_int64 * n)(*[d.numel() * d.element_size() for d in dst])
    check(_lib.load().itermvs_copy_multi(sp, dp, nb, n, _stream()), "itermvs_copy_multi")


def box_probe(sink: Tensor, blocks: int, iters: int, clocks: Optional[Tensor] = None) -> None:
    """itermvs_box_probe: the fixed fp32-MFMA issue loop of bench.py's box calibration (sink: float32[blocks*256];
    clocks: int64[2] -> shader-clock ticks, 100 MHz ticks of workgroup 0)"""
    if not sink.is_cuda or sink.dtype != torch.float32 or sink.numel() < blocks * 256:
        raise RuntimeError("box_probe: sink must be a float32 CUDA/ROCm tensor of blocks*256 elements")
    check(_lib.load().itermvs_box_probe(sink.data_ptr(), blocks, iters, _ptr(clocks), _stream()), "itermvs_box_probe")


def box_chase(ring: Tensor, steps: int, start: int = 0):
    """itermvs_box_chase: (nanoseconds per dependent load along ``ring`` -- an int32 device tensor of indices forming one cycle --,
    shader clock in MHz during the walk, index where it ended)"""
    if not ring.is_cuda or ring.dtype != torch.int32:
        raise RuntimeError("box_chase: ring must be an int32 CUDA/ROCm tensor")
    out = torch.zeros((1,), device=ring.device, dtype=torch.int32)
    clocks = torch.zeros((2,), device=ring.device, dtype=torch.int64)
    check(_lib.load().itermvs_box_chase(ring.data_ptr(), start, steps, out.data_ptr(), clocks.data_ptr(), _stream()), "itermvs_box_chase")
    c = clocks.tolist()
    return float(c[0]) * 10.0 / steps, float(c[1]) / max(c[0], 1) * 100.0, int(out.item())


def clock_stamp(out: Tensor) -> None:
    """itermvs_clock_stamp: every XCD writes {100 MHz counter, shader-clock counter} to out[XCC_ID] (int64[16, 2], zeroed)"""
    if not out.is_cuda or out.dtype != torch.int64 or out.numel() != 32:
        raise RuntimeError("clock_stamp: out must be an int64 CUDA/ROCm tensor [16, 2]")
    check(_lib.load().itermvs_clock_stamp(out.data_ptr(), _stream()), "itermvs_clock_stamp")


def ref_quarter_compose(ref1: Tensor, ref2: Tensor, ref3: Tensor, mats: Tensor, nan_flag: Optional[Tensor], depth_range):
    """ref_quarter and compose_proj (with the inverse depth range) in ONE launch -- two independent pieces of work that both
    precede the correlation kernels.  Returns (ref_q, proj, inv_min, inv_max)."""
    b, _, h, w = ref2.shape
    cq = ref1.shape[1] + ref2.shape[1] + ref3.shape[1]
    out = torch.empty((b, h, w, cq), device=ref2.device, dtype=torch.float32)
    f1, f2, f3 = fmap(ref1, "ref1"), fmap(ref2, "ref2"), fmap(ref3, "ref3")
    mats = _dev(mats, "mats").contiguous()
    n, v = mats.shape[0], mats.shape[1]
    proj = torch.empty((n, v - 1, 12), device=mats.device, dtype=torch.float32)
    dmin, dmax = (_dev(t, "depth range").contiguous() for t in depth_range)
    imin, imax = torch.empty_like(dmin), torch.empty_like(dmax)
    check(_lib.load().itermvs_ref_quarter_compose(C.byref(f1), C.byref(f2), C.byref(f3), b, out.data_ptr(), mats.data_ptr(), n, v,
                                                  proj.data_ptr(), _ptr(nan_flag), dmin.data_ptr(), dmax.data_ptr(), dmin.numel(),
                                                  imin.data_ptr(), imax.data_ptr(), _stream()), "itermvs_ref_quarter_compose")
    return out, proj, imin, imax


def corr_iter(src: Dict[int, Sequence[Tensor]], ref_q: Tensor, proj: Tensor, view_w: Tensor,
              inv_min: Tensor, inv_max: Tensor, depth: Optional[Dict[int, Tensor]] = None,
              norm_depth: Optional[Tensor] = None, offsets: Optional[Dict[int, Sequence[float]]] = None,
              out: Optional[List[Tensor]] = None, timed: bool = True) -> List[Tensor]:
    """itermvs.py:84-120 fused (see include/itermvs_hip.h).  ``src[l]`` = S channels-last maps of
    level l; ``proj`` [3,B,S,12]; ``view_w`` [B,S,H,W]; hypotheses either explicit
    ``depth[l]`` [B,N_l,H,W] or generated from ``norm_depth`` [B,1,H,W] + ``offsets[l]``.
    Returns the three aggregated group correlations [B,N_l,8,H,W]."""
    b, h, w, _ = ref_q.shape
    s = len(src[1])
    p = CorrIterParams()
    p.B, p.S, p.H, p.W = b, s, h, w
    keep = []
    outs: List[Tensor] = []
    for i, l in enumerate((1, 2, 3)):
        p.src[i] = level_src(src[l], f"src level {l}")
        if depth is not None and depth.get(l) is not None:
            d = _dev(depth[l], "depth").contiguous()
            keep.append(d)
            n = d.shape[1]
            p.depth[i] = d.data_ptr()
        else:
            if offsets is None or norm_depth is None:
                raise RuntimeError("corr_iter: give either depth[l] or norm_depth + offsets[l]")
            n = len(offsets[l])
            p.depth[i] = None
            for k, o in enumerate(offsets[l]):
                p.offsets[i][k] = o
        if n > MAX_HYP:
            raise RuntimeError(f"corr_iter: at most {MAX_HYP} hypotheses per level")
        p.N[i] = n
        o = out[i] if out is not None else torch.empty((b, n, 8, h, w), device=ref_q.device, dtype=torch.float32)
        outs.append(o)
        p.out[i] = o.data_ptr()
    if norm_depth is not None:
        # [B,1,H,W] view, possibly one channel of a wider contiguous [B,Ct,H,W] buffer
        _dev(norm_depth, "norm_depth")
        if norm_depth.stride(3) != 1 or norm_depth.stride(2) != w:
            norm_depth = norm_depth.contiguous()
        p.norm_depth = norm_depth.data_ptr()
        p.norm_depth_sb = norm_depth.stride(0)
    proj = _dev(proj, "proj").contiguous()
    # [B,S,H,W] with any batch / view / pixel strides whose rows are dense in the pixel stride: contiguous, or the
    # interleaved [B,H,W,S] storage view_aggregate_up(interleaved=True) returns (one vector load per lane quad)
    _dev(view_w, "view_w")
    if tuple(view_w.shape) != (b, s, h, w):
        raise RuntimeError(f"corr_iter: view_w must be [B,S,H,W] = {(b, s, h, w)}, got {tuple(view_w.shape)}")
    if view_w.dtype != torch.float32 or view_w.stride(2) != w * view_w.stride(3) or min(view_w.stride()) < 1:
        view_w = view_w.float().contiguous()
    p.view_w_sb, p.view_w_ss, p.view_w_sp = view_w.stride(0), view_w.stride(1), view_w.stride(3)
    p.ref_q, p.proj, p.view_w = _dev(ref_q, "ref_q").data_ptr(), proj.data_ptr(), view_w.data_ptr()
    p.inv_depth_min, p.inv_depth_max = _dev(inv_min, "inv_min").data_ptr(), _dev(inv_max, "inv_max").data_ptr()
    lib = _lib.load()
    if not timed:               # no timing events around this launch (itermvs_profile_*): mask bit 0 off for the call
        lib.itermvs_profile_set_mask(_PROFILE_MASK[0] & ~1)
    check(lib.itermvs_corr_iter(C.byref(p), _stream()), "itermvs_corr_iter")
    if not timed:
        lib.itermvs_profile_set_mask(_PROFILE_MASK[0])
    return outs


def _corr_iter_params(src, ref_q, proj, view_w, inv_min, inv_max, norm_depth, offsets, outs):
    """parameter block of itermvs_corr_iter for hypotheses generated from ``norm_depth`` + ``offsets``; returns (block, keep-alive)"""
    b, h, w, _ = ref_q.shape
    p = CorrIterParams()
    p.B, p.S, p.H, p.W = b, len(src[1]), h, w
    for i, l in enumerate((1, 2, 3)):
        p.src[i] = level_src(src[l], f"src level {l}")
        p.depth[i] = None
        p.N[i] = len(offsets[l])
        for k, o in enumerate(offsets[l]):
            p.offsets[i][k] = o
        p.out[i] = outs[i].data_ptr() if outs is not None else None
    nd = norm_depth if (norm_depth.stride(3) == 1 and norm_depth.stride(2) == w) else norm_depth.contiguous()
    p.norm_depth, p.norm_depth_sb = nd.data_ptr(), nd.stride(0)
    p.ref_q, p.proj, p.view_w = ref_q.data_ptr(), proj.data_ptr(), view_w.data_ptr()
    p.inv_depth_min, p.inv_depth_max = inv_min.data_ptr(), inv_max.data_ptr()
    return p, nd


def _views(feat: Tensor, b: int, v: int):
    """[B*V,C,H,W] dense channels-last pyramid level -> (reference view [B,C,H,W], list of the V-1 source views), all views"""
    pv = feat.view(b, v, *feat.shape[1:])
    return pv[:, 0], [pv[:, i] for i in range(1, v)]


def _need_cl(t: Tensor, name: str) -> Tensor:
    _feat(t, name)
    if not t.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError(f"{name}: expected a dense channels-last [B*V,C,H,W] tensor")
    return t


class FeatureGradPool:
    """Shared accumulators for the feature gradients of ONE training forward.

    Every ``Evaluation`` call of a step (one initialisation + ``iteration`` iteration calls, models/itermvs.py:271-298) sends a
    gradient to the same three pyramid tensors.  As separate autograd outputs that is one zero-filled dense tensor per call
    and level plus autograd's sums (at train_dtu.sh's batch of 4: 177 MB per call, ~3 ms of element-wise kernels per step).
    With a pool the backward kernels of all calls scatter into ONE zero-filled fp32 tensor per level; ``feature_grad_sink``
    hands those to autograd once, after the last call's backward has been enqueued."""

    def __init__(self):
        self.acc = {}

    def get(self, level: int, like: Tensor) -> Tensor:
        a = self.acc.get(level)
        if a is None:
            a = self.acc[level] = torch.zeros_like(like, dtype=torch.float32)     # dense channels-last like the features
        return a


class _FeatGradSink(torch.autograd.Function):
    """identity on the three pyramid levels; its backward adds the pool's accumulators to whatever other consumers of the
    features (reference-view resampling, up-sampling head) sent.  The fused correlation Functions take THESE outputs as their
    feature inputs, so autograd runs this backward only after every one of theirs."""

    @staticmethod
    def forward(ctx, pool, f1, f2, f3):
        ctx.pool = pool
        ctx.set_materialize_grads(False)
        return f1.view_as(f1), f2.view_as(f2), f3.view_as(f3)

    @staticmethod
    def backward(ctx, g1, g2, g3):
        out = []
        for level, g in zip((1, 2, 3), (g1, g2, g3)):
            a = ctx.pool.acc.pop(level, None)
            out.append(g if a is None else (a if g is None else a.add_(g)))
        return (None,) + tuple(out)


def feature_grad_sink(pool: FeatureGradPool, feats: Dict[int, Tensor]) -> Dict[int, Tensor]:
    """route the feature gradients of every fused correlation call given ``pool=`` through ``pool`` (see FeatureGradPool)"""
    pool.acc.clear()            # a new forward: accumulators a previous (partial or failed) backward left behind are not its
    f1, f2, f3 = _FeatGradSink.apply(pool, feats[1], feats[2], feats[3])
    for t in (f1, f2, f3):
        t._itermvs_sink_pool = pool        # corr_*_train(pool=...) accepts only these tensors (_check_sink)
    return {1: f1, 2: f2, 3: f3}


def _check_sink(pool: Optional["FeatureGradPool"], feats: Sequence[Tensor], what: str) -> None:
    """``pool=`` routes the feature gradient through the pool instead of returning it: that is only sound when the features
    ARE the outputs of ``feature_grad_sink(pool, ...)`` -- otherwise the gradient would be dropped silently"""
    if pool is not None and any(getattr(t, "_itermvs_sink_pool", None) is not pool for t in feats):
        raise RuntimeError(f"{what}: pool= needs the feature tensors returned by feature_grad_sink(pool, ...)")


class _CorrIterFn(torch.autograd.Function):
    """itermvs.py:84-120 as one differentiable op: forward = itermvs_corr_iter, backward = itermvs_corr_iter_backward.
    Inputs: ref_q [B,H,W,96], proj, view_w (detached by the caller, itermvs.py:295), inverse depth range, normalised
    depth (no gradient, module.py:77), and the three channels-last pyramid levels [B*V,C_l,H_l,W_l] whose views 1..V-1 are
    the sources.  The gradient w.r.t. each level is one dense tensor the kernel scatter-adds into (view 0 stays zero).
    ``f1..f3`` are the fp32 tensors the gradient is routed to; ``s1..s3`` the tensors the kernels GATHER from -- the same
    objects for fp32 storage, their bf16 / fp16 roundings for 16-bit feature storage (BASELINE cfg 4): the gradient is
    taken at the stored values and handed to the fp32 tensors in fp32 (straight-through rounding, no 16-bit gradient)."""

    @staticmethod
    def forward(ctx, ref_q, proj, view_w, inv_min, inv_max, norm_depth, offsets, b, v, f1, f2, f3, s1, s2, s3, pool=None):
        ctx.pool = pool
        f1, f2, f3 = s1, s2, s3
        src = {l: _views(f, b, v)[1] for l, f in ((1, f1), (2, f2), (3, f3))}
        _, h, w, _ = ref_q.shape
        outs = [torch.empty((b, len(offsets[l]), 8, h, w), device=ref_q.device, dtype=torch.float32) for l in (1, 2, 3)]
        p, nd = _corr_iter_params(src, ref_q, proj, view_w, inv_min, inv_max, norm_depth, offsets, outs)
        check(_lib.load().itermvs_corr_iter(C.byref(p), _stream()), "itermvs_corr_iter")
        ctx.save_for_backward(ref_q, proj, view_w, inv_min, inv_max, nd, f1, f2, f3)
        ctx.meta = (offsets, b, v)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        ref_q, proj, view_w, inv_min, inv_max, nd, f1, f2, f3 = ctx.saved_tensors
        offsets, b, v = ctx.meta
        feats = (f1, f2, f3)
        src = {l: _views(f, b, v)[1] for l, f in zip((1, 2, 3), feats)}
        p, nd = _corr_iter_params(src, ref_q, proj, view_w, inv_min, inv_max, nd, offsets, None)
        gouts = [g.contiguous() for g in gouts]
        # dense channels-last like the features, fp32 whatever their storage type (fp32 atomics); with a pool: the step's
        # shared accumulators (FeatureGradPool), handed to autograd once by the sink
        pool = ctx.pool
        gfeat = [torch.zeros_like(f, dtype=torch.float32) if pool is None else pool.get(l, f) for l, f in zip((1, 2, 3), feats)]
        gref = torch.empty_like(ref_q)
        go = (C.c_void_p * 3)(*[g.data_ptr() for g in gouts])
        lv = [(C.c_void_p * (v - 1))(*[t.data_ptr() for t in _views(gf, b, v)[1]]) for gf in gfeat]
        gs = (C.POINTER(C.c_void_p) * 3)(*[C.cast(a, C.POINTER(C.c_void_p)) for a in lv])
        check(_lib.load().itermvs_corr_iter_backward(C.byref(p), C.byref(go), C.byref(gs), gref.data_ptr(), _stream()),
              "itermvs_corr_iter_backward")
        return (gref, None, None, None, None, None, None, None, None) + (tuple(gfeat) if pool is None else (None, None, None)) + (None, None, None, None)


def corr_iter_train(feats: Dict[int, Tensor], b: int, v: int, ref_q: Tensor, proj: Tensor, view_w: Tensor, inv_min: Tensor,
                    inv_max: Tensor, norm_depth: Tensor, offsets: Dict[int, Sequence[float]],
                    stored: Optional[Dict[int, Tensor]] = None, pool: Optional[FeatureGradPool] = None) -> Tuple[Tensor, ...]:
    """differentiable itermvs_corr_iter: ``feats[l]`` = dense channels-last fp32 [B*V,C_l,H_l,W_l] (gradient to the source
    views) and ``ref_q`` [B,H,W,96] (gradient) -> three [B,N_l,8,H,W] tensors.  ``stored[l]``: the 16-bit copies the kernels
    gather from (feature storage bf16 / fp16); default: ``feats`` themselves.  ``pool``: ``feats`` are the outputs of
    ``feature_grad_sink(pool, ...)`` and their gradient is accumulated there instead of being returned to autograd per call."""
    _check_sink(pool, [feats[l] for l in (1, 2, 3)], "corr_iter_train")
    f = [_need_cl(_dev(feats[l], f"feature level {l}"), f"feature level {l}") for l in (1, 2, 3)]
    st = f if stored is None else [_need_cl(stored[l].detach(), f"stored feature level {l}") for l in (1, 2, 3)]
    if any(a.shape != c.shape for a, c in zip(f, st)):
        raise RuntimeError("corr_iter_train: stored features must have the shapes of the fp32 features")
    return _CorrIterFn.apply(_dev(ref_q, "ref_q").contiguous(), _dev(proj, "proj").contiguous(), _dev(view_w, "view_w").contiguous(),
                             inv_min, inv_max, norm_depth.detach(), {l: tuple(offsets[l]) for l in (1, 2, 3)}, b, v, *f, *st, pool)


def _corr_init_params(src3, ref3, proj, inv_min, inv_max, n, out):
    b, _, h, w = ref3.shape
    p = CorrInitParams()
    p.B, p.S, p.H, p.W, p.N = b, len(src3), h, w, n
    p.src = level_src(src3, "src level 3")
    p.ref = fmap(ref3, "ref3")
    p.proj = proj.data_ptr()
    p.depth = None
    p.inv_depth_min, p.inv_depth_max = inv_min.data_ptr(), inv_max.data_ptr()
    p.out = out.data_ptr() if out is not None else None
    return p


class _CorrInitFn(torch.autograd.Function):
    """itermvs.py:48-51 (+ :11-19) as one differentiable op on the level-3 features [B*V,48,H,W] (channels-last):
    per-view group correlation [B,S,N,8,H,W]; gradient = one dense tensor (reference view gathered, sources scattered)"""

    @staticmethod
    def forward(ctx, f3, proj, inv_min, inv_max, n, b, v, s3, pool=None):
        ctx.pool = pool
        f3 = s3                 # the tensor the kernel reads (f3 itself, or its 16-bit rounding); gradient goes to f3 in fp32
        ref3, src3 = _views(f3, b, v)
        out = torch.empty((b, v - 1, n, 8) + tuple(f3.shape[2:]), device=f3.device, dtype=torch.float32)
        p = _corr_init_params(src3, ref3, proj, inv_min, inv_max, n, out)
        check(_lib.load().itermvs_corr_init(C.byref(p), _stream()), "itermvs_corr_init")
        ctx.save_for_backward(f3, proj, inv_min, inv_max)
        ctx.meta = (n, b, v)
        return out

    @staticmethod
    def backward(ctx, gout):
        f3, proj, inv_min, inv_max = ctx.saved_tensors
        n, b, v = ctx.meta
        ref3, src3 = _views(f3, b, v)
        p = _corr_init_params(src3, ref3, proj, inv_min, inv_max, n, None)
        gout = gout.contiguous()
        gf = torch.zeros_like(f3, dtype=torch.float32) if ctx.pool is None else ctx.pool.get(3, f3)
        gref, gsrc = _views(gf, b, v)
        ptrs = (C.c_void_p * (v - 1))(*[t.data_ptr() for t in gsrc])
        check(_lib.load().itermvs_corr_init_backward(C.byref(p), gout.data_ptr(), ptrs, gref.data_ptr(), _stream()),
              "itermvs_corr_init_backward")
        return (gf if ctx.pool is None else None), None, None, None, None, None, None, None, None


def corr_init_train(f3: Tensor, b: int, v: int, proj: Tensor, inv_min: Tensor, inv_max: Tensor, num_samples: int = 32,
                    stored: Optional[Tensor] = None, pool: Optional[FeatureGradPool] = None) -> Tensor:
    """differentiable itermvs_corr_init on the dense channels-last fp32 level-3 features [B*V,48,H,W]: -> [B,S,N,8,H,W];
    ``stored``: the bf16 / fp16 copy the kernel reads (16-bit feature storage), default ``f3`` itself"""
    _check_sink(pool, [f3], "corr_init_train")
    f3 = _need_cl(_dev(f3, "feature level 3"), "feature level 3")
    s3 = f3 if stored is None else _need_cl(stored.detach(), "stored feature level 3")
    return _CorrInitFn.apply(f3, _dev(proj, "proj").contiguous(), inv_min, inv_max, num_samples, b, v, s3, pool)


class _BnReluTrainFn(torch.autograd.Function):
    """torch.nn.BatchNorm2d in train() mode (+ ReLU) as one op on the HIP kernels of csrc/bn.hip -- the ConvBnReLU / ConvBn
    layers of FeatureNet under training (models/module.py:33-50): batch statistics, running stats updated in place, the
    ReLU mask recomputed from x in the backward (nothing but x, two [C] vectors and the affine parameters is saved)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        lib = _lib.load()
        ws = torch.empty((check_count(lib.itermvs_bn_workspace_floats(n, c, hw), "itermvs_bn_workspace_floats"),), device=x.device,
                         dtype=torch.float32)
        y = torch.empty_like(x)
        mean = torch.empty((c,), device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        check(lib.itermvs_bn_train_forward(x.data_ptr(), y.data_ptr(), n, c, hw, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                           float(momentum), int(relu), _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                                           invstd.data_ptr(), ws.data_ptr(), _stream()), "itermvs_bn_train_forward")
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = int(relu)
        for t in (running_mean, running_var):       # written by the kernel: tell autograd's version counters
            if t is not None:                       # (itermvs_amd.net invalidates its packed inference engine on them)
                torch.autograd.graph.increment_version(t)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        lib = _lib.load()
        dy = dy.contiguous()
        ws = torch.empty((check_count(lib.itermvs_bn_workspace_floats(n, c, hw), "itermvs_bn_workspace_floats"),), device=x.device,
                         dtype=torch.float32)
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        check(lib.itermvs_bn_train_backward(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c, hw, gamma.data_ptr(), beta.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), ctx.relu, dgamma.data_ptr(), dbeta.data_ptr(),
                                            ws.data_ptr(), _stream()), "itermvs_bn_train_backward")
        return dx, dgamma, dbeta, None, None, None, None, None


def check_count(value: int, what: str) -> int:
    """a C entry point that returns a non-negative count or a negative error code"""
    if value < 0:
        check(value, what)
    return value


def bn_relu_train(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
                  eps: float = 1e-5, momentum: float = 0.1, relu: bool = True) -> Tensor:
    """``relu(F.batch_norm(x, running_mean, running_var, gamma, beta, training=True, momentum, eps))`` (``relu=False``: without
    the ReLU) on dense NCHW fp32 ``x``; differentiable w.r.t. x, gamma, beta; the running statistics are updated in place."""
    x = _dev(x, "x")
    if x.dim() < 2:
        raise RuntimeError("bn_relu_train: expected [N,C,...]")
    for t, name in ((gamma, "gamma"), (beta, "beta")):
        if _dev(t, name).shape != (x.shape[1],):
            raise RuntimeError(f"bn_relu_train: {name} must have shape [{x.shape[1]}]")
    for t, name in ((running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None and (_dev(t, name).shape != (x.shape[1],) or not t.is_contiguous()):
            raise RuntimeError(f"bn_relu_train: {name} must be a contiguous [{x.shape[1]}] tensor")
    return _BnReluTrainFn.apply(x.contiguous(), gamma.contiguous(), beta.contiguous(), running_mean, running_var, eps, momentum, relu)


def corr_iter_kernel_name() -> str:
    """name of the device kernel itermvs_corr_iter launches (rocprofv3's Kernel_Name contains it)"""
    return "corr_iter_kernel"


def corr_init(src3: Sequence[Tensor], ref3: Tensor, proj: Tensor, inv_min: Tensor, inv_max: Tensor,
              num_samples: int = 32, depth: Optional[Tensor] = None, out: Optional[Tensor] = None,
              timed: bool = True, groups_last: bool = False) -> Tensor:
    """itermvs.py:48-51 (+ :11-19): per-view group correlation, [B,S,N,8,H,W].  ``groups_last``: STORED [B,S,N,H,W,8] and returned
    as the [B,S,N,8,H,W] view of that storage -- as [B*S*N,8,H,W] it is a channels-last tensor, which conv2d's 8-channel bf16x3
    layer stages with two 16-byte loads per pixel, and view_aggregate(_up) reads it as it is."""
    b, _, h, w = ref3.shape
    s = len(src3)
    p = CorrInitParams()
    p.B, p.S, p.H, p.W, p.N = b, s, h, w, num_samples
    p.src = level_src(src3, "src level 3")
    p.ref = fmap(ref3, "ref3")
    proj = _dev(proj, "proj").contiguous()
    p.proj = proj.data_ptr()
    if depth is not None:
        depth = _dev(depth, "depth").contiguous()
        p.N = depth.shape[1]
        p.depth = depth.data_ptr()
    p.inv_depth_min, p.inv_depth_max = _dev(inv_min, "inv_min").data_ptr(), _dev(inv_max, "inv_max").data_ptr()
    if out is None:
        if groups_last:
            out = torch.empty((b, s, p.N, h, w, 8), device=ref3.device, dtype=torch.float32).permute(0, 1, 2, 5, 3, 4)
        else:
            out = torch.empty((b, s, p.N, 8, h, w), device=ref3.device, dtype=torch.float32)
    if tuple(out.shape) != (b, s, p.N, 8, h, w) or out.dtype != torch.float32:
        raise RuntimeError(f"corr_init: out must be float32 [B,S,N,8,H,W] = {(b, s, p.N, 8, h, w)}")
    if out.is_contiguous():
        p.out_layout = 0
    elif out.permute(0, 1, 2, 4, 5, 3).is_contiguous():
        p.out_layout = 1
    else:
        raise RuntimeError("corr_init: out must be contiguous [B,S,N,8,H,W] or the view of contiguous [B,S,N,H,W,8] storage")
    p.out = out.data_ptr()
    lib = _lib.load()
    if not timed:               # no timing events around this launch: mask bit 1 off for the call
        lib.itermvs_profile_set_mask(_PROFILE_MASK[0] & ~2)
    check(lib.itermvs_corr_init(C.byref(p), _stream()), "itermvs_corr_init")
    if not timed:
        lib.itermvs_profile_set_mask(_PROFILE_MASK[0])
    return out


def tap_indices(proj: Tensor, inv_min: Tensor, inv_max: Tensor, grid_hw: Tuple[int, int], src_hw: Tuple[int, int], *,
                depth: Optional[Tensor] = None, norm_depth: Optional[Tensor] = None, offsets: Sequence[float] = (),
                init_samples: int = 0, want_coords: bool = False):
    """Diagnostic (itermvs_tap_indices): floor(ix), floor(iy) and the validity bits of the bilinear footprints exactly as
    the fused correlation kernels compute them (module.py:99-115 + grid_sample's floor).  proj [B,S,12]; hypotheses from
    ``depth`` [B,N,h,w], or ``init_samples`` initial planes, or ``norm_depth`` [B,1,h,w] + ``offsets``.
    -> int32 [B,S,N,3,h,w] (x0, y0, bits)  (+ float [B,S,N,2,h,w] coordinates with ``want_coords``)."""
    proj = _dev(proj, "proj").contiguous()
    b, s = proj.shape[0], proj.shape[1]
    h, w = grid_hw
    p = TapParams()
    p.B, p.S, p.H, p.W, p.H1, p.W1 = b, s, h, w, src_hw[0], src_hw[1]
    p.proj = proj.data_ptr()
    if depth is not None:
        depth = _dev(depth, "depth").contiguous()
        p.N = depth.shape[1]
        p.depth = depth.data_ptr()
    elif init_samples:
        p.N, p.init = init_samples, 1
    else:
        nd = _dev(norm_depth, "norm_depth")
        if nd.stride(-1) != 1 or nd.stride(-2) != w:
            nd = nd.contiguous()
        p.N = len(offsets)
        p.norm_depth, p.norm_depth_sb = nd.data_ptr(), nd.stride(0)
        for i, o in enumerate(offsets):
            p.offsets[i] = o
    p.inv_depth_min, p.inv_depth_max = _dev(inv_min, "inv_min").data_ptr(), _dev(inv_max, "inv_max").data_ptr()
    out = torch.empty((b, s, p.N, 3, h, w), device=proj.device, dtype=torch.int32)
    coords = torch.empty((b, s, p.N, 2, h, w), device=proj.device, dtype=torch.float32) if want_coords else None
    p.out, p.coords = out.data_ptr(), _ptr(coords)
    check(_lib.load().itermvs_tap_indices(C.byref(p), _stream()), "itermvs_tap_indices")
    return (out, coords) if want_coords else out


def view_aggregate(corr: Tensor, w: Tensor) -> Tensor:
    """itermvs.py:59-69.  corr [B,S,N,8,H,W], w [B,S,H,W] -> [B,N,8,H,W]."""
    b, s, n, g, h, wd = corr.shape
    corr = _dev(corr, "corr").contiguous()
    w = _dev(w, "w").contiguous()
    out = torch.empty((b, n, g, h, wd), device=corr.device, dtype=torch.float32)
    check(_lib.load().itermvs_view_aggregate(corr.data_ptr(), w.data_ptr(), s, b, n, h * wd, out.data_ptr(), _stream()),
          "itermvs_view_aggregate")
    return out


def view_aggregate_up(corr: Tensor, w: Tensor, interleaved: bool = False) -> Tuple[Tensor, Tensor]:
    """view_aggregate and, in the same launch, the x2 bilinear up-sampling of the view weights (itermvs.py:56-57,71):
    corr [B,S,N,8,H,W], w [B,S,H,W] -> ([B,N,8,H,W], [B,S,2H,2W]).  ``interleaved``: the up-sampled weights are STORED
    [B,2H,2W,S] and returned as the permuted [B,S,2H,2W] view of that storage -- same values, the layout corr_iter reads with
    one vector load per lane quad.  ``corr`` may be corr_init(..., groups_last=True)'s view of [B,S,N,H,W,8] storage."""
    b, s, n, g, h, wd = corr.shape
    _dev(corr, "corr")
    corr_layout = 1 if (g == 8 and not corr.is_contiguous() and corr.permute(0, 1, 2, 4, 5, 3).is_contiguous()) else 0
    if corr_layout == 0:
        corr = corr.contiguous()
    w = _dev(w, "w").contiguous()
    out = torch.empty((b, n, g, h, wd), device=corr.device, dtype=torch.float32)
    if interleaved:
        store = torch.empty((b, 2 * h, 2 * wd, s), device=corr.device, dtype=torch.float32)
        w_up = store.permute(0, 3, 1, 2)
    else:
        store = w_up = torch.empty((b, s, 2 * h, 2 * wd), device=corr.device, dtype=torch.float32)
    check(_lib.load().itermvs_view_aggregate_up(corr.data_ptr(), corr_layout, w.data_ptr(), s, b, n, h, wd, out.data_ptr(), store.data_ptr(),
                                                1 if interleaved else 0, _stream()), "itermvs_view_aggregate_up")
    return out, w_up


def pvw_tail(x: Tensor, weight: Tensor, bias: Optional[Tensor], n_hyp: int) -> Tensor:
    """itermvs.py:343-348 fused: x [M*N,16,h,w] (after the 3x3 layer + ReLU), weight [1,16,1,1] -> [M,1,h,w]."""
    _dev(x, "x")
    mn, c, h, w = x.shape
    if mn % n_hyp or not x.is_contiguous():
        raise RuntimeError("pvw_tail: x must be contiguous [M*N,C,H,W]")
    m = mn // n_hyp
    out = torch.empty((m, 1, h, w), device=x.device, dtype=torch.float32)
    wv = _dev(weight, "weight").reshape(-1).contiguous()
    check(_lib.load().itermvs_pvw_tail(x.data_ptr(), wv.data_ptr(), _ptr(bias), m, n_hyp, c, h * w, out.data_ptr(), _stream()),
          "itermvs_pvw_tail")
    return out


def softmax_max(x: Tensor) -> Tensor:
    """itermvs.py:347-348.  x [M,N,H,W] -> max over N of softmax over N, [M,1,H,W]."""
    x = _dev(x, "x").contiguous()
    m, n, h, w = x.shape
    out = torch.empty((m, 1, h, w), device=x.device, dtype=torch.float32)
    check(_lib.load().itermvs_softmax_max(x.data_ptr(), m, n, h * w, out.data_ptr(), _stream()), "itermvs_softmax_max")
    return out


def pack_head_weights(w1: Tensor, w2: Tensor) -> Tuple[Tensor, Tensor]:
    """depth_head[2].weight [64,32,1,1], depth_head[4].weight [256,64,1,1] -> the matrix-core operand
    layouts of itermvs_head_regress (include/itermvs_hip.h)."""
    a1 = w1.reshape(4, 16, 2, 4, 4).permute(0, 2, 3, 1, 4).contiguous()     # [mb,i,u,q,s] -> [mb,u,q,i,s]
    a2 = w2.reshape(16, 16, 4, 4, 4).permute(0, 2, 3, 1, 4).contiguous()    # [mb,i,m,q,r] -> [mb,m,q,i,r]
    return a1, a2


def pack_head_w2_split3(w2: Tensor) -> Tensor:
    """depth_head[4].weight [256,64,1,1] -> the bf16x3 A operands of itermvs_head_fused's 64 -> 256 layer (w2_format 3):
    bfloat16 [16][2][3][64][8] whose element (ob, g, p, lane = 16 q + i, j) is term p (h, m, l of ``split_bf16x3``) of
    W2[ob*16 + i][(2g + j//4)*16 + 4q + j%4]."""
    if w2.numel() != 256 * 64:
        raise RuntimeError("pack_head_w2_split3: expects the [256,64,1,1] weight")
    t = w2.reshape(16, 16, 2, 2, 4, 4).float()                    # [ob, i, g, jj, q, r]: channel (2g + jj)*16 + 4q + r
    t = t.permute(0, 2, 4, 1, 3, 5).reshape(16, 2, 64, 8)         # [ob, g, (q, i), (jj, r)]
    return torch.stack(split_bf16x3(t), 2).contiguous()           # [16, 2, 3, 64, 8]


def pack_head_w0_split3(w0: Tensor) -> Tensor:
    """depth_head[0].weight [32,32,3,3] (dilated 3x3) -> the bf16x3 A operands of itermvs_head_fused's first layer (w0_format 3):
    bfloat16 [2][9][3][64][8] whose element (mb, tap, p, lane = 16 q + i, j) is term p of W0[16 mb + i][(j//4)*16 + 4q + j%4][tap]."""
    if tuple(w0.shape) != (32, 32, 3, 3):
        raise RuntimeError("pack_head_w0_split3: expects the [32,32,3,3] weight")
    t = w0.float().reshape(2, 16, 2, 4, 4, 9).permute(0, 5, 3, 1, 2, 4).reshape(2, 9, 64, 8)      # [mb, tap, (q, i), (jj, r)]
    return torch.stack(split_bf16x3(t), 2).contiguous()                                            # [2, 9, 3, 64, 8]


def pack_conv1x1_operand(w1: Tensor, bias: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """Conv2d(64, NO, 1) weight [NO,64,1,1] (+ bias [NO]) -> the matrix-core operand layout of itermvs_conv3x3_conv1x1:
    [NOB][4][4][16][4] with element (ob,m,q,i,r) = W1[ob*16+i][m*16+q*4+r], NO zero-padded to NOB*16 (bias likewise)."""
    no, ci = w1.shape[0], w1.shape[1]
    if ci != 64 or w1.numel() != no * 64:
        raise RuntimeError("pack_conv1x1_operand: expects a [NO,64,1,1] weight")
    nob = (no + 15) // 16
    full = torch.zeros((nob * 16, 64), device=w1.device, dtype=torch.float32)
    full[:no] = w1.reshape(no, 64).float()
    a = full.reshape(nob, 16, 4, 4, 4).permute(0, 2, 3, 1, 4).contiguous()     # [ob,i,m,q,r] -> [ob,m,q,i,r]
    bp = None
    if bias is not None:
        bp = torch.zeros((nob * 16,), device=w1.device, dtype=torch.float32)
        bp[:no] = bias.float()
    return a, bp


def pack_conv3x3_conv1x1_split3(w0: Tensor, w1: Tensor, bias: Optional[Tensor] = None):
    """Conv2d(32, 64, 3) weight [64,32,3,3] and Conv2d(64, NO, 1) weight [NO,64,1,1] (+ bias) -> the bf16x3 operands of
    itermvs_conv3x3_conv1x1 (weight_format 3): (bf16 [4][9][3][64][8], bf16 [NOB][2][3][64][8], fp32 bias [NOB*16] | None).  A lane
    (q, i) holds, as its eight K slots j of a 32-channel group, the channels (j // 4) * 16 + 4 q + j % 4."""
    if tuple(w0.shape) != (64, 32, 3, 3) or w1.shape[1] != 64 or w1.numel() != w1.shape[0] * 64:
        raise RuntimeError("pack_conv3x3_conv1x1_split3: expects [64,32,3,3] and [NO,64,1,1] weights")
    t0 = w0.float().reshape(4, 16, 2, 4, 4, 9)                       # [w, i, jj, q, r, tap]: input channel jj*16 + 4q + r
    t0 = t0.permute(0, 5, 3, 1, 2, 4).reshape(4, 9, 64, 8)           # [w, tap, (q, i), (jj, r)]
    a0 = torch.stack(split_bf16x3(t0), 2).contiguous()               # [4, 9, 3, 64, 8]
    no = w1.shape[0]
    nob = (no + 15) // 16
    full = torch.zeros((nob * 16, 64), device=w1.device, dtype=torch.float32)
    full[:no] = w1.reshape(no, 64).float()
    t1 = full.reshape(nob, 16, 2, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(nob, 2, 64, 8)      # [ob, g, (q, i), (jj, r)]
    a1 = torch.stack(split_bf16x3(t1), 2).contiguous()               # [nob, 2, 3, 64, 8]
    bp = None
    if bias is not None:
        bp = torch.zeros((nob * 16,), device=w1.device, dtype=torch.float32)
        bp[:no] = bias.float()
    return a0, a1, bp


def conv3x3_conv1x1(x: Tensor, w0, w1p: Tensor, bias_p: Optional[Tensor], no: int, out: Optional[Tensor] = None) -> Tensor:
    """itermvs_conv3x3_conv1x1: conv3x3 32 -> 64 + ReLU + conv1x1 64 -> ``no`` in one launch; x [B,32,H,W] -> [B,no,H,W].
    ``w0``: the 3x3 layer's MfmaWeight (fp32 tile format) with ``w1p`` / ``bias_p`` from pack_conv1x1_operand (exact fp32 matrix
    instruction), or both bfloat16 tensors of pack_conv3x3_conv1x1_split3 (bf16x3 arithmetic)."""
    ptr, sb = _planes(x, "conv3x3_conv1x1 input")
    b, c, h, w = x.shape
    nob = (no + 15) // 16
    if isinstance(w0, Tensor):
        if not (w0.is_cuda and w1p.is_cuda and w0.dtype == w1p.dtype == torch.bfloat16 and tuple(w0.shape) == (4, 9, 3, 64, 8)
                and tuple(w1p.shape) == (nob, 2, 3, 64, 8) and w0.is_contiguous() and w1p.is_contiguous()) or c != 32:
            raise RuntimeError("conv3x3_conv1x1: bfloat16 weights must come from pack_conv3x3_conv1x1_split3 for this output width")
        if bias_p is not None and bias_p.numel() != nob * 16:
            raise RuntimeError("conv3x3_conv1x1: bias_p must come from pack_conv3x3_conv1x1_split3")
        if out is None:
            out = torch.empty((b, no, h, w), device=x.device, dtype=torch.float32)
        elif tuple(out.shape) != (b, no, h, w):
            raise RuntimeError(f"conv3x3_conv1x1: output has shape {tuple(out.shape)}, expected {(b, no, h, w)}")
        po, so = _planes(out, "conv3x3_conv1x1 output")
        check(_lib.load().itermvs_conv3x3_conv1x1(ptr, sb, b, h, w, w0.data_ptr(), w1p.data_ptr(), 3, _ptr(bias_p), no, po, so, _stream()),
              "itermvs_conv3x3_conv1x1")
        return out
    if c != 32 or w0.tile is None or w0.cin != 32 or w0.cout != 64 or w0.ksize != 3:
        raise RuntimeError("conv3x3_conv1x1: expects a 32-channel input and the 32 -> 64 3x3 weight")
    if w1p.numel() != nob * 1024 or (bias_p is not None and bias_p.numel() != nob * 16):
        raise RuntimeError("conv3x3_conv1x1: w1p / bias_p must come from pack_conv1x1_operand for this output width")
    if out is None:
        out = torch.empty((b, no, h, w), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (b, no, h, w):
        raise RuntimeError(f"conv3x3_conv1x1: output has shape {tuple(out.shape)}, expected {(b, no, h, w)}")
    po, so = _planes(out, "conv3x3_conv1x1 output")
    check(_lib.load().itermvs_conv3x3_conv1x1(ptr, sb, b, h, w, w0.tile.data_ptr(), _dev(w1p, "w1p").data_ptr(), 0, _ptr(bias_p), no,
                                              po, so, _stream()), "itermvs_conv3x3_conv1x1")
    return out


def pack_gru_conv_split3(w: Tensor) -> Tensor:
    """ConvGRU 3x3 weight [NO,43,3,3] (NO = 64: convz and convr stacked; 32: convq) -> the bf16x3 operands of itermvs_gru_conv:
    bfloat16 [NO/16][9][6][64][8].  Operands 0..2: terms h, m, l of input channel (j // 4) * 16 + 4 q + j % 4 (lane = 16 q + i,
    output channel 16 ob + i); operands 3..5: channel 32 + 8 (q % 2) + j (zero from 43 on), terms h, m and (l if q < 2 else h)."""
    no = w.shape[0]
    if tuple(w.shape[1:]) != (43, 3, 3) or no % 16:
        raise RuntimeError("pack_gru_conv_split3: expects a [16 k, 43, 3, 3] weight")
    nob = no // 16
    wf = w.float().reshape(nob, 16, 43, 9)
    ta = wf[:, :, :32].reshape(nob, 16, 2, 4, 4, 9).permute(0, 5, 3, 1, 2, 4).reshape(nob, 9, 64, 8)      # [ob, tap, (q, i), (jj, r)]
    ah, am, al = split_bf16x3(ta)
    full = torch.zeros((nob, 16, 16, 9), device=w.device, dtype=torch.float32)
    full[:, :, :11] = wf[:, :, 32:]
    tb = full.reshape(nob, 16, 2, 8, 9).permute(0, 4, 2, 1, 3)                                             # [ob, tap, half, i, j]
    tb = torch.stack([tb, tb], 2).reshape(nob, 9, 64, 8)                                                   # q = 2 second + half
    bh, bm, bl = split_bf16x3(tb)
    b3 = torch.cat([bl.reshape(nob, 9, 4, 16, 8)[:, :, :2], bh.reshape(nob, 9, 4, 16, 8)[:, :, 2:]], 2).reshape(nob, 9, 64, 8)
    return torch.stack([ah, am, al, bh, bm, b3], 2).contiguous()


def gru_conv(x: Tensor, wp: Tensor, bias: Optional[Tensor], h: Tensor, out: Tensor, out2: Optional[Tensor] = None,
             z: Optional[Tensor] = None) -> None:
    """itermvs_gru_conv.  ``z`` None: the update / reset gates (``wp`` [4,9,6,64,8]): x = [h | inputs] [B,43,H,W] ->
    out = z, out2 = r * h.  ``z`` given: the candidate state and the update (``wp`` [2,9,6,64,8]): x = [r*h | inputs] ->
    out (= out2 if given) = (1 - z) h + z tanh(convq(x))  (models/module.py:59-66)."""
    px, sx = _planes(x, "gru_conv input")
    b, c, hh, ww = x.shape
    mode = 0 if z is None else 1
    nob = 4 if mode == 0 else 2
    if c != 43 or not (wp.is_cuda and wp.dtype == torch.bfloat16 and tuple(wp.shape) == (nob, 9, 6, 64, 8) and wp.is_contiguous()):
        raise RuntimeError("gru_conv: expects a 43-channel input and the weights of pack_gru_conv_split3")
    if bias is not None and bias.numel() != 16 * nob:
        raise RuntimeError("gru_conv: bias size")
    if mode == 0 and out2 is None:
        raise RuntimeError("gru_conv: the gate form needs both outputs")
    for t in (h, out, out2, z):
        if t is not None and tuple(t.shape) != (b, 32, hh, ww):
            raise RuntimeError(f"gru_conv: expected [B,32,H,W] tensors, got {tuple(t.shape)}")
    ph, sh = _planes(h, "gru_conv h")
    po, so = _planes(out, "gru_conv out")
    po2, so2 = _planes(out2, "gru_conv out2") if out2 is not None else (None, 0)
    pz, sz = _planes(z, "gru_conv z") if z is not None else (None, 0)
    check(_lib.load().itermvs_gru_conv(px, sx, b, hh, ww, mode, wp.data_ptr(), _ptr(bias), ph, sh, pz, sz, po, so, po2, so2, _stream()),
          "itermvs_gru_conv")
    if CONV_FLOP_COUNTER["enabled"]:         # the dilated 3x3 layer(s) over the 43 input channels, one bracketed launch
        CONV_FLOP_COUNTER["flops"] += 2.0 * b * hh * ww * 16 * nob * 43 * 9
        CONV_FLOP_COUNTER["launches"] += 1


def head_regress(x: Tensor, w1p: Tensor, w2p: Tensor, bias2: Tensor,
                 nd_out: Optional[Sequence[Tuple[Tensor, int]]] = None, want_best: bool = False):
    """itermvs_head_regress: x [B,32,H,W] (after depth_head[0:2]) -> normalised depth (and arg-max bin),
    written like prob_regress's ``nd_out``.  Returns (nd | None, best | None)."""
    _dev(x, "x")
    b, c, h, w = x.shape
    if c != 32:
        raise RuntimeError("head_regress: expects the 32-channel depth-head feature")
    ptr, sb = _planes(x, "head input")
    p = h * w
    nd, dests = None, []
    if nd_out is None:
        nd = torch.empty((b, 1, h, w), device=x.device, dtype=torch.float32)
        dests.append((nd.data_ptr(), p))
    else:
        for buf, ch in nd_out:
            _dev(buf, "nd_out")
            assert buf.is_contiguous() and buf.shape[2:] == (h, w)
            dests.append((buf.data_ptr() + 4 * ch * p, buf.shape[1] * p))
    while len(dests) < 2:
        dests.append((None, 0))
    best = torch.empty((b, 1, h, w), device=x.device, dtype=torch.int64) if want_best else None
    check(_lib.load().itermvs_head_regress(ptr, sb, b, p, _dev(w1p, "w1").data_ptr(), _dev(w2p, "w2").data_ptr(),
                                           _dev(bias2, "bias2").data_ptr(), dests[0][0], dests[0][1], dests[1][0],
                                           dests[1][1], _ptr(best), _stream()), "itermvs_head_regress")
    return nd, best


def head_fused(hidden: Tensor, w0, w1p: Tensor, w2p: Tensor, bias2: Tensor,
               nd_out: Optional[Sequence[Tuple[Tensor, int]]] = None, want_best: bool = False, conf=None):
    """itermvs_head_fused: hidden [B,32,H,W] -> normalised depth (and arg-max bin); the dilated 3x3 layer (``w0``: its
    MfmaWeight), both 1x1 layers and the regression in one launch.  Returns (nd | None, best | None).
    ``conf`` = (MfmaWeight of confidence_head[0], its 1x1 layer as 33 floats {w, bias}, out [B,1,H,W]): itermvs_head_fused_conf,
    the confidence head (itermvs.py:147-151,198) rides in the same launch and writes sigmoid(...) to ``out``."""
    _dev(hidden, "hidden")
    b, c, h, w = hidden.shape
    if isinstance(w0, Tensor):               # pack_head_w0_split3: the dilated 3x3 layer in bf16x3 (needs the bf16x3 w2, no conf rider)
        if not (w0.is_cuda and w0.dtype == torch.bfloat16 and tuple(w0.shape) == (2, 9, 3, 64, 8) and w0.is_contiguous()) or c != 32:
            raise RuntimeError("head_fused: a tensor w0 must come from pack_head_w0_split3")
        if conf is not None or not (isinstance(w2p, Tensor) and w2p.dtype == torch.bfloat16):
            raise RuntimeError("head_fused: the bf16x3 3x3 layer needs the bf16x3 last layer (pack_head_w2_split3) and no confidence rider")
        w0_ptr, w0_format = w0.data_ptr(), 3
    else:
        if c != 32 or w0.tile is None or w0.cin != 32 or w0.cout != 32:
            raise RuntimeError("head_fused: expects the 32-channel hidden state and the 32 -> 32 3x3 weight")
        w0_ptr, w0_format = w0.tile.data_ptr(), 0
    ptr, sb = _planes(hidden, "hidden")
    p = h * w
    if not isinstance(w2p, torch.Tensor) or not w2p.is_cuda:
        raise RuntimeError("w2: expected a CUDA/ROCm tensor - the IterMVS HIP engine has no CPU path")
    if w2p.dtype == torch.bfloat16:          # pack_head_w2_split3: the 64 -> 256 layer in the bf16x3 form
        if tuple(w2p.shape) != (16, 2, 3, 64, 8) or not w2p.is_contiguous():
            raise RuntimeError("head_fused: a bfloat16 w2 must come from pack_head_w2_split3")
        w2_format = 3
    elif w2p.dtype == torch.float32 and w2p.numel() == 256 * 64 and w2p.is_contiguous():
        w2_format = 0
    else:
        raise RuntimeError("head_fused: w2 must come from pack_head_weights (float32) or pack_head_w2_split3 (bfloat16)")
    nd, dests = None, []
    if nd_out is None:
        nd = torch.empty((b, 1, h, w), device=hidden.device, dtype=torch.float32)
        dests.append((nd.data_ptr(), p))
    else:
        for buf, ch in nd_out:
            _dev(buf, "nd_out")
            assert buf.is_contiguous() and buf.shape[2:] == (h, w)
            dests.append((buf.data_ptr() + 4 * ch * p, buf.shape[1] * p))
    while len(dests) < 2:
        dests.append((None, 0))
    best = torch.empty((b, 1, h, w), device=hidden.device, dtype=torch.int64) if want_best else None
    if conf is not None:
        wc, cdot, cout = conf
        if wc.tile is None or wc.cin != 32 or wc.cout != 32 or cdot.numel() != 33 or tuple(cout.shape) != (b, 1, h, w) or not cout.is_contiguous():
            raise RuntimeError("head_fused: conf = (32 -> 32 3x3 MfmaWeight in the fp32 tile format, 33 floats, contiguous [B,1,H,W] output)")
        check(_lib.load().itermvs_head_fused_conf(ptr, sb, b, h, w, w0_ptr, _dev(w1p, "w1").data_ptr(),
                                                  w2p.data_ptr(), w2_format, _dev(bias2, "bias2").data_ptr(), dests[0][0], dests[0][1],
                                                  dests[1][0], dests[1][1], _ptr(best), wc.tile.data_ptr(), _dev(cdot, "conf_dot").data_ptr(),
                                                  _dev(cout, "conf").data_ptr(), p, _stream()), "itermvs_head_fused_conf")
        return nd, best
    check(_lib.load().itermvs_head_fused(ptr, sb, b, h, w, w0_ptr, w0_format, _dev(w1p, "w1").data_ptr(),
                                         w2p.data_ptr(), w2_format, _dev(bias2, "bias2").data_ptr(), dests[0][0], dests[0][1],
                                         dests[1][0], dests[1][1], _ptr(best), _stream()), "itermvs_head_fused")
    return nd, best


def prob_regress(logits: Tensor, nd_out: Optional[Sequence[Tuple[Tensor, int]]] = None, want_prob: bool = False,
                 want_best: bool = False):
    """itermvs.py:171-190 / 201-219.  logits [B,256,H,W] (NCHW or channels-last).
    ``nd_out``: up to two (buffer [B,Ct,H,W] contiguous, channel) destinations written in place;
    default: a fresh [B,1,H,W].  Returns (nd, prob | None, best | None)."""
    _dev(logits, "logits")
    b, k, h, w = logits.shape
    if k != 256:
        raise RuntimeError("prob_regress: expects 256 bins")
    sb, sc, sy, sx = logits.stride()
    if sy != w * sx:
        logits = logits.contiguous()
        sb, sc, sy, sx = logits.stride()
    p = h * w
    nd = None
    dests = []
    if nd_out is None:
        nd = torch.empty((b, 1, h, w), device=logits.device, dtype=torch.float32)
        dests.append((nd.data_ptr(), p))
    else:
        for buf, ch in nd_out:
            _dev(buf, "nd_out")
            assert buf.is_contiguous() and buf.shape[2:] == (h, w)
            dests.append((buf.data_ptr() + 4 * ch * p, buf.shape[1] * p))
    while len(dests) < 2:
        dests.append((None, 0))
    prob = torch.empty((b, 256, h, w), device=logits.device, dtype=torch.float32) if want_prob else None
    best = torch.empty((b, 1, h, w), device=logits.device, dtype=torch.int64) if want_best else None
    check(_lib.load().itermvs_prob_regress(logits.data_ptr(), sb, sc, sx, b, p, dests[0][0], dests[0][1], dests[1][0],
                                           dests[1][1], _ptr(prob), _ptr(best), _stream()), "itermvs_prob_regress")
    return nd, prob, best


def gru_rh(zr: Tensor, h_buf: Tensor, rh_buf: Tensor, hid: int = 32) -> None:
    """module.py:63-64: rh_buf[:, :hid] = sigmoid(zr[:, hid:]) * h_buf[:, :hid]  (buffers [B,Ct,H,W])."""
    b, _, h, w = zr.shape
    p = h * w
    assert zr.is_contiguous() and h_buf.is_contiguous() and rh_buf.is_contiguous()
    check(_lib.load().itermvs_gru_rh(_dev(zr, "zr").data_ptr(), _dev(h_buf, "h").data_ptr(), h_buf.shape[1] * p,
                                     _dev(rh_buf, "rh").data_ptr(), rh_buf.shape[1] * p, b, hid, p, _stream()),
          "itermvs_gru_rh")


def gru_out(zr: Tensor, q: Tensor, h_buf: Tensor, h_copy: Optional[Tensor] = None, hid: int = 32) -> None:
    """module.py:62,65: h_buf[:, :hid] = (1-z)*h + z*tanh(q), in place; ``h_copy`` [B,hid,H,W]
    optionally receives a contiguous copy of the new state."""
    b, _, h, w = zr.shape
    p = h * w
    assert zr.is_contiguous() and q.is_contiguous() and h_buf.is_contiguous()
    assert h_copy is None or (h_copy.is_contiguous() and h_copy.shape == (b, hid, h, w))
    check(_lib.load().itermvs_gru_out(_dev(zr, "zr").data_ptr(), _dev(q, "q").data_ptr(), _dev(h_buf, "h").data_ptr(),
                                      h_buf.shape[1] * p, _ptr(h_copy), b, hid, p, _stream()), "itermvs_gru_out")


def pack_scores(scores: Sequence[Tensor], dst0: Tensor, dst1: Optional[Tensor], ch0: int) -> None:
    """itermvs.py:124,193: write the three CorrNet outputs [B,N_l,H,W] into channels ch0.. of the buffers."""
    b, _, h, w = scores[0].shape
    n = (C.c_int32 * 3)(*[s.shape[1] for s in scores])
    sc = [_dev(s, "score").contiguous() for s in scores]
    assert dst0.is_contiguous() and (dst1 is None or (dst1.is_contiguous() and dst1.shape == dst0.shape))
    check(_lib.load().itermvs_pack_scores(sc[0].data_ptr(), sc[1].data_ptr(), sc[2].data_ptr(), n, b, h * w,
                                          dst0.data_ptr(), _ptr(dst1), dst0.shape[1] * h * w, ch0, _stream()),
          "itermvs_pack_scores")


def convex_upsample(logits: Tensor, nd: Tensor, inv_min: Tensor, inv_max: Tensor, nd_channel: int = 0,
                    want_norm: bool = False):
    """itermvs.py:262-264 + module.py:127-140 + :148-152.  logits [B,144,H,W] (raw, pre-softmax),
    nd = buffer [B,Ct,H,W] whose channel ``nd_channel`` holds the normalised depth.
    -> depth [B,1,4H,4W] (and the normalised up-sampled map if ``want_norm``)."""
    _dev(logits, "logits"); _dev(nd, "nd")
    b, k, h, w = logits.shape
    assert k == 144 and nd.is_contiguous()
    sb, sc, sy, sx = logits.stride()
    depth = torch.empty((b, 1, 4 * h, 4 * w), device=logits.device, dtype=torch.float32)
    norm = torch.empty_like(depth) if want_norm else None
    check(_lib.load().itermvs_convex_upsample(logits.data_ptr(), sb, sc, sy, sx, nd.data_ptr() + 4 * nd_channel * h * w,
                                              nd.shape[1] * h * w, _dev(inv_min, "inv_min").data_ptr(),
                                              _dev(inv_max, "inv_max").data_ptr(), b, h, w, depth.data_ptr(),
                                              _ptr(norm), _stream()), "itermvs_convex_upsample")
    return (depth, norm) if want_norm else depth


def final_upsample(logits: Tensor, nd: Tensor, inv_min: Tensor, inv_max: Tensor, conf: Tensor, nd_channel: int = 0):
    """convex_upsample (depth) and the x4 bilinear up-sampling of the confidence (itermvs.py:321-324) in ONE launch:
    logits [B,144,H,W], nd buffer [B,Ct,H,W], conf [B,1,H,W] -> (depth [B,1,4H,4W], confidence [B,1,4H,4W])."""
    _dev(logits, "logits"); _dev(nd, "nd")
    conf = _dev(conf, "conf").contiguous()
    b, k, h, w = logits.shape
    assert k == 144 and nd.is_contiguous() and tuple(conf.shape[-2:]) == (h, w)
    sb, sc, sy, sx = logits.stride()
    depth = torch.empty((b, 1, 4 * h, 4 * w), device=logits.device, dtype=torch.float32)
    m = conf.numel() // (h * w)
    conf_up = torch.empty(tuple(conf.shape[:-2]) + (4 * h, 4 * w), device=logits.device, dtype=torch.float32)
    check(_lib.load().itermvs_final_upsample(logits.data_ptr(), sb, sc, sy, sx, nd.data_ptr() + 4 * nd_channel * h * w,
                                             nd.shape[1] * h * w, _dev(inv_min, "inv_min").data_ptr(),
                                             _dev(inv_max, "inv_max").data_ptr(), b, h, w, depth.data_ptr(), conf.data_ptr(), m,
                                             conf_up.data_ptr(), _stream()), "itermvs_final_upsample")
    return depth, conf_up


def bilinear_up_into(x: Tensor, scale: int, out: Tensor, out2: Optional[Tensor] = None, act: str = "none") -> Tensor:
    """F.interpolate(x, scale_factor=scale, 'bilinear') (+tanh) written into ``out`` and ``out2`` -- [B,C,sH,sW] views with
    dense planes, e.g. channel slices of wider buffers (itermvs_bilinear_up2)."""
    x = _dev(x, "x").contiguous()
    b, c, h, w = x.shape
    for t in (out, out2):
        if t is not None and tuple(t.shape) != (b, c, h * scale, w * scale):
            raise RuntimeError(f"bilinear_up_into: destination has shape {tuple(t.shape)}")
    po, so = _planes(out, "out")
    p2, s2 = _planes(out2, "out2") if out2 is not None else (None, 0)
    check(_lib.load().itermvs_bilinear_up2(x.data_ptr(), b, c, h, w, scale, 1 if act == "tanh" else 0, po, so, p2, s2, _stream()),
          "itermvs_bilinear_up2")
    return out


def bilinear_up(x: Tensor, scale: int, act: str = "none") -> Tensor:
    """F.interpolate(x, scale_factor=scale, mode='bilinear') (+ tanh): x [B,C,H,W] -> [B,C,sH,sW]."""
    x = _dev(x, "x").contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, c, scale * h, scale * w), device=x.device, dtype=torch.float32)
    check(_lib.load().itermvs_bilinear_up(x.data_ptr(), b * c, h, w, scale, {"none": 0, "tanh": 1}[act], out.data_ptr(),
                                          _stream()), "itermvs_bilinear_up")
    return out


# ------------------------------------------------------------------------------------------
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "gru_rh": 4, "gru_out": 5, "relu_dot": 6, "relu_dot_sigmoid": 7}


def pack_conv_weight(w: Tensor, transposed: bool = False) -> Tensor:
    """VALU-kernel format: [Cout,Cin,k,k] (Conv2d) or [Cin,Cout,k,k] (ConvTranspose2d) -> [Cin,k,k,Cout]."""
    return (w.permute(0, 2, 3, 1) if transposed else w.permute(1, 2, 3, 0)).contiguous()


def tile_k_split(cin: int) -> int:
    """S of the LDS-tiled kernel (conv_tile.hip): consecutive channels held by one k-slot of a chunk."""
    return 1 if cin <= 4 else 2 if cin <= 8 else 4


def split_bf16x3(t: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """fp32 tensor -> (h, m, l) bfloat16 tensors with h + m + l == t EXACTLY (finite t): h = t truncated to bf16's 8
    significant bits, m = (t - h) truncated, l = t - h - m (at most 8 bits left).  The split conv_tile3.hip applies to the
    activations when it stages them."""
    t = t.float().contiguous()
    mask = -65536
    h = (t.view(torch.int32) & mask).view(torch.float32)
    r = t - h
    m = (r.view(torch.int32) & mask).view(torch.float32)
    l = r - m
    return h.to(torch.bfloat16), m.to(torch.bfloat16), l.to(torch.bfloat16)


# Default arithmetic of the 3x3 convolutions with more than 8 input channels (MfmaWeight(split3=None)):
#   True  -> bf16x3 split on v_mfma_f32_16x16x32_bf16 (conv_tile3.hip, weight_format 3): fp32-comparable error, 2.7x less matrix time
#   False -> exact fp32 MFMA (conv_tile.hip, weight_format 2): bit-for-bit an fmaf chain
MFMA_SPLIT3_DEFAULT = True
SPLIT3_STRIDE2 = False      # True: stride-2 3x3 layers (Cin > 8) of a split3 MfmaWeight on the bf16 instruction as well (measured slower)


class MfmaWeight:
    """Matrix-core formats of a Conv2d weight [Cout,Cin,k,k], zero padded:
    ``data`` (weight_format 1, conv_mfma.hip): [k*k, Cin_pad4, Cout_pad16];
    ``tile`` (weight_format 2, conv_tile.hip, 3x3 only): [9, chunks, 4, Cout_pad16, S] whose element
    (tap, ch, q, co, s) is the weight of input channel ch*4*S + q*S + s;
    ``tile3`` (weight_format 3, conv_tile3.hip, 3x3 with Cin > 8, ``split3``): bfloat16 [9, chunks, 3, Cout_pad16, 16] whose
    element (tap, ch, p, co, c) is term p (h, m, l of ``split_bf16x3``) of the weight of input channel ch*16 + c."""

    def __init__(self, w: Tensor, transposed: bool = False, split3: Optional[bool] = None):
        """``transposed``: ``w`` is a ConvTranspose2d weight [Cin,Cout,k,k] (only the tile format is built).
        ``split3``: build (and let conv2d use) the bf16x3 form; None = ``MFMA_SPLIT3_DEFAULT``."""
        self.transposed = transposed
        if transposed:
            w = w.permute(1, 0, 2, 3)
        cout, cin, k, _ = w.shape
        cin_p, cout_p = (cin + 3) // 4 * 4, (cout + 15) // 16 * 16
        packed = torch.zeros((k * k, cin_p, cout_p), device=w.device, dtype=torch.float32)
        packed[:, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(k * k, cin, cout)
        self.data, self.cin, self.cout, self.ksize = packed.contiguous(), cin, cout, k
        self.tile = None
        if k == 3:
            s = tile_k_split(cin)
            nch = (cin + 4 * s - 1) // (4 * s)
            t = torch.zeros((9, nch * 4 * s, cout_p), device=w.device, dtype=torch.float32)
            t[:, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(9, cin, cout)
            self.tile = t.reshape(9, nch, 4, s, cout_p).permute(0, 1, 2, 4, 3).contiguous()
        self.tile3 = None
        # (up to 64 input channels: the split weights of a 16-channel block -- 13.8 KB per input chunk -- share the LDS with the tile)
        if k == 3 and 8 < cin <= 64 and not transposed and (MFMA_SPLIT3_DEFAULT if split3 is None else split3):
            nch = (cin + 15) // 16
            t = torch.zeros((9, nch * 16, cout_p), device=w.device, dtype=torch.float32)
            t[:, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(9, cin, cout)
            t = t.reshape(9, nch, 16, cout_p).permute(0, 1, 3, 2)                      # [9, chunks, Cout_pad, 16]
            self.tile3 = torch.stack(split_bf16x3(t), 2).contiguous()                  # [9, chunks, 3, Cout_pad, 16] bf16
        # 5 .. 8 input channels: the TAP-PAIR form of conv_tile3.hip (stride 1, no dilation): K-step v = 0..4 carries the 8 channels
        # at tap 2v in slots 0..7 and at tap 2v + 1 in slots 8..15 (the tenth tap: zeros) -> [5, 1, 3, Cout_pad, 16] bf16
        elif k == 3 and 4 < cin <= 8 and not transposed and (MFMA_SPLIT3_DEFAULT if split3 is None else split3):
            t = torch.zeros((10, 8, cout_p), device=w.device, dtype=torch.float32)
            t[:9, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(9, cin, cout)
            t = t.reshape(5, 2, 8, cout_p).permute(0, 3, 1, 2).reshape(5, 1, cout_p, 16)      # [pair, chunk, Cout_pad, (tap in pair, ch)]
            self.tile3 = torch.stack(split_bf16x3(t), 2).contiguous()                  # [5, 1, 3, Cout_pad, 16] bf16


_TILE_SHAPES = {(1, 1), (2, 1), (1, 2)}   # (stride, dilation) instantiated in conv_tile.hip


def _use_tile(wt: "MfmaWeight", stride: int, dilation: int) -> bool:
    if wt.tile is None:
        return False
    return (stride, dilation) in _TILE_SHAPES and (wt.cin > 4 or (stride, dilation) == (1, 1))


def _planes(t: Tensor, name: str):
    """[N,C,H,W] view whose channel planes are dense (a channel slice of a contiguous buffer is fine)."""
    _dev(t, name)
    n, c, h, w = t.shape
    if t.stride(3) != 1 or t.stride(2) != w or (c > 1 and t.stride(1) != h * w):
        raise RuntimeError(f"{name}: needs dense [C,H,W] planes (NCHW); got strides {t.stride()}")
    return t.data_ptr(), (t.stride(0) if n > 1 else c * h * w)


def conv2d(x: Tensor, weight, bias=None, *, ksize: int = 3, stride: int = 1, pad: int = 1, dilation: int = 1,
           act: str = "none", add: Optional[Tensor] = None, aux1: Optional[Tensor] = None,
           aux2: Optional[Tensor] = None, out: Optional[Tensor] = None, out2: Optional[Tensor] = None,
           transposed: bool = False, seg_end: Sequence[int] = (), add_up2: bool = False,
           channels_last_out: bool = False, split=None) -> Tensor:
    """itermvs_conv2d.  ``weight`` / ``bias``: packed tensor(s) (see pack_conv_weight); pass lists of up to
    three for per-segment weight sets with ``seg_end`` = batch boundaries.  ``add_up2``: ``add`` is the
    half-resolution tensor whose x2 bilinear up-sampling is added (fused F.interpolate).
    ``channels_last_out``: ``out`` is written in channels-last memory format ([N,H,W,C] dense, what the
    correlation kernels read); ``out2`` can still take the planar copy.
    ``split`` = (channel, act_b, out_b): output channels from ``channel`` on are a second result with its own
    activation and destination (two convolutions of one input in one launch).  Returns ``out``."""
    weights = list(weight) if isinstance(weight, (list, tuple)) else [weight]
    biases = list(bias) if isinstance(bias, (list, tuple)) else [bias] * len(weights)
    n, cin, hin, win = x.shape
    mfma, tiled, split3 = isinstance(weights[0], MfmaWeight), False, False
    if mfma:
        if transposed != weights[0].transposed:
            raise RuntimeError("conv2d: weight was packed for the other direction (MfmaWeight(transposed=...))")
        if any(wt.cin != cin or wt.ksize != ksize for wt in weights):
            raise RuntimeError("conv2d: MfmaWeight does not match the input channels / kernel size")
        cout = weights[0].cout
        tiled = transposed or all(_use_tile(wt, stride, dilation) for wt in weights)
        # (stride-2 layers keep the fp32 form: in the step the split form measured 30.2 us per launch against 28.0 -- their halo tiles
        #  leave no LDS for two resident workgroups and four channel blocks re-stage and re-split each tile; SPLIT3_STRIDE2 = True
        #  selects the split form for them, profiles/r06/r06l_*)
        split3 = (tiled and not transposed and (stride == 1 or (stride == 2 and cin > 8 and dilation == 1 and SPLIT3_STRIDE2))
                  and all(wt.tile3 is not None for wt in weights)
                  and (cin > 8 or dilation == 1))                    # (the tap-pair form of 5..8 channels: no dilation)
        weights = [(wt.tile3 if split3 else wt.tile) if tiled else wt.data for wt in weights]
    else:
        cout = weights[0].shape[3]
    if transposed:
        hout, wout = 2 * hin, 2 * win
    else:
        span = (ksize - 1) * dilation + 1
        hout, wout = (hin + 2 * pad - span) // stride + 1, (win + 2 * pad - span) // stride + 1
    cout_total = cout
    dot = None
    if act in ("relu_dot", "relu_dot_sigmoid"):
        # ReLU + a 1x1 convolution to ONE channel (+ sigmoid) in the epilogue: aux1 = Cout + 1 floats {w[Cout], bias}
        if not (mfma and tiled and cout in (16, 32) and aux1 is not None and aux1.numel() == cout + 1 and add is None and out2 is None
                and split is None):
            raise RuntimeError("conv2d: act='relu_dot' needs a 16- or 32-channel 3x3 matrix-core layer and aux1 = Cout + 1 floats")
        dot, aux1, cout = _dev(aux1, "aux1").contiguous(), None, 1
    if split is not None:
        if not (mfma and ksize == 3 and not transposed):
            raise RuntimeError("conv2d: split results need the tiled matrix-core kernel (3x3 MfmaWeight)")
        cout = int(split[0])
    if out is None:
        out = torch.empty((n, cout, hout, wout), device=x.device, dtype=torch.float32,
                          memory_format=torch.channels_last if channels_last_out else torch.contiguous_format)
    if not channels_last_out and out.dtype != torch.float32:
        raise RuntimeError("conv2d: 16-bit storage exists for channels-last outputs only")
    p = ConvParams()
    if (cin == 8 and split3 and not transposed and x.dim() == 4 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last)):
        # channels-last input [N,H,W,8] (corr_init's groups-last volume): two 16-byte loads per pixel in the tap-pair kernel
        _dev(x, "conv input")
        p.inp, p.in_sn, p.in_layout = x.data_ptr(), 8 * hin * win, 1
    else:
        p.inp, p.in_sn = _planes(x, "conv input")
    if channels_last_out:
        if not out.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("conv2d: channels_last_out needs a dense channels-last `out`")
        # out_layout 1 / 2 / 3 = fp32 / fp16 / bf16 storage of the feature map (rounded to nearest even from fp32)
        p.out, p.out_sn, p.out_layout = out.data_ptr(), cout * hout * wout, 1 + _feat(out, "conv output")
    else:
        p.out, p.out_sn = _planes(out, "conv output")
    if out.shape != (n, cout, hout, wout):
        raise RuntimeError(f"conv2d: out has shape {tuple(out.shape)}, expected {(n, cout, hout, wout)}")
    if out2 is not None:
        assert out2.is_contiguous() and out2.shape == out.shape
        p.out2 = out2.data_ptr()
    for name, t in (("add", add), ("aux1", aux1), ("aux2", aux2)):
        if t is not None:
            ptr, sn = _planes(t, name)
            want = (n, cout, hout // 2, wout // 2) if (name == "add" and add_up2) else tuple(out.shape)
            if tuple(t.shape) != want:
                raise RuntimeError(f"conv2d: {name} has shape {tuple(t.shape)}, expected {want}")
            setattr(p, name, ptr)
            setattr(p, name + "_sn", sn)
    p.n_seg = len(weights)
    for i, (wt, bs) in enumerate(zip(weights, biases)):
        if not (mfma and split3):
            _dev(wt, "weight")
        elif not wt.is_cuda:
            raise RuntimeError("weight: expected a CUDA/ROCm tensor - the IterMVS HIP engine has no CPU path")
        if not mfma and (not wt.is_contiguous() or wt.shape != (cin, ksize, ksize, cout)):
            raise RuntimeError(f"conv2d: packed weight must be contiguous [{cin},{ksize},{ksize},{cout}], got {tuple(wt.shape)}")
        p.weight[i] = wt.data_ptr()
        p.bias[i] = None if bs is None else _dev(bs, "bias").data_ptr()
    for i, e in enumerate(seg_end):
        p.seg_end[i] = e
    p.N, p.Cin, p.Hin, p.Win, p.Cout = n, cin, hin, win, cout_total
    if dot is not None:
        p.aux1, p.aux1_sn = dot.data_ptr(), 0
    if split is not None:
        ob = split[2]
        if tuple(ob.shape) != (n, cout_total - cout, hout, wout):
            raise RuntimeError(f"conv2d: split output has shape {tuple(ob.shape)}, expected {(n, cout_total - cout, hout, wout)}")
        p.split_cout, p.act_b = cout, ACT[split[1]]
        p.out_b, p.out_b_sn = _planes(ob, "split output")
    p.ksize, p.stride, p.pad, p.dilation = ksize, stride, pad, dilation
    p.transposed, p.act = int(transposed), ACT[act]
    p.weight_format = ((3 if split3 else 2) if tiled else 1) if mfma else 0
    p.add_mode = int(add_up2)
    if CONV_FLOP_COUNTER["enabled"]:
        CONV_FLOP_COUNTER["flops"] += 2.0 * n * hout * wout * cout_total * cin * ksize * ksize / (4.0 if transposed else 1.0)
        CONV_FLOP_COUNTER["launches"] += 1
    check(_lib.load().itermvs_conv2d(C.byref(p), _stream()), "itermvs_conv2d")
    return out


# ------------------------------------------------------------------------------------------
CONV_FLOP_COUNTER = {"enabled": False, "flops": 0.0, "launches": 0}


def fuse_depth(depth_ref: Tensor, conf_ref: Tensor, depth_src: Sequence[Tensor], mats: Tensor, geo_pixel_thres: float = 1.0,
               geo_depth_thres: float = 0.01, photo_thres: float = 0.3, geo_mask_thres: int = 3):
    """itermvs_fuse_depth (eval.py:154-269 for one reference view).  depth_ref / conf_ref [H,W], depth_src: S maps [H,W],
    mats [S,60] (itermvs_amd.fusion.pair_matrices).  -> (depth_avg float64 [H,W], photo, geo, final uint8, count int32)."""
    h, w = depth_ref.shape
    s = len(depth_src)
    depth_ref, conf_ref = _dev(depth_ref, "depth_ref").contiguous(), _dev(conf_ref, "conf_ref").contiguous()
    srcs = [_dev(t, "depth_src").contiguous() for t in depth_src]
    if any(tuple(t.shape) != (h, w) for t in srcs + [conf_ref]) or tuple(mats.shape) != (s, 60):
        raise RuntimeError("fuse_depth: all maps must be [H,W] and mats [S,60]")
    mats = _dev(mats, "mats").float().contiguous()
    ptrs = (C.c_void_p * s)(*[t.data_ptr() for t in srcs])
    dev = depth_ref.device
    avg = torch.empty((h, w), device=dev, dtype=torch.float64)
    photo, geo, final = (torch.empty((h, w), device=dev, dtype=torch.uint8) for _ in range(3))
    cnt = torch.empty((h, w), device=dev, dtype=torch.int32)
    check(_lib.load().itermvs_fuse_depth(depth_ref.data_ptr(), conf_ref.data_ptr(), ptrs, mats.data_ptr(), s, h, w,
                                         float(geo_pixel_thres), float(geo_depth_thres), float(photo_thres), int(geo_mask_thres),
                                         avg.data_ptr(), photo.data_ptr(), geo.data_ptr(), final.data_ptr(), cnt.data_ptr(),
                                         _stream()), "itermvs_fuse_depth")
    return avg, photo, geo, final, cnt


CORRNET_WEIGHT_FLOATS = 14288
CORRNET_WEIGHT_FLOATS_SPLIT3 = 23120     # bf16x3 form: conv0 + conv2 + the two transposed convolutions as split bf16 operands


def _mfma_operand_order(w_tap_ci_co: Tensor, co_pad: int) -> Tensor:
    """[9, Cin, Cout] -> [9][Cin/4][4][co_pad] (zero padded): the A-operand order of v_mfma_f32_16x16x4_f32 in itermvs_corrnet"""
    taps, cin, cout = w_tap_ci_co.shape
    out = torch.zeros((taps, cin // 4, 4, co_pad), device=w_tap_ci_co.device, dtype=torch.float32)
    out[..., :cout] = w_tap_ci_co.reshape(taps, cin // 4, 4, cout)
    return out.reshape(-1)


def pack_corrnet_weights(w: Dict[str, Tensor], prefix: str, split3: bool = False) -> Tensor:
    """the six layers of one CorrNet (state-dict names ``prefix`` + conv0.conv.weight ... conv5.bias) in the layout of
    itermvs_corrnet (include/itermvs_hip.h): the five matrix-core layers in operand order [tap][k-step][q][co], conv5 as
    [ci][tap], the bias, padding.  ``split3``: the set of itermvs_corrnet_bf16x3 -- conv0 as the bf16 A operands of its 18
    v_mfma_f32_16x16x32_bf16 ([mfma][q][row m][8 channels], three-term split of the fp32 weights), conv1 unchanged (fp32), conv2 and
    the two transposed convolutions in itermvs_conv2d's weight_format 3 (bf16 [tap][chunk][h, m, l][row co][16 ci]; a transposed
    convolution as the convolution weight [co][ci][ky][kx] = w[ci][co][ky][kx]), conv5 and the bias unchanged."""
    conv = lambda name, pad: _mfma_operand_order(w[prefix + name].float().permute(2, 3, 1, 0).reshape(9, w[prefix + name].shape[1], -1), pad)
    # ConvTranspose2d weights are [ci, co, ky, kx]
    dconv = lambda name, pad: _mfma_operand_order(w[prefix + name].float().permute(2, 3, 0, 1).reshape(9, w[prefix + name].shape[0], -1), pad)
    # conv0 computes two output rows per matrix-core tile: rows 0..7 = the 8 channels with the taps in window rows 0..2,
    # rows 8..15 = the same channels with the taps in window rows 1..3 ([window row 4][kx 3][ci 8][16])
    w0 = w[prefix + "conv0.conv.weight"].float()                                   # [co 8, ci 8, ky 3, kx 3]
    two = torch.zeros((4, 3, 8, 16), device=w0.device, dtype=torch.float32)
    two[0:3, :, :, 0:8] = w0.permute(2, 3, 1, 0)
    two[1:4, :, :, 8:16] = w0.permute(2, 3, 1, 0)
    if split3:
        wh, wm, wl = split_bf16x3(two)                                              # each [4, 3, 8 ci, 16 rows] bf16
        wp = lambda t: t.reshape(12, 8, 16).permute(0, 2, 1)                        # [window position 12, row m 16, ci 8]
        h, m_, l = wp(wh), wp(wm), wp(wl)
        ops_a = []
        for v in range(6):                  # MFMA 2v = [h h | h h], 2v + 1 = [m m | m m]: lanes q = 0..3 -> window position 2v + (q & 1)
            for term in (h, m_):
                ops_a.append(torch.stack([term[2 * v], term[2 * v + 1], term[2 * v], term[2 * v + 1]]))       # [q 4, m 16, ci 8]
        for v in range(6):                  # MFMA 12 + v = [l l | h h]
            ops_a.append(torch.stack([l[2 * v], l[2 * v + 1], h[2 * v], h[2 * v + 1]]))
        first = torch.stack(ops_a).contiguous().view(torch.int16).reshape(-1).view(torch.float32)              # 18 x 64 x 8 bf16 = 4608 floats
    else:
        first = _mfma_operand_order(two.reshape(12, 8, 16), 16)
    if split3:
        as_floats = lambda wt: MfmaWeight(wt, split3=True).tile3.contiguous().view(torch.int16).reshape(-1).view(torch.float32)
        mid = [as_floats(w[prefix + "conv2.conv.weight"].float()),                       # [9][1][3][32][16] bf16 = 6912 floats
               as_floats(w[prefix + "conv3.weight"].float().permute(1, 0, 2, 3)),          # [9][2][3][16][16]
               as_floats(w[prefix + "conv4.weight"].float().permute(1, 0, 2, 3))]          # [9][1][3][16 (8 used)][16] = 3456 floats
    else:
        mid = [conv("conv2.conv.weight", 32), dconv("conv3.weight", 16), dconv("conv4.weight", 16)]
    parts = [first, conv("conv1.conv.weight", 16)] + mid + [
             w[prefix + "conv5.weight"].float().permute(1, 2, 3, 0).reshape(-1), w[prefix + "conv5.bias"].float().reshape(-1)]
    flat = torch.cat(parts + [torch.zeros(7, device=parts[0].device)])
    assert flat.numel() == (CORRNET_WEIGHT_FLOATS_SPLIT3 if split3 else CORRNET_WEIGHT_FLOATS)
    return flat.contiguous()


STEM_W0_FLOATS, STEM_W1_FLOATS = 224, 2336


def pack_stem_weights(w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor, wd: Tensor, bd: Tensor) -> Tuple[Tensor, Tensor]:
    """weights of itermvs_stem (BatchNorm already folded): FeatureNet.conv1 (w0 [8,3,3,3], b0) as [ci][ky][kx][co] + biases;
    layer1[0].conv1 (w1 [16,8,3,3], b1) and layer1[0].downsample (wd, bd), output channels concatenated, in matrix-core
    operand order [tap][k-step][q][co 32] + the 32 biases"""
    if tuple(w0.shape) != (8, 3, 3, 3) or tuple(w1.shape) != (16, 8, 3, 3) or tuple(wd.shape) != (16, 8, 3, 3):
        raise RuntimeError("pack_stem_weights: expects conv1 [8,3,3,3] and two stride-2 layers [16,8,3,3]")
    p0 = torch.cat([w0.float().permute(1, 2, 3, 0).reshape(-1), b0.float().reshape(-1)]).contiguous()
    wc = torch.cat([w1, wd]).float()                                              # [32, 8, 3, 3]
    p1 = torch.cat([_mfma_operand_order(wc.permute(2, 3, 1, 0).reshape(9, 8, 32), 32), b1.float().reshape(-1),
                    bd.float().reshape(-1)]).contiguous()
    assert p0.numel() == STEM_W0_FLOATS and p1.numel() == STEM_W1_FLOATS
    return p0, p1


def stem(x: Tensor, w0: Tensor, w1: Tensor, compose=None, quads: bool = False):
    """itermvs_stem: x [M,3,H,W] -> (relu(layer1[0].conv1(f0)), layer1[0].downsample(f0)) with f0 = FeatureNet.conv1(x)
    kept in LDS; both [M,16,H2,W2].  ``w0`` / ``w1`` from pack_stem_weights.
    ``compose`` = (mats [n,V,4,4], nan_flag | None, (depth_min, depth_max)): itermvs_stem_compose -- the camera composition
    (compose_proj + inverse depth range) rides in the same launch; returns (y, sc, proj [n,V-1,12], inv_min, inv_max).
    ``quads``: y and sc are written as channel quads -- tensors of shape [M,4,H2,W2,4] holding channel 4*cq + c at [m,cq,y,x,c] --
    the layout res_chain16(..., quads=True) fetches with 16-byte loads."""
    ptr, x_sn = _planes(x, "stem input")
    m, c, h, w = x.shape
    if c != 3:
        raise RuntimeError("stem: expects [M,3,H,W]")
    if w0.numel() != STEM_W0_FLOATS or w1.numel() != STEM_W1_FLOATS:
        raise RuntimeError("stem: weights must come from pack_stem_weights")
    h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((m, 4, h2, w2, 4) if quads else (m, 16, h2, w2), device=x.device, dtype=torch.float32)
    sc = torch.empty_like(y)
    if compose is not None:
        mats, nan_flag, depth_range = compose
        mats = _dev(mats, "mats").contiguous()
        n, v = mats.shape[0], mats.shape[1]
        proj = torch.empty((n, v - 1, 12), device=mats.device, dtype=torch.float32)
        dmin, dmax = (_dev(t, "depth range").contiguous() for t in depth_range)
        imin, imax = torch.empty_like(dmin), torch.empty_like(dmax)
        check(_lib.load().itermvs_stem_compose(ptr, x_sn, m, h, w, _dev(w0, "stem weights").data_ptr(), _dev(w1, "stem weights").data_ptr(),
                                               y.data_ptr(), sc.data_ptr(), 16 * h2 * w2, int(quads), mats.data_ptr(), n, v, proj.data_ptr(),
                                               _ptr(nan_flag), dmin.data_ptr(), dmax.data_ptr(), dmin.numel(), imin.data_ptr(),
                                               imax.data_ptr(), _stream()), "itermvs_stem_compose")
        return y, sc, proj, imin, imax
    check(_lib.load().itermvs_stem(ptr, x_sn, m, h, w, _dev(w0, "stem weights").data_ptr(), _dev(w1, "stem weights").data_ptr(),
                             y.data_ptr(), sc.data_ptr(), 16 * h2 * w2, int(quads), _stream()), "itermvs_stem")
    return y, sc


def res_chain16(y1: Tensor, shortcut: Tensor, weights: Sequence["MfmaWeight"], biases: Sequence[Optional[Tensor]],
                out: Optional[Tensor] = None, quads: bool = False) -> Tensor:
    """itermvs_res_chain16: FeatureNet.layer1 behind the stem in one launch -- a = relu(conv(y1; W0) + b0 + shortcut),
    b = relu(conv(a; W1) + b1), out = relu(conv(b; W2) + b2 + a); ``y1`` / ``shortcut`` [N,16,H,W] (ops.stem's results) or,
    with ``quads``, [N,4,H,W,4] (ops.stem(..., quads=True)); ``weights``: the three layers' MfmaWeight (bf16x3 form, 16 -> 16),
    ``biases``: [16] each or None.  ``out`` [N,16,H,W] planes."""
    if quads:
        for t, name in ((y1, "res_chain16 input"), (shortcut, "res_chain16 shortcut")):
            _dev(t, name)
            if t.dim() != 5 or t.shape[1] != 4 or t.shape[4] != 4 or not t.is_contiguous():
                raise RuntimeError(f"{name}: quads layout is a contiguous [N,4,H,W,4] tensor")
        n, _, h, w, _ = y1.shape
        if tuple(shortcut.shape) != tuple(y1.shape):
            raise RuntimeError("res_chain16: input and shortcut differ in shape")
        ptr, y_sn, sp, s_sn = y1.data_ptr(), 16 * h * w, shortcut.data_ptr(), 16 * h * w
    else:
        ptr, y_sn = _planes(y1, "res_chain16 input")
        sp, s_sn = _planes(shortcut, "res_chain16 shortcut")
        n, c, h, w = y1.shape
        if c != 16 or tuple(shortcut.shape) != (n, c, h, w):
            raise RuntimeError("res_chain16: expects two [N,16,H,W] tensors")
    if len(weights) != 3 or len(biases) != 3:
        raise RuntimeError("res_chain16: three layers")
    for wt in weights:
        if not isinstance(wt, MfmaWeight) or wt.tile3 is None or wt.cin != 16 or wt.cout != 16 or wt.ksize != 3:
            raise RuntimeError("res_chain16: weights must be 3x3 16 -> 16 MfmaWeight built with split3=True")
        if not wt.tile3.is_cuda or wt.tile3.dtype != torch.bfloat16:
            raise RuntimeError("res_chain16 weights: expected bfloat16 ROCm tensors (MfmaWeight.tile3)")
    if out is None:
        out = torch.empty((n, 16, h, w), device=y1.device, dtype=torch.float32)
    elif tuple(out.shape) != (n, 16, h, w):
        raise RuntimeError(f"res_chain16: output has shape {tuple(out.shape)}, expected {(n, 16, h, w)}")
    po, o_sn = _planes(out, "res_chain16 output")
    wp = (C.c_void_p * 3)(*[wt.tile3.data_ptr() for wt in weights])
    bl = [None if b is None else _dev(b, "res_chain16 bias").float().contiguous() for b in biases]
    bp = (C.c_void_p * 3)(*[_ptr(b) for b in bl])
    check(_lib.load().itermvs_res_chain16(ptr, y_sn, sp, s_sn, int(quads), n, h, w, wp, bp, po, o_sn, _stream()), "itermvs_res_chain16")
    if CONV_FLOP_COUNTER["enabled"]:         # the three layers' FLOPs (the recomputed halo is not counted), one bracketed launch
        CONV_FLOP_COUNTER["flops"] += 3 * 2.0 * n * h * w * 16 * 16 * 9
        CONV_FLOP_COUNTER["launches"] += 1
    return out


def lateral_conv3x3(fine: Tensor, coarse: Tensor, w_lat: "MfmaWeight", b_lat: Optional[Tensor], w_out: "MfmaWeight",
                    b_out: Optional[Tensor], out: Optional[Tensor] = None, channels_last_out: bool = False,
                    out2: Optional[Tensor] = None) -> Tensor:
    """itermvs_lateral_conv3x3: out = conv3x3(F.interpolate(coarse, x2, bilinear) + conv1x1(fine; w_lat) + b_lat; w_out) + b_out
    in one launch (net.py:48-50), the 48-channel intermediate map stays on the chip.  ``fine`` [N,16,H,W], ``coarse``
    [N,48,H/2,W/2] planes; ``w_lat``: MfmaWeight of the 1x1 layer 16 -> 48, ``w_out``: MfmaWeight(split3=True) of the 3x3 layer
    48 -> 16.  ``channels_last_out``: ``out`` in channels-last memory format, fp32 / fp16 / bf16 storage (the feature maps the
    correlation kernels gather from); ``out2``: optional fp32 planar copy."""
    fp, f_sn = _planes(fine, "lateral_conv3x3 fine input")
    cp, c_sn = _planes(coarse, "lateral_conv3x3 coarse input")
    n, cf, h, w = fine.shape
    if h % 2 or w % 2 or tuple(coarse.shape) != (n, 48, h // 2, w // 2):
        raise RuntimeError(f"lateral_conv3x3: coarse input {tuple(coarse.shape)} is not [N,48,H/2,W/2] of fine input {tuple(fine.shape)}")
    if not (isinstance(w_lat, MfmaWeight) and w_lat.ksize == 1 and w_lat.cin == cf and w_lat.cout == 48):
        raise RuntimeError("lateral_conv3x3: w_lat must be the MfmaWeight of a 1x1 layer Cf -> 48")
    if not (isinstance(w_out, MfmaWeight) and w_out.ksize == 3 and w_out.cin == 48 and w_out.tile3 is not None):
        raise RuntimeError("lateral_conv3x3: w_out must be the MfmaWeight (split3=True) of a 3x3 layer 48 -> Cout")
    if not (w_out.tile3.is_cuda and w_lat.data.is_cuda):
        raise RuntimeError("lateral_conv3x3 weights: expected ROCm tensors")
    cout = w_out.cout
    if out is None:
        out = torch.empty((n, cout, h, w), device=fine.device, dtype=torch.float32,
                          memory_format=torch.channels_last if channels_last_out else torch.contiguous_format)
    if tuple(out.shape) != (n, cout, h, w):
        raise RuntimeError(f"lateral_conv3x3: output has shape {tuple(out.shape)}, expected {(n, cout, h, w)}")
    if channels_last_out:
        if not out.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("lateral_conv3x3: channels_last_out needs a dense channels-last `out`")
        po, o_sn, layout = out.data_ptr(), cout * h * w, 1 + _feat(out, "lateral_conv3x3 output")
    else:
        if out.dtype != torch.float32:
            raise RuntimeError("lateral_conv3x3: planar output is fp32")
        po, o_sn = _planes(out, "lateral_conv3x3 output")
        layout = 0
    if out2 is not None and not (out2.dtype == torch.float32 and out2.is_contiguous() and tuple(out2.shape) == (n, cout, h, w)):
        raise RuntimeError("lateral_conv3x3: out2 must be a dense fp32 [N,Cout,H,W] tensor")
    bl = None if b_lat is None else _dev(b_lat, "lateral_conv3x3 bias").float().contiguous()
    bo = None if b_out is None else _dev(b_out, "lateral_conv3x3 bias").float().contiguous()
    check(_lib.load().itermvs_lateral_conv3x3(fp, f_sn, cf, cp, c_sn, n, h, w, w_lat.data.data_ptr(), _ptr(bl), w_out.tile3.data_ptr(),
                                              _ptr(bo), cout, po, o_sn, layout, _ptr(out2), _stream()), "itermvs_lateral_conv3x3")
    if CONV_FLOP_COUNTER["enabled"]:         # 1x1 layer + 3x3 layer (the recomputed halo is not counted), one bracketed launch
        CONV_FLOP_COUNTER["flops"] += 2.0 * n * h * w * (cf * 48 + 48 * cout * 9)
        CONV_FLOP_COUNTER["launches"] += 1
    return out


def corrnet(x: Tensor, weight_sets: Sequence[Tensor], seg_end: Sequence[int] = (), out: Optional[Tensor] = None,
            out2: Optional[Tensor] = None) -> Tensor:
    """itermvs_corrnet: x [M,8,H,W] -> [M,1,H,W]; ``weight_sets`` = 1..3 tensors from pack_corrnet_weights, ``seg_end`` the
    batch boundaries between them; ``out`` / ``out2``: [M,1,H,W] views with dense planes (e.g. channels of the GRU buffers)"""
    ptr, x_sn = _planes(x, "corrnet input")
    m, c, h, w = x.shape
    if c != 8 or h % 4 or w % 4:
        raise RuntimeError("corrnet: expects [M,8,H,W] with H and W multiples of 4")
    if out is None:
        out = torch.empty((m, 1, h, w), device=x.device, dtype=torch.float32)
    po, o_sn = _planes(out, "corrnet output")
    p2, o2_sn = _planes(out2, "corrnet output 2") if out2 is not None else (None, 0)
    for t in (out, out2):
        if t is not None and tuple(t.shape) != (m, 1, h, w):
            raise RuntimeError(f"corrnet: output has shape {tuple(t.shape)}, expected {(m, 1, h, w)}")
    wp = (C.c_void_p * len(weight_sets))(*[_dev(t, "corrnet weights").data_ptr() for t in weight_sets])
    se = (C.c_int32 * 3)(*(list(seg_end) + [m] * (3 - len(seg_end))))
    sizes = {t.numel() for t in weight_sets}
    if sizes == {CORRNET_WEIGHT_FLOATS_SPLIT3}:
        check(_lib.load().itermvs_corrnet_bf16x3(ptr, x_sn, wp, se, len(weight_sets), m, h, w, po, o_sn, p2, o2_sn, _stream()),
              "itermvs_corrnet_bf16x3")
        return out
    if sizes != {CORRNET_WEIGHT_FLOATS}:
        raise RuntimeError("corrnet: weight sets must all come from pack_corrnet_weights (one arithmetic)")
    check(_lib.load().itermvs_corrnet(ptr, x_sn, wp, se, len(weight_sets), m, h, w, po, o_sn, p2, o2_sn, _stream()), "itermvs_corrnet")
    return out


def image_pyramid(raw: Tensor, height: int, width: int, all_levels: bool = True, out0: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """datasets/dtu_yao_eval.py:61-74 on the GPU: raw [V,Hs,Ws,3] uint8 RGB (device) -> {'level_0': [V,3,H,W] float32 in
    -1..1, resized like cv2.resize(INTER_LINEAR), 'level_1'..'level_3': the reference's lower pyramid levels}"""
    if not raw.is_cuda or raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[3] != 3:
        raise RuntimeError("image_pyramid: expected a CUDA uint8 tensor [V,Hs,Ws,3]")
    raw = raw.contiguous()
    v, hs, ws, _ = raw.shape
    if out0 is not None and (tuple(out0.shape) != (v, 3, height, width) or not out0.is_contiguous() or out0.dtype != torch.float32):
        raise RuntimeError("image_pyramid: out0 must be a contiguous float32 [V,3,H,W] tensor")
    out = {"level_0": out0 if out0 is not None else torch.empty((v, 3, height, width), device=raw.device, dtype=torch.float32)}
    if all_levels:
        for l in (1, 2, 3):
            out[f"level_{l}"] = torch.empty((v, 3, height >> l, width >> l), device=raw.device, dtype=torch.float32)
    check(_lib.load().itermvs_image_pyramid(raw.data_ptr(), v, hs, ws, height, width, out["level_0"].data_ptr(),
                                            _ptr(out.get("level_1")), _ptr(out.get("level_2")), _ptr(out.get("level_3")), _stream()),
          "itermvs_image_pyramid")
    return out


_PROFILE_MASK = [0x3]


def profile_enable(capacity: int, mask: int = 0x3) -> None:
    """mask: bit 0 corr_iter, bit 1 corr_init, bit 2 every itermvs_conv2d launch"""
    _PROFILE_MASK[0] = mask
    check(_lib.load().itermvs_profile_set_mask(mask), "itermvs_profile_set_mask")
    check(_lib.load().itermvs_profile_enable(capacity), "itermvs_profile_enable")


def profile_graph_count() -> int:
    """event pairs embedded so far in captured hipGraphs (see itermvs_profile_graph_read)"""
    return _lib.load().itermvs_profile_graph_count()


def profile_graph_read(first: int, count: int):
    """-> list of (kind, milliseconds) of pairs [first, first+count) for the latest replay of their graph (waits)"""
    kinds = (C.c_int32 * max(count, 1))()
    ms = (C.c_float * max(count, 1))()
    n = _lib.load().itermvs_profile_graph_read(first, count, kinds, ms)
    return [(kinds[i], ms[i]) for i in range(n)]


def profile_collect(max_samples: int = 4096):
    """-> list of (kind, milliseconds); kind 1 = corr_iter, 2 = corr_init."""
    kinds = (C.c_int32 * max_samples)()
    ms = (C.c_float * max_samples)()
    n = _lib.load().itermvs_profile_collect(kinds, ms, max_samples)
    return [(kinds[i], ms[i]) for i in range(n)]
