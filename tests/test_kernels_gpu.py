"""HIP kernels (through the C ABI, itermvs_amd.ops) vs golden vectors from the reference and
vs the CPU oracle on identical inputs.  Needs an MI355X: run with ``-m gpu``."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, load_weights
from oracle import itermvs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from itermvs_amd import ops as _ops
    return _ops


def cu(t):
    return t.to(DEV)


def maxdiff(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


def proj12_cpu(src_proj, ref_proj):
    """the reference's own fp32 projection (module.py:77-90) as [B,12] rows of [rot|trans]"""
    return O.compose_projection(src_proj, ref_proj)[:, :3, :4].reshape(-1, 12).contiguous()


WARP_CASES = ["l1", "l2_b2", "l3", "init", "l1_behind", "l3_behind"]


def test_compose_proj_matches_fp64_and_reference():
    g = golden("e2e_small_seed0.npz")
    for l in (1, 2, 3):
        mats = g[f"proj.level_{l}"]                       # [B,V,4,4]
        out = ops().compose_proj(cu(mats)).cpu()          # [B,V-1,12]
        for b in range(mats.shape[0]):
            inv = torch.inverse(mats[b, 0].double())
            for s in range(1, mats.shape[1]):
                exact = (mats[b, s].double() @ inv)[:3, :4].reshape(12)
                got = out[b, s - 1].double()
                assert float(((got - exact).abs() / exact.abs().clamp(min=1e-3)).max()) < 1e-6
                ref32 = proj12_cpu(mats[b:b + 1, s], mats[b:b + 1, 0])[0].double()
                assert float(((got - ref32).abs() / ref32.abs().clamp(min=1.0)).max()) < 1e-4


def test_compose_proj_tap_indices_against_reference_projection():
    """Bilinear tap indices floor(ix), floor(iy) with the DEVICE projection (fp64 inverse, rounded once) vs the
    reference's fp32 ``src @ inverse(ref)`` (module.py:77-90; LAPACK on the CPU, MAGMA / cuSOLVER on a GPU -- the reference
    itself has no single bit pattern here), through the same fp32 coordinate arithmetic (the oracle's restatement of
    module.py:89-115), for every level and source view of the cfg-1 geometry and of the golden fixtures.
    The two roundings of the same matrix move the sampling position by < 3e-4 px (a few ulps of a 320-px coordinate); a
    tap index can therefore differ only where the position sits within that distance of an integer, i.e. where the tap
    that changes carries a weight < 3e-4 and the interpolated value is continuous (measured: 2e-5 of the coordinates).  Asserted: position shift, flip rate, and that every flip is of that kind."""
    from itermvs_amd import synthetic
    samples = [synthetic.make_sample(1, 5, 512, 640, seed=0), synthetic.make_scene_sample(5, 512, 640, seed=1),
               synthetic.make_sample(2, 11, 96, 160, seed=5)]
    gen = torch.Generator().manual_seed(3)
    worst_shift, flips, total = 0.0, 0, 0
    for sm in samples:
        for l in (1, 2, 3):
            mats = sm["proj_matrices"][f"level_{l}"].float()
            b, v = mats.shape[:2]
            hh, ww = sm["imgs"]["level_0"].shape[-2:]
            h1, w1 = hh >> l, ww >> l
            h, w = hh // 4, ww // 4
            dev12 = ops().compose_proj(cu(mats)).cpu()                                   # [B,V-1,12]
            depth = 425.0 + 510.0 * torch.rand((b, 4, h, w), generator=gen)
            for k in range(1, v):
                ref44 = O.compose_projection(mats[:, k], mats[:, 0])
                dev44 = torch.cat([dev12[:, k - 1].view(b, 3, 4), torch.tensor([0.0, 0, 0, 1]).expand(b, 1, 4)], 1)
                ix_r, iy_r, _ = O.warp_source_coords(ref44, depth, h1, w1)
                ix_d, iy_d, _ = O.warp_source_coords(dev44, depth, h1, w1)
                inside = (ix_r > -1) & (ix_r < w1) & (iy_r > -1) & (iy_r < h1)
                for a_r, a_d in ((ix_r, ix_d), (iy_r, iy_d)):
                    worst_shift = max(worst_shift, float(((a_r - a_d).abs() * inside).max()))
                    fl = (torch.floor(a_r) != torch.floor(a_d)) & inside
                    flips += int(fl.sum())
                    total += int(inside.sum())
                    # a flipped index sits at an integer boundary of BOTH positions
                    near = torch.minimum((a_r - torch.round(a_r)).abs(), (a_d - torch.round(a_d)).abs())
                    assert float((near * fl).max()) < 3e-4
    print(f"compose_proj tap indices: worst position shift {worst_shift:.2e} px, {flips} of {total} coordinates "
          f"change their floor ({flips / total:.2e})")
    assert worst_shift < 3e-4 and flips / total < 2e-4


def test_compose_proj_nan_flag():
    mats = torch.eye(4).repeat(1, 3, 1, 1).clone()
    mats[0, 0] = 0.0                                      # singular reference -> inf/nan
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops().compose_proj(cu(mats), flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("case", WARP_CASES)
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_warp_seam(case, layout):
    g = golden("warp_cases.npz")
    src, depth = g[f"{case}.src"], g[f"{case}.depth"]
    p12 = proj12_cpu(g[f"{case}.src_proj"], g[f"{case}.ref_proj"])
    s = cu(src)
    if layout == "nhwc":
        s = s.contiguous(memory_format=torch.channels_last)
    warped, mask = ops().warp(s, cu(p12), cu(depth), return_mask=True)
    ref = g[f"{case}.warped"]
    assert warped.shape == ref.shape
    assert maxdiff(warped, ref) <= 5e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(mask.cpu().bool(), g[f"{case}.mask"].bool())     # a boolean of the reference: equality (the seam divides like IEEE)
    zr = ref.abs().sum(1) == 0
    zo = warped.cpu().abs().sum(1) == 0
    assert float((zr != zo).float().mean()) < 1e-3


@pytest.mark.parametrize("case", ["l1", "l2_b2", "l1_behind"])
def test_differentiable_warping_api_and_backward(case):
    """module.py:68 signature incl. the projection composed on the device, and d/d(src_fea)."""
    from itermvs_amd.module import differentiable_warping
    g = golden("warp_cases.npz")
    src = cu(g[f"{case}.src"]).requires_grad_(True)
    warped = differentiable_warping(src, cu(g[f"{case}.src_proj"]), cu(g[f"{case}.ref_proj"]), cu(g[f"{case}.depth"]))
    ref = g[f"{case}.warped"]
    assert maxdiff(warped, ref) <= 3e-4 * max(1.0, float(ref.abs().max()))
    gen = torch.Generator().manual_seed(1)
    gout = torch.randn(ref.shape, generator=gen)
    warped.backward(cu(gout))
    src_c = g[f"{case}.src"].clone().requires_grad_(True)
    O.differentiable_warping(src_c, g[f"{case}.src_proj"], g[f"{case}.ref_proj"], g[f"{case}.depth"]).backward(gout)
    scale = float(src_c.grad.abs().max())
    assert maxdiff(src.grad, src_c.grad) <= 1e-3 * scale


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_ref_quarter(layout):
    g = golden("e2e_small_seed0.npz")
    ref = {l: g[f"feat.level{l}"][:, 0] for l in (1, 2, 3)}
    want = O.ref_feature_quarter(ref)
    want = torch.cat([want[1], want[2], want[3]], 1).permute(0, 2, 3, 1)
    fm = {l: cu(ref[l]) for l in ref}
    if layout == "nhwc":
        fm = {l: t.contiguous(memory_format=torch.channels_last) for l, t in fm.items()}
    got = ops().ref_quarter(fm[1], fm[2], fm[3])
    assert got.shape == want.shape
    assert maxdiff(got, want) <= 1e-6 * max(1.0, float(want.abs().max()))
    if layout == "nhwc":
        # channel-slice views of wider channels-last tensors: a 4-channel-aligned slice is fine (vector path) ...
        wide = {l: torch.cat([torch.zeros_like(t[:, :4]), t, torch.zeros_like(t[:, :4])], 1).contiguous(memory_format=torch.channels_last)
                for l, t in fm.items()}
        got = ops().ref_quarter(*[wide[l][:, 4:4 + fm[l].shape[1]] for l in (1, 2, 3)])
        assert maxdiff(got, want) <= 1e-6 * max(1.0, float(want.abs().max()))
        # ... a slice whose base pointer is not 16-byte aligned is refused with the documented code, not read wrongly
        with pytest.raises(RuntimeError, match="ITERMVS_ERR_ALIGN|aligned"):
            ops().ref_quarter(wide[1][:, 3:3 + fm[1].shape[1]], wide[2][:, 4:4 + fm[2].shape[1]], wide[3][:, 4:4 + fm[3].shape[1]])


def _small(tag):
    g = golden(f"e2e_small_{tag}.npz")
    feats = {l: g[f"feat.level{l}"] for l in (1, 2, 3)}                    # [B,V,C,h,w]
    b, v = feats[1].shape[:2]
    cl = {l: cu(feats[l].reshape(b * v, *feats[l].shape[2:])).contiguous(memory_format=torch.channels_last)
          for l in (1, 2, 3)}
    pv = {l: cl[l].view(b, v, *cl[l].shape[1:]) for l in (1, 2, 3)}
    src = {l: [pv[l][:, i] for i in range(1, v)] for l in (1, 2, 3)}
    ref = {l: pv[l][:, 0] for l in (1, 2, 3)}
    projs = torch.stack([g[f"proj.level_{l}"] for l in (1, 2, 3)])         # [3,B,V,4,4]
    p12 = torch.stack([torch.stack([proj12_cpu(projs[i][:, s], projs[i][:, 0]) for s in range(1, v)], 1)
                       for i in range(3)])                                   # [3,B,S,12] reference fp32 projections
    inv_min = cu(1.0 / g["depth_min"])
    inv_max = cu(1.0 / g["depth_max"])
    return g, src, ref, cu(p12), inv_min, inv_max


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_corr_init_and_aggregate(tag):
    g, src, ref, p12, inv_min, inv_max = _small(tag)
    corr = ops().corr_init(src[3], ref[3], p12[2], inv_min, inv_max, 32)    # [B,S,32,8,h,w]
    b, s = corr.shape[:2]
    for i in range(s):
        want = g[f"init.corr_view{i}"]                                      # [B,8,32,h,w]
        got = corr[:, i].permute(0, 2, 1, 3, 4)
        assert maxdiff(got, want) <= 2e-5 * max(1.0, float(want.abs().max())), i
    # explicit hypotheses give the same result as in-kernel DepthInitialization
    corr2 = ops().corr_init(src[3], ref[3], p12[2], inv_min, inv_max, depth=cu(g["init.samples"]))
    assert maxdiff(corr, corr2) <= 1e-5 * max(1.0, float(corr.abs().max()))
    # view-weighted aggregation against the reference's CorrNet input, using oracle view weights
    w = load_weights(tag)
    vws = [O.pixel_view_weight(w, g[f"init.corr_view{i}"]) for i in range(s)]       # [B,1,h,w] each
    vw = torch.cat(vws, 1)                                                  # [B,S,h,w]
    agg = ops().view_aggregate(corr, cu(vw))                                # [B,32,8,h,w]
    want = g["init.agg"].permute(0, 2, 1, 3, 4)
    assert maxdiff(agg, want) <= 5e-5 * max(1.0, float(want.abs().max()))
    # groups-last storage of the per-view volume ([B,S,N,H,W,8] behind the same logical tensor: the layout PixelViewWeight's
    # 3x3 layer stages with two 16-byte loads per pixel): same bits out of corr_init, same bits through the consumers
    corr_gl = ops().corr_init(src[3], ref[3], p12[2], inv_min, inv_max, 32, groups_last=True)
    assert corr_gl.shape == corr.shape and corr_gl.permute(0, 1, 2, 4, 5, 3).is_contiguous() and torch.equal(corr_gl, corr)
    agg_a, up_a = ops().view_aggregate_up(corr, cu(vw))
    agg_b, up_b = ops().view_aggregate_up(corr_gl, cu(vw))
    assert torch.equal(agg_a, agg) and torch.equal(agg_b, agg) and torch.equal(up_a, up_b)
    h3, w3 = corr.shape[-2:]
    pw = cu(w["iter_mvs.evaluation.pixel_view_weight.conv.0.conv.weight"])
    pk = ops().MfmaWeight(pw, split3=True)
    x_pl = corr.reshape(-1, 8, h3, w3)
    x_cl = corr_gl.reshape(-1, 8, h3, w3)
    assert x_cl.is_contiguous(memory_format=torch.channels_last) and not x_cl.is_contiguous()
    assert torch.equal(ops().conv2d(x_cl, pk, None, act="relu"), ops().conv2d(x_pl, pk, None, act="relu"))


def test_bilinear_up_into_two_destinations():
    gen = torch.Generator().manual_seed(4)
    x = torch.randn((2, 32, 9, 11), generator=gen)
    want = torch.tanh(F.interpolate(x, scale_factor=2, mode="bilinear"))
    a = torch.zeros((2, 32, 18, 22), device=DEV)
    wide = torch.zeros((2, 43, 18, 22), device=DEV)
    ops().bilinear_up_into(cu(x), 2, a, wide[:, :32], act="tanh")
    assert maxdiff(a, want) <= 1e-6 and torch.equal(wide[:, :32], a) and float(wide[:, 32:].abs().max()) == 0.0
    assert torch.equal(a, ops().bilinear_up(cu(x), 2, act="tanh"))


def test_softmax_max():
    gen = torch.Generator().manual_seed(9)
    x = torch.randn((6, 32, 8, 12), generator=gen) * 3
    want = torch.softmax(x, 1).max(1, keepdim=True)[0]
    assert maxdiff(ops().softmax_max(cu(x)), want) <= 1e-6


@pytest.mark.parametrize("shape", [(4, 32, 16, 20), (3, 32, 7, 9), (2, 16, 8, 12), (5, 5, 8, 12)])
def test_pvw_tail_fused(shape):
    """PixelViewWeight's 1x1 layer + softmax over the hypotheses + max in one launch (itermvs.py:343-348)"""
    m, n, h, w = shape
    gen = torch.Generator().manual_seed(m * 10 + n)
    x = torch.relu(torch.randn((m * n, 16, h, w), generator=gen))
    wt = torch.randn((1, 16, 1, 1), generator=gen) * 0.7
    b = torch.randn((1,), generator=gen)
    logits = F.conv2d(x, wt, b).view(m, n, h, w)
    want = torch.softmax(logits, 1).max(1, keepdim=True)[0]
    got = ops().pvw_tail(cu(x), cu(wt), cu(b), n)
    assert got.shape == (m, 1, h, w) and maxdiff(got, want) <= 2e-6
    assert maxdiff(ops().pvw_tail(cu(x), cu(wt), None, n), torch.softmax(F.conv2d(x, wt).view(m, n, h, w), 1).max(1, keepdim=True)[0]) <= 2e-6


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
@pytest.mark.parametrize("mode", ["explicit", "generated"])
def test_corr_iter(tag, mode):
    g, src, ref, p12, inv_min, inv_max = _small(tag)
    ref_q = ops().ref_quarter(ref[1], ref[2], ref[3])
    vw = cu(g["init.view_weights"])
    from itermvs_amd.engine import sample_offsets
    for it in range(int(g.np("iteration"))):
        if mode == "explicit":
            depth = {l: cu(g[f"iter{it}.samples.level{l}"]) for l in (1, 2, 3)}
            aggs = ops().corr_iter(src, ref_q, p12, vw, inv_min, inv_max, depth=depth)
        else:
            aggs = ops().corr_iter(src, ref_q, p12, vw, inv_min, inv_max, norm_depth=cu(g[f"iter{it}.nd_in"]),
                                   offsets=sample_offsets())
        for i, l in enumerate((1, 2, 3)):
            want = g[f"iter{it}.agg.level{l}"].permute(0, 2, 1, 3, 4)       # [B,N,8,h,w]
            assert aggs[i].shape == want.shape
            assert maxdiff(aggs[i], want) <= 5e-5 * max(1.0, float(want.abs().max())), (it, l)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16bit_feature_storage_matches_oracle_on_the_rounded_features(dtype):
    """BASELINE cfg 4 / cfg 5 feature storage: the correlation kernels read fp16 / bf16 pyramids and compute in fp32, so
    on features that ARE representable in 16 bits they must agree with the fp32 oracle like the fp32 path does
    (ref_quarter <= 1e-6, per-view and aggregated correlations <= 5e-5 * scale)."""
    g, src, ref, p12, inv_min, inv_max = _small("dtu")
    r16 = lambda t: t.to(dtype)                                   # round to nearest even, what the conv epilogue does
    src16 = {l: [r16(t) for t in src[l]] for l in src}
    ref16 = {l: r16(ref[l]) for l in ref}
    srcr = {l: [t.float() for t in src16[l]] for l in src}        # the same values held in fp32
    refr = {l: ref16[l].float() for l in ref}
    rq16 = ops().ref_quarter(ref16[1], ref16[2], ref16[3])
    rq32 = ops().ref_quarter(refr[1], refr[2], refr[3])
    assert torch.equal(rq16, rq32)
    vw = cu(g["init.view_weights"])
    from itermvs_amd.engine import sample_offsets
    nd = cu(g["iter0.nd_in"])
    a16 = ops().corr_iter(src16, rq16, p12, vw, inv_min, inv_max, norm_depth=nd, offsets=sample_offsets())
    a32 = ops().corr_iter(srcr, rq32, p12, vw, inv_min, inv_max, norm_depth=nd, offsets=sample_offsets())
    for x, y in zip(a16, a32):
        assert torch.equal(x, y)                                  # same arithmetic on the same values
    c16 = ops().corr_init(src16[3], ref16[3], p12[2], inv_min, inv_max, 32)
    c32 = ops().corr_init(srcr[3], refr[3], p12[2], inv_min, inv_max, 32)
    assert torch.equal(c16, c32)
    # and against the oracle (CPU, fp32) on those rounded features
    b, v = g["feat.level3"].shape[:2]
    f3 = torch.stack([refr[3].cpu()] + [t.cpu() for t in srcr[3]], 1)
    depth = O.initial_depth_samples((1.0 / g["depth_min"]).view(b, 1, 1, 1), (1.0 / g["depth_max"]).view(b, 1, 1, 1), *f3.shape[-2:])
    for sidx in range(v - 1):
        m = torch.cat([p12[2][:, sidx].cpu().view(b, 3, 4), torch.zeros(b, 1, 4)], 1)
        ix, iy, _ = O.warp_source_coords(m, depth, *f3.shape[-2:])
        want = O.group_correlation(O.bilinear_gather(f3[:, sidx + 1], ix, iy), f3[:, 0]).permute(0, 2, 1, 3, 4)
        assert maxdiff(c16[:, sidx], want) <= 5e-5 * max(1.0, float(want.abs().max()))
    with pytest.raises(RuntimeError, match="float32|storage type"):      # the plain warp seam is fp32 only
        ops().warp(src16[1][0], p12[0][:, 0], cu(g["iter0.samples.level1"]))


def _bwd_case(b, v, h, w, seed):
    """random pyramid + cameras + depth for the gradient tests: CPU leaves (requires_grad) and the shared geometry"""
    gen = torch.Generator().manual_seed(seed)
    from itermvs_amd import synthetic
    sm = synthetic.make_sample(b, v, 4 * h, 4 * w, seed=seed)      # (4h, 4w must be multiples of 32)
    sizes = {1: (2 * h, 2 * w), 2: (h, w), 3: (h // 2, w // 2)}
    chans = {1: 16, 2: 32, 3: 48}
    feats = {l: torch.randn((b * v, chans[l]) + sizes[l], generator=gen).requires_grad_(True) for l in (1, 2, 3)}
    projs = torch.stack([sm["proj_matrices"][f"level_{l}"] for l in (1, 2, 3)])
    p12 = torch.stack([torch.stack([proj12_cpu(projs[i][:, s], projs[i][:, 0]) for s in range(1, v)], 1) for i in range(3)])
    inv_min, inv_max = torch.full((b,), 1 / 425.0), torch.full((b,), 1 / 935.0)
    return gen, feats, sizes, chans, p12, inv_min, inv_max


def _oracle_corr(feat, ref, p12_l, depth, b, v, size):
    """per-view group correlations [B,G,N,h,w] of one level with the oracle's differentiable pieces"""
    pv = feat.view(b, v, *feat.shape[1:])
    out = []
    for s in range(1, v):
        m = torch.cat([p12_l[:, s - 1].view(b, 3, 4), torch.zeros(b, 1, 4)], 1)
        with torch.no_grad():
            ix, iy, _ = O.warp_source_coords(m, depth, size[0], size[1])
        out.append(O.group_correlation(O.bilinear_gather(pv[:, s], ix, iy), ref))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("b,v", [(1, 3), (2, 6)])
def test_corr_iter_backward_matches_autograd(b, v, dtype):
    """itermvs_corr_iter_backward (scatter-add to the source features, gather to ref_q) vs torch autograd through the
    oracle's warp + group correlation + view-weighted mean (itermvs.py:84-120), hypotheses built from nd + offsets"""
    from itermvs_amd.engine import sample_offsets
    h, w = 24, 40
    gen, feats, sizes, chans, p12, inv_min, inv_max = _bwd_case(b, v, h, w, 7)
    if dtype != torch.float32:                # 16-bit feature storage: both sides see the rounded values
        feats = {l: f.detach().to(dtype).float().requires_grad_(True) for l, f in feats.items()}
    ref_q = torch.randn((b, h, w, 96), generator=gen).requires_grad_(True)
    vw = torch.rand((b, v - 1, h, w), generator=gen)
    nd = torch.rand((b, 1, h, w), generator=gen)
    samples = O.iteration_depth_samples(nd, inv_min.view(b, 1, 1, 1), inv_max.view(b, 1, 1, 1))
    gw = {l: torch.randn((b, n, 8, h, w), generator=gen) for l, n in ((1, 4), (2, 4), (3, 2))}
    off = {1: 0, 2: 16, 3: 48}
    loss = 0
    for i, l in enumerate((1, 2, 3)):
        refl = ref_q[..., off[l]:off[l] + chans[l]].permute(0, 3, 1, 2)
        acc, wsum = 0, 1e-5
        for s, corr in enumerate(_oracle_corr(feats[l], refl, p12[i], samples[l], b, v, sizes[l])):
            wv = vw[:, s].view(b, 1, 1, h, w)
            acc, wsum = acc + corr * wv, wsum + wv
        loss = loss + ((acc / wsum).permute(0, 2, 1, 3, 4) * gw[l]).sum()
    loss.backward()
    # HIP
    # the fp32 tensors the gradient is routed to, and (16-bit storage) the copies the kernels gather from
    fg = {l: cu(feats[l].detach()).contiguous(memory_format=torch.channels_last).requires_grad_(True) for l in (1, 2, 3)}
    stored = None if dtype == torch.float32 else {l: fg[l].detach().to(dtype) for l in fg}
    rq = cu(ref_q.detach()).requires_grad_(True)
    outs = ops().corr_iter_train(fg, b, v, rq, cu(p12), cu(vw), cu(inv_min), cu(inv_max), cu(nd), sample_offsets(), stored=stored)
    sum((o * cu(gw[l])).sum() for o, l in zip(outs, (1, 2, 3))).backward()
    for l in (1, 2, 3):        # fp32 gradients whatever the storage type (straight-through rounding)
        g_ref = feats[l].grad
        assert fg[l].grad.dtype == torch.float32
        assert maxdiff(fg[l].grad, g_ref) <= 1e-4 * max(1.0, float(g_ref.abs().max())), l
        assert float(fg[l].grad.reshape(b, v, -1)[:, 0].abs().max()) == 0.0            # the reference view of the pyramid gets none here
    assert maxdiff(rq.grad, ref_q.grad) <= 1e-4 * max(1.0, float(ref_q.grad.abs().max()))


@pytest.mark.parametrize("cams", ["regular", "vanishing_line", "singular"])
@pytest.mark.parametrize("b,v", [(1, 3), (2, 5)])
def test_corr_init_backward_matches_autograd(b, v, cams):
    """itermvs_corr_init_backward vs autograd through the oracle (itermvs.py:48-51), 32 hypotheses uniform in inverse depth.
    The gradient to the source views comes from the atomic-free gather through each plane's inverse homography
    (csrc/corr_bwd.hip: init_gather_kernel); ``vanishing_line`` gives the first source view a camera whose depth changes
    sign in the middle of the image (module.py:105-108 sends those pixels outside); ``singular`` gives it three identical
    matrix rows (every reference pixel lands on source pixel (1, 1): no inverse homography exists and that kernel must fall
    back to scanning the reference grid)."""
    h, w = 24, 40
    gen, feats, sizes, chans, p12, inv_min, inv_max = _bwd_case(b, v, h, w, 9)
    if cams == "vanishing_line":
        p12 = p12.clone()
        p12[2][:, 0, 8] = -0.1            # Z = d * (-0.1 x + ..) + t_z: zero near x = 10 of the 20 columns
    if cams == "singular":
        p12 = p12.clone()
        p12[2][:, 0, 0:4] = p12[2][:, 0, 8:12]
        p12[2][:, 0, 4:8] = p12[2][:, 0, 8:12]
    f3 = feats[3]
    h3, w3 = sizes[3]
    depth = O.initial_depth_samples(inv_min.view(b, 1, 1, 1), inv_max.view(b, 1, 1, 1), h3, w3)
    gw = torch.randn((b, v - 1, 32, 8, h3, w3), generator=gen)
    ref = f3.view(b, v, *f3.shape[1:])[:, 0]
    corrs = _oracle_corr(f3, ref, p12[2], depth, b, v, sizes[3])
    sum((c.permute(0, 2, 1, 3, 4) * gw[:, s]).sum() for s, c in enumerate(corrs)).backward()
    fg = cu(f3.detach()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = ops().corr_init_train(fg, b, v, cu(p12[2]), cu(inv_min), cu(inv_max), 32)
    (out * cu(gw)).sum().backward()
    assert maxdiff(fg.grad, f3.grad) <= 1e-4 * max(1.0, float(f3.grad.abs().max()))


def test_corr_init_backward_degenerate_camera_at_full_size_is_bounded():
    """a singular camera in the batch at the largest configuration's level-3 size (1/8 of 1920x1280 = 160x240): round 3's
    gather scanned the whole reference grid per source pixel for such a view (O(P1 * N * P) = 4.7e10 projections, seconds);
    the (view, plane) pairs are now routed to the atomic scatter: the gradient still equals autograd through the oracle, and
    the cost is bounded by the scatter's worst case -- every atomic of the collapsed view lands on ONE source pixel
    (236 M colliding lane-atomics: 0.48 s measured on the MI355X, profiles/r04), not by P1 * N * P"""
    import time
    b, v, h3, w3 = 1, 3, 160, 240
    gen = torch.Generator().manual_seed(3)
    from itermvs_amd import synthetic
    sm = synthetic.make_sample(b, v, 8 * h3, 8 * w3, seed=1)
    pr = sm["proj_matrices"]["level_3"]
    p12 = torch.stack([proj12_cpu(pr[:, s], pr[:, 0]) for s in range(1, v)], 1)            # [B,S,12]
    p12[:, 0, 0:4] = p12[:, 0, 8:12]                                                        # first view: three identical rows
    p12[:, 0, 4:8] = p12[:, 0, 8:12]
    inv_min, inv_max = torch.full((b,), 1 / 425.0), torch.full((b,), 1 / 935.0)
    f3 = torch.randn((b * v, 48, h3, w3), generator=gen).requires_grad_(True)
    depth = O.initial_depth_samples(inv_min.view(b, 1, 1, 1), inv_max.view(b, 1, 1, 1), h3, w3)
    gw = torch.randn((b, v - 1, 32, 8, h3, w3), generator=gen)
    ref = f3.view(b, v, *f3.shape[1:])[:, 0]
    corrs = _oracle_corr(f3, ref, p12, depth, b, v, (h3, w3))
    sum((c.permute(0, 2, 1, 3, 4) * gw[:, s]).sum() for s, c in enumerate(corrs)).backward()
    fg = cu(f3.detach()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = ops().corr_init_train(fg, b, v, cu(p12), cu(inv_min), cu(inv_max), 32)
    gwd = cu(gw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (out * gwd).sum().backward()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    assert maxdiff(fg.grad, f3.grad) <= 1e-4 * max(1.0, float(f3.grad.abs().max()))
    assert ms < 2000.0, ms           # (incl. the element-wise part of the toy loss)
    print(f"degenerate camera, 160x240 level-3 map: backward {ms:.1f} ms")


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_prob_regress_bit_exact_indices(tag):
    g = golden(f"e2e_small_{tag}.npz")
    keys = [("logits0", "best0", "nd0")] + [(f"iter{it}.logits", f"iter{it}.best", f"iter{it}.nd")
                                            for it in range(int(g.np("iteration")))]
    for lk, bk, nk in keys:
        logits = g[lk]
        for layout in ("nchw", "nhwc"):
            x = cu(logits)
            if layout == "nhwc":
                x = x.contiguous(memory_format=torch.channels_last)
            nd, prob, best = ops().prob_regress(x, want_prob=True, want_best=True)
            assert torch.equal(best.cpu(), g[bk]), (lk, layout)              # pixel indices bit-exact
            assert maxdiff(nd, g[nk]) <= 1e-6
            assert maxdiff(prob, torch.softmax(logits, 1)) <= 1e-6
    # in-place destinations inside wider buffers
    b, _, h, w = g["logits0"].shape
    buf0 = torch.zeros((b, 5, h, w), device=DEV)
    buf1 = torch.zeros((b, 5, h, w), device=DEV)
    ops().prob_regress(cu(g["logits0"]), nd_out=[(buf0, 3), (buf1, 1)])
    assert maxdiff(buf0[:, 3:4], g["nd0"]) <= 1e-6 and maxdiff(buf1[:, 1:2], g["nd0"]) <= 1e-6
    assert float(buf0[:, :3].abs().max()) == 0.0 and float(buf1[:, 2:].abs().max()) == 0.0


def test_prob_regress_edges():
    """arg-max at the borders (window clamps, duplicates double-counted) and exact ties (first max)."""
    b, h, w = 1, 2, 40
    logits = torch.full((b, 256, h, w), -5.0)
    for x in range(w):
        logits[0, min(255, x * 7), 0, x] = 3.0          # peaks incl. bins 0..3 (left clamp)
        logits[0, 255 - (x % 6), 1, x] = 3.0            # right clamp
    logits[0, 10, 0, 5] = 3.0                            # tie with bin 35 -> first (10) wins
    logits[0, 200, 1, 7] = 3.0
    p = torch.softmax(logits, 1)
    nd_ref, best_ref = O.window_regression(p)
    nd, _, best = ops().prob_regress(cu(logits), want_best=True)
    assert torch.equal(best.cpu(), best_ref)
    assert maxdiff(nd, nd_ref) <= 1e-6


def _head_reference(x, w1, w2, b2):
    """depth_head[2:5] + regression with torch ops (oracle restatement of itermvs.py:121-126,171-190)"""
    import torch.nn.functional as F
    logits = F.conv2d(F.relu(F.conv2d(x, w1)), w2, b2)
    nd, best = O.window_regression(torch.softmax(logits, 1))
    return logits, nd, best


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_head_regress_fused_matches_chain(tag):
    """itermvs_head_regress (two 1x1 layers + softmax regression in one launch) against the oracle chain and
    against the unfused kernels on the golden hidden states"""
    g = golden(f"e2e_small_{tag}.npz")
    w = load_weights(tag)
    p = "iter_mvs.update.depth_head."
    import torch.nn.functional as F
    w0, w1, w2, b2 = (cu(w[p + k]) for k in ("0.weight", "2.weight", "4.weight", "4.bias"))
    a1, a2 = ops().pack_head_weights(w1, w2)
    for it in range(int(g.np("iteration"))):
        hidden = cu(g[f"iter{it}.hidden"])
        x = F.relu(F.conv2d(hidden, w0, padding=2, dilation=2))
        logits, nd_ref, best_ref = _head_reference(x.cpu(), w1.cpu(), w2.cpu(), b2.cpu())
        nd, best = ops().head_regress(x, a1, a2, b2, want_best=True)
        # the arg-max may legitimately move between bins whose logits differ by rounding noise
        top2 = logits.topk(2, dim=1).values
        clear = ((top2[:, :1] - top2[:, 1:2]) > 1e-4)
        assert torch.equal(best.cpu()[clear], best_ref[clear])
        assert float((nd.cpu() - nd_ref)[clear].abs().max()) <= 2e-6
        assert float(clear.float().mean()) > 0.9
        # in-place destinations inside wider buffers, like the GRU input buffers
        b, _, h, wd = x.shape
        buf0, buf1 = torch.zeros((b, 5, h, wd), device=DEV), torch.zeros((b, 5, h, wd), device=DEV)
        ops().head_regress(x, a1, a2, b2, nd_out=[(buf0, 3), (buf1, 1)])
        assert torch.equal(buf0[:, 3:4], nd) and torch.equal(buf1[:, 1:2], nd)
        assert float(buf0[:, :3].abs().max()) == 0.0 and float(buf1[:, 2:].abs().max()) == 0.0


@pytest.mark.parametrize("w2_form", ["fp32", "bf16x3", "bf16x3_all"])
@pytest.mark.parametrize("size", [(1, 16, 32), (2, 23, 37), (1, 128, 160)])
def test_head_fused_equals_conv_plus_head_regress(size, w2_form):
    """the one-launch depth head (3x3 dilated layer + two 1x1 layers + regression; one tile shared by the four waves of a
    workgroup) against the two-launch form: the two input-channel chunks of the 3x3 layer and the four 64-bin slices of
    the softmax sum are accumulated separately and then added, so sums may differ in the last bit -- arg-max bins equal
    wherever the two best probabilities are not within rounding of each other, normalised depth to 1e-6.
    ``bf16x3``: the 64 -> 256 layer on the bf16 matrix instruction (both operands split exactly into three bf16 terms, six cross
    products, fp32 accumulation: pack_head_w2_split3) -- same gates.  ``bf16x3_all``: the dilated 3x3 layer in that arithmetic as well
    (pack_head_w0_split3; all 32 input channels per matrix instruction, the nine taps split over two waves)."""
    b, h, w = size
    wts = load_weights("seed0")
    p = "iter_mvs.update.depth_head."
    w0, w1, w2, b2 = (cu(wts[p + k]) for k in ("0.weight", "2.weight", "4.weight", "4.bias"))
    gen = torch.Generator().manual_seed(h * w)
    hidden = torch.tanh(torch.randn((b, 32, h, w), generator=gen)).to(DEV)
    pk0 = ops().MfmaWeight(w0)
    a1, a2 = ops().pack_head_weights(w1, w2)
    x = ops().conv2d(hidden, pk0, None, pad=2, dilation=2, act="relu")
    nd_ref, best_ref = ops().head_regress(x, a1, a2, b2, want_best=True)
    if w2_form != "fp32":
        a2 = ops().pack_head_w2_split3(w2)
        assert a2.dtype == torch.bfloat16 and tuple(a2.shape) == (16, 2, 3, 64, 8)
        # the three terms add up to the weight exactly, element (ob, g, p, 16 q + i, j) = W2[ob*16 + i][(2g + j//4)*16 + 4q + j%4]
        back = a2.float().sum(2).reshape(16, 2, 4, 16, 2, 4).permute(0, 3, 1, 4, 2, 5).reshape(256, 64)
        assert torch.equal(back, w2.reshape(256, 64))
        with pytest.raises(RuntimeError):
            ops().head_fused(hidden, pk0, a1, a2[:8], b2)
    if w2_form == "bf16x3_all":
        with pytest.raises(RuntimeError):      # the bf16x3 3x3 layer needs the bf16x3 last layer
            ops().head_fused(hidden, ops().pack_head_w0_split3(w0), a1, ops().pack_head_weights(w1, w2)[1], b2)
        pk0 = ops().pack_head_w0_split3(w0)
        assert pk0.dtype == torch.bfloat16 and tuple(pk0.shape) == (2, 9, 3, 64, 8)
        back0 = pk0.float().sum(2).reshape(2, 9, 4, 16, 2, 4).permute(0, 3, 4, 2, 5, 1).reshape(32, 32, 3, 3)
        assert torch.equal(back0, w0)
    nd, best = ops().head_fused(hidden, pk0, a1, a2, b2, want_best=True)
    flips = float((best != best_ref).float().mean())
    assert flips <= 2e-4, flips
    assert float(((nd - nd_ref).abs() * (best == best_ref)).max()) <= 1e-6
    wide = torch.zeros((b, 43, h, w), device=DEV)
    ops().head_fused(hidden, pk0, a1, a2, b2, nd_out=[(wide, 32)])
    assert torch.equal(wide[:, 32:33], nd) and float(wide[:, :32].abs().max()) == 0.0 and float(wide[:, 33:].abs().max()) == 0.0
    nd2, best2 = ops().head_fused(hidden, pk0, a1, a2, b2, want_best=True)
    assert torch.equal(nd2, nd) and torch.equal(best2, best)                   # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("w2_form", ["fp32", "bf16x3"])
@pytest.mark.parametrize("size", [(1, 16, 32), (2, 23, 37), (1, 128, 160)])
def test_head_fused_with_confidence_head(size, w2_form):
    """itermvs_head_fused_conf: the confidence head (itermvs.py:147-151 + the sigmoid of :198) evaluated in the depth head's launch
    on the same staged tile.  Depth outputs must be bit-identical to itermvs_head_fused; the confidence equals
    sigmoid(conv1x1(relu(conv3x3 dil 2 (hidden)))) by torch to 2e-6 and the separate one-launch form (itermvs_conv2d act
    relu_dot_sigmoid) to rounding (the two input-channel chunks are summed in another order)."""
    import torch.nn.functional as F
    b, h, w = size
    wts = load_weights("dtu")
    p, c = "iter_mvs.update.depth_head.", "iter_mvs.update.confidence_head."
    w0, w1, w2, b2 = (cu(wts[p + k]) for k in ("0.weight", "2.weight", "4.weight", "4.bias"))
    c0, c2w, c2b = (cu(wts[c + k]) for k in ("0.weight", "2.weight", "2.bias"))
    gen = torch.Generator().manual_seed(h * w + 1)
    hidden = torch.tanh(torch.randn((b, 32, h, w), generator=gen)).to(DEV)
    pk0, pkc = ops().MfmaWeight(w0, split3=False), ops().MfmaWeight(c0, split3=False)
    a1, a2 = ops().pack_head_weights(w1, w2)
    if w2_form == "bf16x3":
        a2 = ops().pack_head_w2_split3(w2)
    cdot = torch.cat([c2w.reshape(-1), c2b.reshape(-1)]).contiguous()
    nd_ref, best_ref = ops().head_fused(hidden, pk0, a1, a2, b2, want_best=True)
    conf = torch.full((b, 1, h, w), -1.0, device=DEV)
    nd, best = ops().head_fused(hidden, pk0, a1, a2, b2, want_best=True, conf=(pkc, cdot, conf))
    assert torch.equal(nd, nd_ref) and torch.equal(best, best_ref)
    want = torch.sigmoid(F.conv2d(F.relu(F.conv2d(hidden, c0, padding=2, dilation=2)), c2w, c2b))
    assert maxdiff(conf, want) <= 2e-6, maxdiff(conf, want)
    sep = ops().conv2d(hidden, pkc, None, pad=2, dilation=2, act="relu_dot_sigmoid", aux1=cdot)
    assert maxdiff(conf, sep) <= 1e-6
    with pytest.raises(RuntimeError):
        ops().head_fused(hidden, pk0, a1, a2, b2, conf=(pkc, cdot[:5], conf))


def test_head_regress_edges_and_ties():
    """window clamps at both ends and exact ties (first max): weights that route x straight to chosen bins"""
    h, wd = 2, 40
    x = torch.zeros((1, 32, h, wd))
    w1 = torch.zeros((64, 32, 1, 1))
    w2 = torch.zeros((256, 64, 1, 1))
    b2 = torch.full((256,), -5.0)
    for c in range(32):
        w1[c, c] = 1.0                     # y[c] = relu(x[c])
    # channel c lights bin bins[c]
    bins = [0, 1, 2, 3, 4, 7, 100, 128, 200, 250, 251, 252, 253, 254, 255, 35, 10, 77, 78, 33, 34, 36, 37, 60, 61, 62, 63, 64, 65, 66, 67, 68]
    for c, k in enumerate(bins):
        w2[k, c] = 8.0
    for xx in range(wd):
        x[0, xx % 32, 0, xx] = 1.0                          # one peak: incl. bins 0..4 and 250..255 (clamps)
        x[0, 15, 1, xx] = 1.0                               # tie between bin 35 ...
        x[0, 16 + (xx % 3), 1, xx] = 1.0                    # ... and bin 10 / 77 / 78: the lower index wins
    logits, nd_ref, best_ref = _head_reference(x, w1, w2, b2)
    a1, a2 = ops().pack_head_weights(cu(w1), cu(w2))
    nd, best = ops().head_regress(cu(x), a1, a2, cu(b2), want_best=True)
    assert torch.equal(best.cpu(), best_ref)
    assert maxdiff(nd, nd_ref) <= 1e-6


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("shape", [(4, 8, 64, 80), (3, 16, 33, 20), (2, 64, 8, 10), (1, 5, 130, 131), (20, 8, 128, 160)],
                         ids=["vec", "vec-short-slab", "tiny-planes", "scalar-odd", "many-slabs"])
def test_bn_relu_train_matches_torch(shape, relu):
    """itermvs_bn_train_forward / _backward (csrc/bn.hip) vs torch.nn.functional.batch_norm(training=True) [+ relu] in fp64 on
    the CPU: output, running statistics (unbiased variance, momentum 0.1) and the gradients w.r.t. x, gamma, beta
    (models/module.py:33-50 in train() mode).  Shapes cover the float4 and the scalar form, planes shorter and longer than
    one 8192-float slab, channel means far from zero."""
    gen = torch.Generator().manual_seed(5)
    n, c, h, w = shape
    x = torch.randn(shape, generator=gen) * (1.0 + torch.arange(c).view(1, c, 1, 1)) + 3.0 * torch.arange(c).view(1, c, 1, 1)
    gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen)
    rm, rv = torch.randn(c, generator=gen), torch.rand(c, generator=gen) + 0.5
    dy = torch.randn(shape, generator=gen)
    xg = cu(x).requires_grad_(True)
    gg, bg = cu(gamma).requires_grad_(True), cu(beta).requires_grad_(True)
    rmg, rvg = cu(rm), cu(rv)
    v0 = rmg._version
    yg = ops().bn_relu_train(xg, gg, bg, rmg, rvg, eps=1e-5, momentum=0.1, relu=relu)
    assert rmg._version > v0                               # the in-place update of the running statistics is visible to autograd
    yg.backward(cu(dy))
    # reference in fp64; the ReLU mask of the backward is taken from the GPU's own output (a pre-activation within fp32
    # rounding of zero may fall on either side -- the forward comparison bounds what that can change)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    pre = F.batch_norm(xd, rmd, rvd, gd, bd, training=True, momentum=0.1, eps=1e-5)
    yd = F.relu(pre) if relu else pre
    mask = (yg.detach().cpu() > 0).double() if relu else torch.ones_like(pre)
    (pre * mask).backward(dy.double())
    scale = lambda t: max(1.0, float(t.detach().abs().max()))       # noqa: E731
    assert maxdiff(yg, yd.detach().float()) <= 2e-5 * scale(yd)
    assert maxdiff(rmg, rmd.float()) <= 1e-5 * scale(rmd) and maxdiff(rvg, rvd.float()) <= 1e-5 * scale(rvd)
    assert maxdiff(xg.grad, xd.grad.float()) <= 2e-4 * scale(xd.grad)
    assert maxdiff(gg.grad, gd.grad.float()) <= 2e-4 * scale(gd.grad)
    assert maxdiff(bg.grad, bd.grad.float()) <= 2e-4 * scale(bd.grad)


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_gru_gates(tag):
    g = golden(f"e2e_small_{tag}.npz")
    w = load_weights(tag)
    it = 0
    h_in, nd_in, score = g[f"iter{it}.hidden_in"], g[f"iter{it}.nd_in"], g[f"iter{it}.score"]
    want = g[f"iter{it}.hidden"]
    b, _, hh, ww = h_in.shape
    hx = cu(torch.cat([h_in, nd_in, score], 1)).contiguous()
    hx2 = hx.clone()
    hx2[:, :32] = 0
    p = "iter_mvs.update.gru."
    w_zr = cu(torch.cat([w[p + "convz.weight"], w[p + "convr.weight"]], 0))
    b_zr = cu(torch.cat([w[p + "convz.bias"], w[p + "convr.bias"]], 0))
    zr = F.conv2d(hx, w_zr, b_zr, padding=2, dilation=2)
    ops().gru_rh(zr, hx, hx2)
    q = F.conv2d(hx2, cu(w[p + "convq.weight"]), cu(w[p + "convq.bias"]), padding=2, dilation=2)
    hcopy = torch.empty((b, 32, hh, ww), device=DEV)
    ops().gru_out(zr, q, hx, hcopy)
    assert maxdiff(hx[:, :32], want) <= 2e-5
    assert maxdiff(hcopy, want) <= 2e-5
    assert maxdiff(hx[:, 32:], torch.cat([nd_in, score], 1)) == 0.0


def test_pack_scores():
    gen = torch.Generator().manual_seed(2)
    s = [torch.randn((2, n, 6, 10), generator=gen) for n in (4, 4, 2)]
    d0 = torch.zeros((2, 43, 6, 10), device=DEV)
    d1 = torch.zeros((2, 43, 6, 10), device=DEV)
    ops().pack_scores([cu(t) for t in s], d0, d1, 33)
    want = torch.cat(s, 1)
    assert maxdiff(d0[:, 33:], want) == 0.0 and maxdiff(d1[:, 33:], want) == 0.0
    assert float(d0[:, :33].abs().max()) == 0.0


def test_convex_upsample():
    g = golden("upsample.npz")
    x = g["x"]
    b, _, h, w = x.shape
    buf = torch.zeros((b, 3, h, w), device=DEV)
    buf[:, 1:2] = cu(x)
    inv_min, inv_max = cu(g["inv_min"].view(-1)), cu(g["inv_max"].view(-1))
    for layout in ("nchw", "nhwc"):
        lg = cu(g["logits"])
        if layout == "nhwc":
            lg = lg.contiguous(memory_format=torch.channels_last)
        depth, norm = ops().convex_upsample(lg, buf, inv_min, inv_max, nd_channel=1, want_norm=True)
        assert maxdiff(norm, g["up"]) <= 1e-6
        assert float(((depth.cpu() - g["depth"]).abs() / g["depth"]).max()) <= 1e-6
    from itermvs_amd.module import upsample, depth_unnormalization, depth_normalization
    wsm = torch.softmax(g["logits"].view(b, 1, 9, 4, 4, h, w), dim=2)
    assert maxdiff(upsample(cu(x), cu(wsm)), g["up"]) <= 2e-6
    d = depth_unnormalization(cu(g["up"]), cu(g["inv_min"]), cu(g["inv_max"]))
    assert float(((d.cpu() - g["depth"]).abs() / g["depth"]).max()) <= 1e-6
    assert maxdiff(depth_normalization(cu(g["depth"]), cu(g["inv_min"]), cu(g["inv_max"])), g["renorm"]) <= 1e-5


def test_bilinear_up():
    gen = torch.Generator().manual_seed(4)
    x = torch.randn((2, 5, 7, 9), generator=gen)
    for s in (2, 4):
        assert maxdiff(ops().bilinear_up(cu(x), s), F.interpolate(x, scale_factor=s, mode="bilinear")) <= 1e-6
    assert maxdiff(ops().bilinear_up(cu(x), 2, act="tanh"), torch.tanh(F.interpolate(x, scale_factor=2, mode="bilinear"))) <= 1e-6


def test_ragged_sizes_and_many_views():
    """pixel count not a multiple of the tile, non-integer map/grid ratios, S = 10 source views
    (the pair.txt maximum), B = 2."""
    gen = torch.Generator().manual_seed(12)
    from itermvs_amd import synthetic
    b, v = 2, 11
    sm = synthetic.make_sample(b, v, 96, 160, seed=5)
    h, w = 23, 37                                              # 851 pixels
    sizes = {1: (46, 74), 2: (23, 37), 3: (12, 19)}
    chans = {1: 16, 2: 32, 3: 48}
    feats = {l: torch.randn((b, v, chans[l]) + sizes[l], generator=gen) for l in (1, 2, 3)}
    ref_q = torch.randn((b, h, w, 96), generator=gen)
    projs = torch.stack([sm["proj_matrices"][f"level_{l}"] for l in (1, 2, 3)])
    p12 = torch.stack([torch.stack([proj12_cpu(projs[i][:, s], projs[i][:, 0]) for s in range(1, v)], 1)
                       for i in range(3)])
    vw = torch.rand((b, v - 1, h, w), generator=gen)
    depth = {l: 425 + 510 * torch.rand((b, n, h, w), generator=gen) for l, n in ((1, 4), (2, 4), (3, 2))}
    cl = {l: cu(feats[l].reshape(b * v, *feats[l].shape[2:])).contiguous(memory_format=torch.channels_last)
          for l in feats}
    pv = {l: cl[l].view(b, v, *cl[l].shape[1:]) for l in feats}
    src = {l: [pv[l][:, i] for i in range(1, v)] for l in feats}
    inv_min, inv_max = cu(torch.full((b,), 1 / 425.0)), cu(torch.full((b,), 1 / 935.0))
    aggs = ops().corr_iter(src, cu(ref_q), cu(p12), cu(vw), inv_min, inv_max, depth={l: cu(d) for l, d in depth.items()})
    off = {1: 0, 2: 16, 3: 48}
    for i, l in enumerate((1, 2, 3)):
        refl = ref_q[..., off[l]:off[l] + chans[l]].permute(0, 3, 1, 2)
        acc, wsum = 0, 1e-5
        for s in range(1, v):
            m = torch.cat([p12[i][:, s - 1].view(b, 3, 4), torch.zeros(b, 1, 4)], 1)
            ix, iy, _ = O.warp_source_coords(m, depth[l], sizes[l][0], sizes[l][1])
            corr = O.group_correlation(O.bilinear_gather(feats[l][:, s], ix, iy), refl)
            wv = vw[:, s - 1].view(b, 1, 1, h, w)
            acc = acc + corr * wv
            wsum = wsum + wv
        want = (acc / wsum).permute(0, 2, 1, 3, 4)
        assert float((want != 0).float().mean()) > 0.3          # the case really samples inside the maps
        assert maxdiff(aggs[i], want) <= 5e-5 * max(1.0, float(want.abs().max())), l


def test_error_codes_on_gpu_tensors():
    o = ops()
    with pytest.raises(RuntimeError, match="CUDA"):
        o.bilinear_up(torch.zeros(1, 1, 2, 2), 2)
    x = torch.zeros((1, 20, 4, 4), device=DEV).contiguous(memory_format=torch.channels_last)   # C=20 unsupported
    with pytest.raises(RuntimeError, match="channel"):
        o.corr_init([x], x, torch.zeros((1, 1, 12), device=DEV), torch.ones(1, device=DEV), torch.ones(1, device=DEV))
    y = torch.zeros((1, 48, 4, 4), device=DEV)                                                  # NCHW -> layout error
    with pytest.raises(RuntimeError, match="channels-last"):
        o.corr_init([y], y, torch.zeros((1, 1, 12), device=DEV), torch.ones(1, device=DEV), torch.ones(1, device=DEV))


@pytest.mark.gpu
def test_launches_that_carry_a_second_piece_of_work_equal_the_separate_ones():
    """ref_quarter_compose / view_aggregate_up / final_upsample run two independent kernels' work in one launch (extra
    blocks): results must be bit-identical to the separate entry points, incl. the NaN flag of compose_proj"""
    gen = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(s, generator=gen).to(DEV)
    b, v, h, w = 2, 4, 24, 40
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    r1, r2, r3 = cl(r(b, 16, 2 * h, 2 * w)), cl(r(b, 32, h, w)), cl(r(b, 48, h // 2, w // 2))
    mats = torch.eye(4).repeat(3 * b, v, 1, 1).to(DEV) + 0.1 * r(3 * b, v, 4, 4)
    dmin, dmax = torch.tensor([400.0, 425.0], device=DEV), torch.tensor([900.0, 935.0], device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    proj, imin, imax = ops().compose_proj(mats, flag, (dmin, dmax))
    rq, proj2, imin2, imax2 = ops().ref_quarter_compose(r1, r2, r3, mats, flag, (dmin, dmax))
    assert torch.equal(rq, ops().ref_quarter(r1, r2, r3)) and torch.equal(proj, proj2)
    assert torch.equal(imin, imin2) and torch.equal(imax, imax2) and int(flag.item()) == 0
    bad = mats.clone()
    bad[1, 0] = 0.0                                                   # singular reference camera
    ops().ref_quarter_compose(r1, r2, r3, bad, flag, (dmin, dmax))
    assert int(flag.item()) == 1

    # the composition as the first workgroup of the stem launch (the engine's form): same bits, same flag, same stem results
    from itermvs_amd.engine import fold_batchnorm
    from conftest import load_weights
    wts = {k: v.to(DEV) for k, v in load_weights("dtu").items() if k.startswith("feature_net.")}
    sw = ops().pack_stem_weights(*[t for n in ("conv1.", "layer1.0.conv1.", "layer1.0.downsample.") for t in fold_batchnorm(wts, "feature_net." + n)])
    for mm, hh, ww in ((b * v, 64, 96), (1, 17, 23)):
        x = r(mm, 3, hh, ww)
        flag.zero_()
        y0, sc0 = ops().stem(x, *sw)
        y1, sc1, proj3, imin3, imax3 = ops().stem(x, *sw, compose=(mats, flag, (dmin, dmax)))
        assert torch.equal(y0, y1) and torch.equal(sc0, sc1)
        assert torch.equal(proj, proj3) and torch.equal(imin, imin3) and torch.equal(imax, imax3) and int(flag.item()) == 0
        ops().stem(x, *sw, compose=(bad, flag, (dmin, dmax)))
        assert int(flag.item()) == 1

    s, n, h3, w3 = 3, 32, 12, 20
    corr, vw = r(b, s, n, 8, h3, w3), torch.rand((b, s, h3, w3), generator=gen).to(DEV)
    agg, up = ops().view_aggregate_up(corr, vw)
    assert torch.equal(agg, ops().view_aggregate(corr, vw))
    assert torch.equal(up, ops().bilinear_up(vw.view(b * s, 1, h3, w3), 2).view(b, s, 2 * h3, 2 * w3))

    logits, nd = r(b, 144, h, w), torch.rand((b, 43, h, w), generator=gen).to(DEV)
    conf = torch.rand((b, 1, h, w), generator=gen).to(DEV)
    inv_min, inv_max = 1.0 / dmin, 1.0 / dmax
    depth, conf_up = ops().final_upsample(logits, nd, inv_min, inv_max, conf, nd_channel=32)
    assert torch.equal(depth, ops().convex_upsample(logits, nd, inv_min, inv_max, nd_channel=32))
    assert torch.equal(conf_up, ops().bilinear_up(conf, 4))


@pytest.mark.gpu
def test_copy_multi_stages_a_sample_in_one_launch():
    """itermvs_copy_multi (GraphedRunner's staging of images + cameras + depth range): sizes from 4 bytes to 20 MB, a
    source that is not 16-byte aligned, a dtype other than float"""
    gen = torch.Generator().manual_seed(5)
    src = [torch.randn((1, 5, 3, 512, 640), generator=gen).to(DEV), torch.randn((3, 1, 5, 4, 4), generator=gen).to(DEV),
           torch.randn((1,), generator=gen).to(DEV), torch.randn((1027,), generator=gen).to(DEV)[1:],
           torch.randint(0, 255, (7, 13), generator=gen, dtype=torch.uint8).to(DEV)]
    dst = [torch.zeros_like(t) for t in src]
    ops().copy_multi(dst, src)
    assert all(torch.equal(d, t) for d, t in zip(dst, src))
    with pytest.raises(RuntimeError):
        ops().copy_multi(dst[:1], [src[0].double()])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["cfg1", "cfg3", "cfg5"])
def test_fused_correlation_properties_at_full_size(cfg):
    """BASELINE's full sizes (cfg 1: 5 views 640x512; cfg 3: 5 views 1600x1152; cfg 5: 11 views 1920x1280), where the CPU
    oracle is too slow to run inside a test: size-independent properties of itermvs_corr_iter / itermvs_corr_init.
      * linearity: source features x 2 and reference features x 0.5 are exact in fp32 -> outputs bit-identical; source x 4
        -> outputs x 4 exactly;
      * determinism (two launches, equal bits) and finiteness;
      * a view whose weight is 0 contributes nothing: zeroing its weight == dropping it (same summation order of the rest
        is not guaranteed -> compared to 1e-6);
      * a random subset of 64 pixels is checked against the CPU oracle's arithmetic evaluated ONLY at those pixels (warp,
        module.py:89-115, + bilinear gather + group correlation + view-weighted mean, itermvs.py:84-120)."""
    from itermvs_amd import synthetic
    from itermvs_amd.engine import sample_offsets
    views, hh, ww = {"cfg1": (5, 512, 640), "cfg3": (5, 1152, 1600), "cfg5": (11, 1280, 1920)}[cfg]
    gen = torch.Generator().manual_seed(views)
    sm = synthetic.make_sample(1, views, hh, ww, seed=1)
    s = views - 1
    chans, sizes = {1: 16, 2: 32, 3: 48}, {1: (hh // 2, ww // 2), 2: (hh // 4, ww // 4), 3: (hh // 8, ww // 8)}
    h, w = sizes[2]
    feats = {l: torch.randn((views, chans[l]) + sizes[l], generator=gen) for l in (1, 2, 3)}
    cl = {l: cu(feats[l]).contiguous(memory_format=torch.channels_last) for l in feats}
    src = {l: [cl[l][i:i + 1] for i in range(1, views)] for l in cl}
    ref = {l: cl[l][0:1] for l in cl}
    projs = torch.stack([sm["proj_matrices"][f"level_{l}"].float() for l in (1, 2, 3)])
    p12 = torch.stack([torch.stack([proj12_cpu(projs[i][:, k], projs[i][:, 0]) for k in range(1, views)], 1) for i in range(3)])
    nd = torch.rand((1, 1, h, w), generator=gen)
    vw = torch.rand((1, s, h, w), generator=gen)
    inv_min, inv_max = cu(torch.tensor([1 / 425.0])), cu(torch.tensor([1 / 935.0]))
    offs = sample_offsets()
    ref_q = ops().ref_quarter(ref[1], ref[2], ref[3])
    run = lambda srcs, rq, wts: ops().corr_iter(srcs, rq, cu(p12), wts, inv_min, inv_max, norm_depth=cu(nd), offsets=offs)
    base = run(src, ref_q, cu(vw))
    again = run(src, ref_q, cu(vw))
    scaled = run({l: [t * 2 for t in src[l]] for l in src}, ref_q * 0.5, cu(vw))
    times4 = run({l: [t * 4 for t in src[l]] for l in src}, ref_q, cu(vw))
    for a, b_, c, d in zip(base, again, scaled, times4):
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, b_) and torch.equal(a, c) and torch.equal(a * 4, d)
    vw0 = vw.clone()
    vw0[:, -1] = 0.0
    dropped = ops().corr_iter({l: src[l][:-1] for l in src}, ref_q, cu(p12[:, :, :-1].contiguous()), cu(vw[:, :-1].contiguous()), inv_min, inv_max,
                              norm_depth=cu(nd), offsets=offs)
    for a, b_ in zip(run(src, ref_q, cu(vw0)), dropped):
        assert maxdiff(a, b_) <= 1e-6 * max(1.0, float(b_.abs().max()))
    # 64 random pixels against the oracle's arithmetic evaluated at those pixels only
    pix = torch.randint(0, h * w, (64,), generator=gen)
    ys, xs = pix // w, pix % w
    imin, imax = torch.tensor(1 / 425.0).view(1, 1, 1, 1), torch.tensor(1 / 935.0).view(1, 1, 1, 1)
    samples = O.iteration_depth_samples(nd, imin, imax)
    rq_cpu = ref_q.cpu()
    off = {1: 0, 2: 16, 3: 48}
    for i, l in enumerate((1, 2, 3)):
        n = len(offs[l])
        depth_sel = samples[l][0][:, ys, xs].t().reshape(64, n, 1, 1)              # [64,N,1,1]: one "image" of 1x1 per pixel
        refl = rq_cpu[0, ys, xs, off[l]:off[l] + chans[l]].reshape(64, chans[l], 1, 1)
        acc, wsum = 0, 1e-5
        for k in range(1, views):
            m = torch.cat([p12[i][:, k - 1].view(1, 3, 4), torch.zeros(1, 1, 4)], 1)
            # source coordinates of the selected pixels: the oracle's formula on an explicit pixel list
            rot, trans = m[0, :3, :3], m[0, :3, 3]
            xy1 = torch.stack([xs.float() * (sizes[l][1] / w), ys.float() * (sizes[l][0] / h), torch.ones(64)], 0)       # module.py:95-98
            rxyz = rot @ xy1                                                                                               # [3,64]
            pts = rxyz.t().reshape(64, 1, 3) * depth_sel.reshape(64, n, 1) + trans.view(1, 1, 3)
            X, Y, Z = pts[..., 0], pts[..., 1], pts[..., 2]
            neg = ~(Z > 1e-2)
            X = torch.where(neg, torch.full_like(X, float(w)), X); Y = torch.where(neg, torch.full_like(Y, float(h)), Y)
            Z = torch.where(neg, torch.ones_like(Z), Z)
            gx = (X / Z) / ((sizes[l][1] - 1) / 2) - 1
            gy = (Y / Z) / ((sizes[l][0] - 1) / 2) - 1
            ix = ((gx + 1) / 2) * (sizes[l][1] - 1)
            iy = ((gy + 1) / 2) * (sizes[l][0] - 1)
            warped = O.bilinear_gather(feats[l][k:k + 1].expand(64, -1, -1, -1), ix.view(64, n, 1, 1), iy.view(64, n, 1, 1))   # [64,C,N,1,1]
            corr = O.group_correlation(warped, refl)                                                                       # [64,G,N,1,1]
            wv = vw[0, k - 1, ys, xs].view(64, 1, 1, 1, 1)
            acc, wsum = acc + corr * wv, wsum + wv
        want = (acc / wsum)[:, :, :, 0, 0].permute(0, 2, 1)                         # [64,N,G]
        got = base[i][0].cpu()[:, :, ys, xs].permute(2, 0, 1)                       # [N,G,64] -> [64,N,G]
        # unit-variance NOISE features at coordinates up to 960 px: one fp32 ulp of a coordinate (6e-5 px) moves a bilinear
        # blend by ~1e-4 (measured 8.0e-5 at the cfg-3 size); the smooth-feature gates elsewhere hold 5e-5
        assert maxdiff(got, want) <= 2e-4 * max(1.0, float(want.abs().max())), (cfg, l)
    # the initialisation kernel: same linearity / determinism properties
    ci = ops().corr_init(src[3], ref[3], cu(p12[2]), inv_min, inv_max, 32)
    assert torch.equal(ci, ops().corr_init(src[3], ref[3], cu(p12[2]), inv_min, inv_max, 32))
    assert torch.equal(ci, ops().corr_init([t * 2 for t in src[3]], ref[3] * 0.5, cu(p12[2]), inv_min, inv_max, 32))
    assert bool(torch.isfinite(ci).all()) and tuple(ci.shape) == (1, s, 32, 8) + sizes[3]


@pytest.mark.gpu
@pytest.mark.parametrize("head", ["upsample", "hidden_init"])
@pytest.mark.parametrize("size", [(1, 128, 160), (2, 17, 37), (1, 3, 5), (3, 64, 80)])
def test_conv3x3_conv1x1_one_launch_matches_torch(head, size):
    """itermvs_conv3x3_conv1x1 (csrc/stack2.hip) against the torch layer chain of the two heads it serves:
    IterMVS.upsample (itermvs.py:243-247: conv3x3 32 -> 64, ReLU, conv1x1 64 -> 144, no bias) and Update.hidden_init_head
    (:153-157: conv3x3 32 -> 64, ReLU, conv1x1 64 -> 32 + bias) -- exact fp32 MFMA, 2e-6 of the output range; ragged widths
    (tiles past the right edge), maps smaller than a tile, more tiles than resident workgroups, writing into a wider buffer"""
    import torch.nn.functional as F
    b, h, w = size
    wts = load_weights("dtu")
    p = "iter_mvs.upsample." if head == "upsample" else "iter_mvs.update.hidden_init_head."
    w0, w1 = cu(wts[p + "0.weight"]), cu(wts[p + "2.weight"])
    b1 = cu(wts[p + "2.bias"]) if head == "hidden_init" else None
    no = w1.shape[0]
    gen = torch.Generator().manual_seed(h * w + no)
    x = torch.randn((b, 32, h, w), generator=gen).to(DEV)
    want = F.conv2d(F.relu(F.conv2d(x, w0, padding=1)), w1, b1)
    pk0 = ops().MfmaWeight(w0, split3=False)
    w1p, bp = ops().pack_conv1x1_operand(w1, b1)
    got = ops().conv3x3_conv1x1(x, pk0, w1p, bp, no)
    assert got.shape == want.shape
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 2e-6, err
    wide = torch.full((b, no + 3, h, w), 7.0, device=DEV)
    ops().conv3x3_conv1x1(x, pk0, w1p, bp, no, out=wide[:, 2:2 + no])
    assert torch.equal(wide[:, 2:2 + no], got) and float((wide[:, :2] - 7.0).abs().max()) == 0.0 and float((wide[:, 2 + no:] - 7.0).abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops().conv3x3_conv1x1(x[:, :16], pk0, w1p, bp, no)
    # both layers in the bf16x3 arithmetic (weight_format 3: operands split exactly into three bf16 terms, six cross products)
    a0, a1, bp3 = ops().pack_conv3x3_conv1x1_split3(w0, w1, b1)
    assert a0.dtype == a1.dtype == torch.bfloat16 and tuple(a0.shape) == (4, 9, 3, 64, 8) and tuple(a1.shape) == ((no + 15) // 16, 2, 3, 64, 8)
    # the three terms add up to the weight exactly: element (wv, tap, p, 16 q + i, j) = W0[16 wv + i][(j//4)*16 + 4q + j%4][tap]
    back = a0.float().sum(2).reshape(4, 9, 4, 16, 2, 4).permute(0, 3, 4, 2, 5, 1).reshape(64, 32, 3, 3)
    assert torch.equal(back, w0)
    got3 = ops().conv3x3_conv1x1(x, a0, a1, bp3, no)
    err3 = float((got3 - want).abs().max() / want.abs().max())
    assert got3.shape == want.shape and err3 <= 3e-6, err3
    wide3 = torch.full((b, no + 3, h, w), 7.0, device=DEV)
    ops().conv3x3_conv1x1(x, a0, a1, bp3, no, out=wide3[:, 2:2 + no])
    assert torch.equal(wide3[:, 2:2 + no], got3) and float((wide3[:, :2] - 7.0).abs().max()) == 0.0 and float((wide3[:, 2 + no:] - 7.0).abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops().conv3x3_conv1x1(x, a0, a1[:1] if a1.shape[0] > 1 else a1[:, :1], bp3, no)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 128, 160), (2, 37, 53), (1, 3, 9), (3, 64, 80)])
def test_gru_conv_matches_the_torch_gru(size):
    """itermvs_gru_conv (csrc/gru.hip) against the reference's ConvGRU (models/module.py:53-66: dilated 3x3 gates over
    [h | normalised depth | scores], 43 channels) in float64: z, r*h and the updated state, bf16x3 arithmetic -- 2e-6 of the
    pre-activation range on every output; and against the two itermvs_conv2d launches it replaces.  Ragged widths, maps smaller than
    a tile, more tiles than workgroups, outputs written into channel slices of wider buffers, the state updated in place"""
    import torch.nn.functional as F
    b, h, w = size
    wts = load_weights("dtu")
    p = "iter_mvs.update.gru."
    wz, wr, wq = (cu(wts[p + f"conv{k}.weight"]) for k in "zrq")
    bz, br, bq = (cu(wts[p + f"conv{k}.bias"]) for k in "zrq")
    gen = torch.Generator().manual_seed(h * w + 43)
    hx = torch.randn((b, 43, h, w), generator=gen).to(DEV)
    hx[:, :32] = torch.tanh(hx[:, :32])
    hd, xd = hx[:, :32].double(), hx[:, 32:].double()
    cz = F.conv2d(hx.double(), wz.double(), bz.double(), padding=2, dilation=2)
    cr = F.conv2d(hx.double(), wr.double(), br.double(), padding=2, dilation=2)
    z_want, rh_want = torch.sigmoid(cz), torch.sigmoid(cr) * hd
    wp_zr = ops().pack_gru_conv_split3(torch.cat([wz, wr], 0))
    wp_q = ops().pack_gru_conv_split3(wq)
    assert wp_zr.dtype == torch.bfloat16 and tuple(wp_zr.shape) == (4, 9, 6, 64, 8) and tuple(wp_q.shape) == (2, 9, 6, 64, 8)
    # operands 0..2 add up to the weights of channels 0..31, operands 3, 4 and the l half of 5 to those of 32..42
    back = wp_q[:, :, :3].float().sum(2).reshape(2, 9, 4, 16, 2, 4).permute(0, 3, 4, 2, 5, 1).reshape(32, 32, 3, 3)
    assert torch.equal(back, wq[:, :32])
    tail = (wp_q[:, :, 3].float() + wp_q[:, :, 4].float() + wp_q[:, :, 5].float()).reshape(2, 9, 4, 16, 8)[:, :, :2]   # [ob, tap, half, i, j]
    assert torch.equal(tail.permute(0, 3, 2, 4, 1).reshape(32, 16, 3, 3)[:, :11], wq[:, 32:])
    hx2 = torch.full((b, 45, h, w), 7.0, device=DEV)
    hx2[:, 33:44] = hx[:, 32:]
    zbuf = torch.empty((b, 32, h, w), device=DEV)
    ops().gru_conv(hx, wp_zr, torch.cat([bz, br]), hx[:, :32], zbuf, out2=hx2[:, 1:33])
    scale = float(max(cz.abs().max(), cr.abs().max()))
    assert float((zbuf.double() - z_want).abs().max()) <= 2e-6 * scale
    assert float((hx2[:, 1:33].double() - rh_want).abs().max()) <= 2e-6 * scale
    assert float((hx2[:, :1] - 7.0).abs().max()) == 0.0 and float((hx2[:, 44:] - 7.0).abs().max()) == 0.0
    assert torch.equal(hx2[:, 33:44], hx[:, 32:])
    # the second convolution on what the first produced
    x2 = hx2[:, 1:44].contiguous()
    cq = F.conv2d(x2.double(), wq.double(), bq.double(), padding=2, dilation=2)
    h_want = (1.0 - zbuf.double()) * hd + zbuf.double() * torch.tanh(cq)
    state = hx.clone()
    hidden = torch.empty((b, 32, h, w), device=DEV)
    ops().gru_conv(x2, wp_q, bq, state[:, :32], state[:, :32], out2=hidden, z=zbuf)
    assert float((hidden.double() - h_want).abs().max()) <= 2e-6 * float(cq.abs().max())
    assert torch.equal(state[:, :32], hidden) and torch.equal(state[:, 32:], hx[:, 32:])
    # the two launches of the LDS-tiled kernels it replaces (exact fp32 gates, bf16x3 candidate)
    pk_zr = ops().MfmaWeight(torch.cat([wz, wr], 0), split3=False)
    z_old, rh_old = torch.empty_like(zbuf), torch.empty_like(zbuf)
    ops().conv2d(hx, pk_zr, torch.cat([bz, br]), pad=2, dilation=2, act="sigmoid", out=z_old, aux1=hx[:, :32], split=(32, "gru_rh", rh_old))
    assert float((z_old - zbuf).abs().max()) <= 2e-6 * scale and float((rh_old - hx2[:, 1:33]).abs().max()) <= 2e-6 * scale
    with pytest.raises(RuntimeError):
        ops().gru_conv(hx, wp_q, torch.cat([bz, br]), hx[:, :32], zbuf, out2=hx2[:, 1:33])
    with pytest.raises(RuntimeError):
        ops().gru_conv(hx[:, :40], wp_zr, None, hx[:, :32], zbuf, out2=hx2[:, 1:33])
