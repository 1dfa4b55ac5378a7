"""Pin the CPU oracle (oracle/itermvs_oracle.py) against golden vectors captured
from the real reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import TAP_CASES, check_gradient_slices, golden, load_weights, tap_planes
from oracle import itermvs_oracle as O

WARP_CASES = ["l1", "l2_b2", "l3", "init", "l1_behind", "l3_behind"]


def maxdiff(a, b):
    return float((a - b).abs().max())


@pytest.mark.parametrize("case", WARP_CASES)
def test_warp_matches_reference(case):
    g = golden("warp_cases.npz")
    warped, mask = O.differentiable_warping(g[f"{case}.src"], g[f"{case}.src_proj"], g[f"{case}.ref_proj"],
                                            g[f"{case}.depth"], return_mask=True)
    ref = g[f"{case}.warped"]
    assert warped.shape == ref.shape
    # explicit gather vs grid_sample: pure fp32 re-association noise (<= 4e-5 abs)
    assert maxdiff(warped, ref) <= 5e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(mask, g[f"{case}.mask"].bool())
    # zeros-padding pattern must agree exactly where the reference is exactly zero everywhere in C
    zr = ref.abs().sum(1) == 0
    zo = warped.abs().sum(1) == 0
    assert float((zr != zo).float().mean()) < 1e-3


@pytest.mark.parametrize("case", TAP_CASES)
def test_sampling_positions_floor_equal_the_references(case):
    """tap_cases.npz holds the floor / bounds decisions of the REFERENCE's own sampling grids (observed at F.grid_sample,
    hypotheses built by the reference's DepthInitialization / iteration expressions).  The oracle's warp_source_coords on the
    reference's composed projection reproduces every one of them: its coordinates are the reference's at the bit level where
    it matters (module.py:99-115 restated op for op) -- which lets the GPU tests use the oracle (``ray_dot="fma"``) as the
    tap-index checker at sizes that have no fixture."""
    g = golden("tap_cases.npz")
    lvl, h, w, h1, w1, init = (int(v) for v in g.np(f"{case}.meta"))
    proj, depth, want = g[f"{case}.proj"], g[f"{case}.depth"], g[f"{case}.taps"]
    host_blas_differs = 0
    for s in range(proj.shape[1]):
        # host-independent form (the k-ordered fma dot MKL ran on the golden host): EVERY decision of the reference
        ix, iy, _ = O.warp_source_coords(proj[:, s], depth, h1, w1, ray_dot="fma")
        got = tap_planes(ix, iy, h1, w1)
        assert torch.equal(got, want[:, s]), (case, s, int((got != want[:, s]).sum()))
        # the literal restatement (torch.matmul): equal on an fma BLAS host; an AMD host's MKL path rounds the K = 3 dot
        # without fma and moves ~1e-6 of the floors (measured on the MI355X box's EPYC: 1 of 983 040 decisions)
        ix, iy, _ = O.warp_source_coords(proj[:, s], depth, h1, w1)
        host_blas_differs += int((tap_planes(ix, iy, h1, w1) != want[:, s]).sum())
    assert host_blas_differs <= max(2, int(1e-5 * want.numel())), host_blas_differs
    # the hypotheses themselves: the oracle's constructions equal the reference's (itermvs.py:11-19, :290-293)
    inv_min, inv_max = g[f"{case}.inv_min"].view(-1, 1, 1, 1), g[f"{case}.inv_max"].view(-1, 1, 1, 1)
    if init:
        assert torch.equal(O.initial_depth_samples(inv_min, inv_max, h, w), depth)
    else:
        assert torch.equal(O.iteration_depth_samples(g[f"{case}.nd"], inv_min, inv_max)[lvl], depth)


def test_resize_bilinear_equals_interpolate():
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((2, 5, 8, 12), generator=gen)
    for s in (0.5, 2.0, 4.0):
        assert maxdiff(O.resize_bilinear(x, s), F.interpolate(x, scale_factor=s, mode="bilinear")) <= 1e-6


def test_upsample_and_depth_mapping():
    g = golden("upsample.npz")
    w = torch.softmax(g["logits"].view(2, 1, 9, 4, 4, 6, 10), dim=2)
    up = O.convex_upsample(g["x"], w)
    assert maxdiff(up, g["up"]) <= 1e-6
    d = O.depth_unnormalization(g["up"], g["inv_min"], g["inv_max"])
    assert torch.allclose(d, g["depth"], rtol=1e-6, atol=0)
    assert torch.allclose(O.depth_normalization(g["depth"], g["inv_min"], g["inv_max"]), g["renorm"], rtol=1e-5, atol=1e-6)


def _split(g, it=None):
    feats = {l: g[f"feat.level{l}"] for l in (1, 2, 3)}
    ref_f = {l: feats[l][:, 0] for l in (1, 2, 3)}
    src_f = {l: [feats[l][:, i] for i in range(1, feats[l].shape[1])] for l in (1, 2, 3)}
    projs = {l: g[f"proj.level_{l}"] for l in (1, 2, 3)}
    ref_p = {l: projs[l][:, 0] for l in (1, 2, 3)}
    src_p = {l: [projs[l][:, i] for i in range(1, projs[l].shape[1])] for l in (1, 2, 3)}
    return ref_f, src_f, ref_p, src_p


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_evaluation_init_seam(tag):
    g = golden(f"e2e_small_{tag}.npz")
    w = load_weights(tag)
    ref_f, src_f, ref_p, src_p = _split(g)
    inv_min = (1.0 / g["depth_min"]).view(-1, 1, 1, 1)
    inv_max = (1.0 / g["depth_max"]).view(-1, 1, 1, 1)
    h, wd = ref_f[3].shape[2:]
    samples = O.initial_depth_samples(inv_min, inv_max, h, wd)
    assert torch.allclose(samples, g["init.samples"], rtol=1e-6, atol=0)
    # per-view correlations (input of PixelViewWeight in the reference)
    for s, (fea, proj) in enumerate(zip(src_f[3], src_p[3])):
        corr = O.group_correlation(O.differentiable_warping(fea, proj, ref_p[3], samples), ref_f[3])
        assert maxdiff(corr, g[f"init.corr_view{s}"]) <= 2e-5 * max(1.0, float(g[f"init.corr_view{s}"].abs().max()))
    vw, score, depth, agg = O.evaluation_init(w, ref_f[3], src_f[3], ref_p[3], src_p[3], samples, inv_min, inv_max)
    scale = max(1.0, float(g["init.agg"].abs().max()))
    assert maxdiff(agg, g["init.agg"]) <= 5e-5 * scale
    assert maxdiff(vw, g["init.view_weights"]) <= 1e-4
    assert maxdiff(score, g["init.score"]) <= 1e-4 * max(1.0, float(g["init.score"].abs().max()))
    assert torch.allclose(depth, g["init.depth"], rtol=1e-4)


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_evaluation_iter_seam(tag):
    g = golden(f"e2e_small_{tag}.npz")
    w = load_weights(tag)
    ref_f, src_f, ref_p, src_p = _split(g)
    inv_min = (1.0 / g["depth_min"]).view(-1, 1, 1, 1)
    inv_max = (1.0 / g["depth_max"]).view(-1, 1, 1, 1)
    for it in range(int(g.np("iteration"))):
        samples = O.iteration_depth_samples(g[f"iter{it}.nd_in"], inv_min, inv_max)
        for l in (1, 2, 3):
            assert torch.allclose(samples[l], g[f"iter{it}.samples.level{l}"], rtol=1e-6, atol=0)
        score, aggs = O.evaluation_iter(w, ref_f, src_f, ref_p, src_p, samples, g["init.view_weights"],
                                        return_aggregates=True)
        for l in (1, 2, 3):
            ref = g[f"iter{it}.agg.level{l}"]
            assert maxdiff(aggs[l - 1], ref) <= 5e-5 * max(1.0, float(ref.abs().max())), (it, l)
        assert maxdiff(score, g[f"iter{it}.score"]) <= 1e-4 * max(1.0, float(g[f"iter{it}.score"].abs().max()))


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_update_seam(tag):
    g = golden(f"e2e_small_{tag}.npz")
    w = load_weights(tag)
    iters = int(g.np("iteration"))
    hid0 = O.hidden_init(w, g["init.score"])
    assert maxdiff(hid0, g["hidden0"]) <= 1e-5
    nd0, prob0, best0 = O.depth_init(w, g["hidden0"])
    assert maxdiff(O.depth_head_logits(w, g["hidden0"]), g["logits0"]) <= 1e-4 * max(1.0, float(g["logits0"].abs().max()))
    for it in range(iters):
        hid, nd, prob, conf, conf0, best = O.update_step(w, g[f"iter{it}.hidden_in"], g[f"iter{it}.nd_in"],
                                                         g[f"iter{it}.score"], want_conf=(it == iters - 1))
        assert maxdiff(hid, g[f"iter{it}.hidden"]) <= 1e-5
        if conf is not None:
            assert maxdiff(conf, g[f"iter{it}.conf"]) <= 1e-5
        # the discrete part is pinned on the reference's own logits: bit-exact indices
        p = torch.softmax(g[f"iter{it}.logits"], dim=1)
        nd_ref, best_ref = O.window_regression(p)
        assert torch.equal(best_ref, g[f"iter{it}.best"])
        assert maxdiff(nd_ref, g[f"iter{it}.nd"]) <= 1e-6
        mism = float((best != g[f"iter{it}.best"]).float().mean())
        assert mism <= 0.01, mism
    p0 = torch.softmax(g["logits0"], dim=1)
    nd_ref, best_ref = O.window_regression(p0)
    assert torch.equal(best_ref, g["best0"]) and maxdiff(nd_ref, g["nd0"]) <= 1e-6
    assert maxdiff(torch.softmax(g[f"iter{iters - 1}.logits"], dim=1), g["iter_last.prob"]) <= 1e-6


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_pipeline_small_end_to_end(tag):
    g = golden(f"e2e_small_{tag}.npz")
    w = load_weights(tag)
    imgs = {"level_0": g["imgs"]}
    proj = {f"level_{l}": g[f"proj.level_{l}"] for l in (1, 2, 3)}
    trace = {}
    with torch.no_grad():
        out = O.pipeline_forward(w, imgs, proj, g["depth_min"], g["depth_max"], iteration=int(g.np("iteration")),
                                 test=True, trace=trace)
    for l in (1, 2, 3):
        ref = g[f"feat.level{l}"]
        got = trace["feats"][l].view(ref.shape)
        assert maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    d, dr = out["depths_upsampled"], g["out.depths_upsampled"]
    rel = ((d - dr).abs() / dr.abs())
    # chaotic argmax flips are possible with trained weights: bound the rate, not the max
    assert float((rel > 1e-4).float().mean()) <= (0.0 if tag == "seed0" else 0.02), float(rel.max())
    c, cr = out["confidence_upsampled"], g["out.confidence_upsampled"]
    assert float(((c - cr).abs() > 1e-4).float().mean()) <= (0.0 if tag == "seed0" else 0.02)


@pytest.mark.parametrize("tag", ["seed0", "dtu", "dtu_scene"])
def test_pipeline_cfg1_shape(tag):
    """BASELINE cfg 1 (V=5, 640x512, 4 iterations) on regenerated seeded inputs.

    ``dtu_scene`` (trained weights on photo-consistent views) is the end-to-end parity
    gate: every pixel within 1e-4 relative.  On pure-noise images the network is chaotic
    (one arg-max flip from a 1e-7 perturbation moves a pixel by >10 %),
    so there the median must be exact-ish and the flip RATE is bounded instead."""
    from itermvs_amd import synthetic
    g = golden(f"cfg1_{tag}.npz")
    w = load_weights(tag.split("_")[0])
    if tag.endswith("scene"):
        s = synthetic.make_scene_sample(num_views=5, height=512, width=640, seed=0)
    else:
        s = synthetic.make_sample(batch=1, num_views=5, height=512, width=640, seed=0)
    with torch.no_grad():
        out = O.pipeline_forward(w, s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"], iteration=4)
    d = out["depths_upsampled"][:, :, ::4, ::4]
    rel = (d - g["depth_sub"]).abs() / g["depth_sub"].abs()
    c = out["confidence_upsampled"][:, :, ::4, ::4]
    bad = float((rel > 1e-4).float().mean())
    badc = float(((c - g["conf_sub"]).abs() > 1e-3).float().mean())
    # On the host that produced the golden (same CPU kernels) the restatement reproduces the
    # reference at every pixel of the photo-consistent case; on another host (e.g. the GPU box's EPYC)
    # the reference's own chaos shows up (measured there: 3 % / 37 % of pixels for scene / noise).
    strict = _same_host_as_golden()
    limit = ({"seed0": 0.01, "dtu": 0.08, "dtu_scene": 0.0} if strict else
             {"seed0": 0.03, "dtu": 0.60, "dtu_scene": 0.08})[tag]
    assert float(rel.median()) <= (1e-6 if strict else 1e-4)
    assert bad <= limit, (bad, float(rel.max()))
    assert badc <= limit + (0.0 if strict else 0.02), badc


def _same_host_as_golden() -> bool:
    """canary: the small seeded end-to-end case reproduces the golden bit-for-bit-ish"""
    g = golden("e2e_small_dtu.npz")
    w = load_weights("dtu")
    with torch.no_grad():
        out = O.pipeline_forward(w, {"level_0": g["imgs"]}, {f"level_{l}": g[f"proj.level_{l}"] for l in (1, 2, 3)},
                                 g["depth_min"], g["depth_max"], iteration=int(g.np("iteration")), test=True)
    rel = (out["depths_upsampled"] - g["out.depths_upsampled"]).abs() / g["out.depths_upsampled"]
    return float(rel.max()) <= 1e-6


def test_train_step_loss_and_grads(weights_seed0):
    g = golden("train_small.npz")
    gt = {"level_0": g["gt0"], "level_2": g["gt2"]}
    mk = {"level_0": g["m0"], "level_2": g["m2"]}
    proj = {f"level_{l}": g[f"proj.level_{l}"] for l in (1, 2, 3)}
    for tag, regress in (("regress", True), ("noregress", False)):
        w = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in weights_seed0.items()}
        out = O.pipeline_forward(w, {"level_0": g["imgs"]}, proj, g["depth_min"], g["depth_max"],
                                 iteration=int(g.np("iteration")), test=False, training=True)
        loss = O.full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mk,
                           g["depth_min"], g["depth_max"], regress)
        assert abs(loss.item() - float(g.np(f"{tag}.loss"))) <= 2e-4 * abs(float(g.np(f"{tag}.loss")))
        loss.backward()
        names = [str(n) for n in g.np(f"{tag}.grad_names")]
        norms = g.np(f"{tag}.grad_norms")
        for n, ref in zip(names, norms):
            got = w[n].grad
            if ref < 0:
                assert got is None or float(got.norm()) == 0.0, n
            else:
                assert got is not None, n
                assert abs(float(got.norm()) - ref) <= 2e-3 * max(ref, 1e-3), (n, float(got.norm()), ref)
        # every differentiated parameter's gradient TENSOR (256 evenly spaced elements each), not only its norm
        rep = check_gradient_slices(g, tag, {n: w[n].grad for n in names}, rel_l2=2e-3, min_cos=0.99999)
        print(f"oracle train {tag}: gradient slices {rep}")
        if regress:
            assert maxdiff(out["depths_upsampled"][0], g["train.depths_upsampled"]) <= 1e-4 * 935
            assert maxdiff(out["depths"]["initial"][0], g["train.initial"]) <= 1e-4 * 935
            best = torch.argmax(out["depths"]["probability"][-1], 1, keepdim=True)
            assert float((best != g["train.best_last"]).float().mean()) <= 0.01
            assert maxdiff(w["feature_net.conv1.bn.running_mean"].detach(), g["train.running_mean_conv1"]) <= 1e-6


def test_train_step_cfg4_full_size(weights_seed0):
    """BASELINE cfg 4 at its real size (B=1, 5 views, 640x512, 4 iterations): loss and all 102 gradient norms of the
    reference's training step (tests/golden/train_cfg4.npz); inputs regenerated from seeds"""
    from itermvs_amd import synthetic
    g = golden("train_cfg4.npz")
    sample, gt, mk = synthetic.make_training_sample(num_views=5, height=512, width=640, seed=2)
    torch.set_num_threads(min(16, max(8, torch.get_num_threads())))
    w = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in weights_seed0.items()}
    out = O.pipeline_forward(w, sample["imgs"], sample["proj_matrices"], sample["depth_min"], sample["depth_max"],
                             iteration=int(g.np("iteration")), test=False, training=True)
    loss = O.full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mk, sample["depth_min"],
                       sample["depth_max"], True)
    ref = float(g.np("regress.loss"))
    assert abs(loss.item() - ref) <= 2e-4 * abs(ref), (loss.item(), ref)
    loss.backward()
    for n, want in zip([str(n) for n in g.np("regress.grad_names")], g.np("regress.grad_norms")):
        got = w[n].grad
        if want < 0:
            assert got is None or float(got.norm()) == 0.0, n
        else:
            assert abs(float(got.norm()) - want) <= 5e-3 * max(want, 1e-3), (n, float(got.norm()), want)
    rep = check_gradient_slices(g, "regress", {n: w[n].grad for n in w if w[n].requires_grad}, rel_l2=1e-2, min_cos=0.9999)
    print(f"oracle train cfg4: gradient slices {rep}")
    d = out["depths_upsampled"][0].detach()
    rel = (d[:, :, ::8, ::8] - g["train.depth_sub"]).abs() / g["train.depth_sub"]
    assert float(rel.median()) <= 1e-6 and float((rel > 1e-4).float().mean()) <= 0.02


def test_train_step_cfg4_batch2(weights_seed0):
    """the same step at B = 2 (two scenes; the reference's `batch == 2` branch of differentiable_warping, module.py:78-84, at
    full size): loss, gradient norms and sliced gradients of tests/golden/train_cfg4_b2.npz"""
    from itermvs_amd import synthetic
    g = golden("train_cfg4_b2.npz")
    imgs, projs, dmin, dmax, gt, mk = synthetic.make_training_batch(int(g.np("batch")), num_views=5, height=512, width=640,
                                                                    seed=2, hole_fraction=0.1)
    torch.set_num_threads(min(16, max(8, torch.get_num_threads())))
    w = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in weights_seed0.items()}
    out = O.pipeline_forward(w, imgs, projs, dmin, dmax, iteration=int(g.np("iteration")), test=False, training=True)
    loss = O.full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mk, dmin, dmax, True)
    ref = float(g.np("regress.loss"))
    assert abs(loss.item() - ref) <= 2e-4 * abs(ref), (loss.item(), ref)
    loss.backward()
    for n, want in zip([str(n) for n in g.np("regress.grad_names")], g.np("regress.grad_norms")):
        got = w[n].grad
        if want < 0:
            assert got is None or float(got.norm()) == 0.0, n
        else:
            assert abs(float(got.norm()) - want) <= 5e-3 * max(want, 1e-3), (n, float(got.norm()), want)
    rep = check_gradient_slices(g, "regress", {n: w[n].grad for n in w if w[n].requires_grad}, rel_l2=1e-2, min_cos=0.9999)
    print(f"oracle train cfg4 B=2: gradient slices {rep}")
    d = out["depths_upsampled"][0].detach()
    rel = (d[:, :, ::8, ::8] - g["train.depth_sub"]).abs() / g["train.depth_sub"]
    assert float(rel.median()) <= 1e-6 and float((rel > 1e-4).float().mean()) <= 0.02


@pytest.mark.parametrize("regress", [True, False])
def test_product_full_loss_with_masked_sums_equals_the_indexed_form(regress):
    """itermvs_amd.train_graph.full_loss replaces every ``x[mask].mean()`` of net.py:131-190 by a masked sum over a count (fixed
    shapes, no host synchronisation: the training step can be captured).  Pure torch, so it is held HERE against the pinned
    oracle's literal restatement: loss and gradients, incl. a batch whose windowed selection is EMPTY (net.py:175 skips it)."""
    from itermvs_amd import train_graph as T
    torch.manual_seed(0)
    b, h, w, n = 2, 8, 10, 3
    for spread in (300.0, 3.0):           # 300 mm off: no prediction inside the +-4-bin window -> the empty selection
        probs = [torch.softmax(torch.randn(b, 256, h, w), 1).requires_grad_(True) for _ in range(n)]
        dmin, dmax = torch.tensor([425.0, 430.0]), torch.tensor([935.0, 900.0])
        gt_q, gt_f = 425 + torch.rand(b, 1, h, w) * 500, 425 + torch.rand(b, 1, 4 * h, 4 * w) * 500
        comb = [(gt_q + torch.randn(b, 1, h, w) * spread).requires_grad_(True) for _ in range(n)]
        init = [(gt_q + torch.randn(b, 1, h, w) * 30).requires_grad_(True)]
        up = [(gt_f + torch.randn(b, 1, 4 * h, 4 * w) * 5).requires_grad_(True)]
        confs = [torch.randn(b, 1, h, w).requires_grad_(True) for _ in range(n)]
        mask = {"level_0": (torch.rand(b, 1, 4 * h, 4 * w) > 0.2).float(), "level_2": (torch.rand(b, 1, h, w) > 0.2).float()}
        gt = {"level_0": gt_f, "level_2": gt_q}
        d = {"probability": probs, "combine": comb, "initial": init}
        got = T.full_loss(d, up, confs, gt, mask, dmin, dmax, regress)
        want = O.full_loss(d, up, confs, gt, mask, dmin, dmax, regress)
        assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
        wrt = [probs[0], comb[1], up[0], init[0]] + ([confs[2]] if regress else [])
        for a, r in zip(torch.autograd.grad(got, wrt, allow_unused=True), torch.autograd.grad(want, wrt, allow_unused=True)):
            assert (a is None) == (r is None)
            if a is not None:
                assert float((a - r).abs().max()) <= 1e-6 * float(r.abs().max() + 1e-12)
    # an empty level-2 mask is NaN in both forms (mean over nothing), not silently zero
    mask["level_2"].zero_()
    assert torch.isnan(T.full_loss(d, up, confs, gt, mask, dmin, dmax, regress)) and torch.isnan(O.full_loss(d, up, confs, gt, mask, dmin, dmax, regress))
