"""Chaos-free full-size parity: every HIP stage of the engine is fed the CPU oracle's OWN inputs for that stage
(teacher forcing), so a deviation never feeds back through the arg-max / GRU and the tolerances can be the kernels'
real ones instead of a platform-floor ratio.

For the init stage and each of the GRU iterations (models/itermvs.py:270-324) the oracle's trace provides the stage
inputs; the engine's stage methods (itermvs_amd/engine.py: stage_init / stage_score0 / stage_hidden0 / stage_head /
stage_corr / stage_corrnets / stage_gru / confidence / convex up-sampling) run on them and are compared with the
oracle's stage outputs:

    aggregated correlations (CorrNet inputs)   <= 5e-5 * scale
    CorrNet scores, hidden states, confidence  <= 1e-4 * scale
    arg-max bins                               IDENTICAL wherever the oracle's top-2 logit gap > 1e-4 * max(1, |top logit|)
                                               (>= 99 % of the pixels in every case)
    normalised depth                           <= 1e-4 there
    up-sampled depth / confidence              <= 1e-5 relative / 1e-6

scale = max(1, max |oracle tensor|).  Cases: BASELINE cfg 1 with the published DTU weights (photo-consistent scene and
noise images) and the cfg-3 shape (5 views, 1600x1152) with the seeded weights.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_weights
from oracle import itermvs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
HID = 32


def cu(t):
    return t.to(DEV)


def rel_err(got, want):
    """max |got - want| / max(1, max |want|)"""
    want = want.detach().cpu().float()
    return float((got.detach().cpu().float() - want).abs().max()) / max(1.0, float(want.abs().max()))


FAILS = []


def need(cond, msg):
    """deferred assert: every stage is measured before the test fails, the report lists all deviations"""
    if not cond:
        FAILS.append(msg)


def check_head(eng, ws, w_cpu, hidden_o, best_o, nd_o, tag):
    """depth head + regression on the oracle's hidden state, fused launch and layer-by-layer form"""
    logits_o = O.depth_head_logits(w_cpu, hidden_o)
    top2 = torch.topk(logits_o, 2, dim=1).values
    # the oracle's own arg-max is not a near-tie: top-2 gap > 1e-4 relative to the winning logit (>= 1e-4 absolute)
    decided = (top2[:, :1] - top2[:, 1:2]) > 1e-4 * top2[:, :1].abs().clamp(min=1.0)
    need(float(decided.float().mean()) >= 0.99, f"{tag}: only {float(decided.float().mean()):.4f} decided pixels")
    out = {}
    for form in ("fused", "layers"):
        ws["hidden"].copy_(cu(hidden_o))
        logits, best = eng.stage_head(ws, want_logits=(form == "layers"), want_best=True)
        best = best.cpu()
        nd = ws["hx"][:, HID:HID + 1].cpu()
        need(torch.equal(ws["hx2"][:, HID:HID + 1].cpu(), nd), f"{tag}/{form}: hx2 depth channel differs")   # both GRU input buffers receive it
        flips = (best != best_o) & decided
        need(int(flips.sum()) == 0, f"{tag}/{form}: {int(flips.sum())} arg-max flips at decided pixels")
        err = float(((nd - nd_o).abs() * decided).max())
        need(err <= 1e-4, f"{tag}/{form}: normalised depth off by {err:.2e}")
        if logits is not None:
            need(rel_err(logits, logits_o) <= 1e-4, f"{tag}: logits {rel_err(logits, logits_o):.2e}")
        out[form] = (float((best != best_o).float().mean()), err)
    return out, float(decided.float().mean())


# the last two: BASELINE cfg 5's geometry (10 source views, 8 GRU iterations) on a 512x384 crop, fp32 and with the fp16
# feature storage that configuration names (the oracle models the storage: features rounded to fp16 before the matching
# stages, ``feature_storage``); the full 1920x1280 size is covered by test_full_size_configs_cross_backend
CASES = [("dtu", "scene", 5, 512, 640, 4, "fp32"), ("dtu", "noise", 5, 512, 640, 4, "fp32"), ("seed0", "noise", 5, 1152, 1600, 4, "fp32"),
         ("dtu", "scene", 11, 384, 512, 8, "fp32"), ("dtu", "scene", 11, 384, 512, 8, "fp16")]
STORAGE = {"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}


@pytest.mark.parametrize("wtag,kind,views,height,width,iters,storage", CASES,
                         ids=["cfg1-dtu-scene", "cfg1-dtu-noise", "cfg3-seed0-noise", "cfg5geom-dtu-scene", "cfg5geom-dtu-scene-fp16"])
def test_every_stage_on_the_oracles_inputs(wtag, kind, views, height, width, iters, storage):
    from itermvs_amd import ops, synthetic
    from itermvs_amd.engine import InferenceEngine
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    w_cpu = load_weights(wtag)
    s = (synthetic.make_scene_sample(num_views=views, height=height, width=width, seed=0) if kind == "scene"
         else synthetic.make_sample(batch=1, num_views=views, height=height, width=width, seed=0))
    t = {}
    with torch.no_grad():
        out_o = O.pipeline_forward(w_cpu, s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"], iters, trace=t,
                                   feature_storage=STORAGE[storage])
    eng = InferenceEngine({k: cu(v) for k, v in w_cpu.items()}, iters, storage)
    b, v = 1, views
    sv = v - 1
    h, wd = height // 4, width // 4
    ws = eng._workspace(b, h, wd)
    hx, hx2 = ws["hx"], ws["hx2"]
    report = {}
    FAILS.clear()

    def lim(name, value, bound):
        report[name] = value
        need(value <= bound, f"{name} = {value:.2e} > {bound:.0e}")

    # FeatureNet on the same images (the only stage whose input is not an oracle intermediate)
    with torch.no_grad():
        feats = eng.feature_net(cu(s["imgs"]["level_0"]).reshape(b * v, 3, height, width).contiguous())
    for l in (1, 2, 3):        # 16-bit storage: the engine's features are its fp32 results rounded once (half an fp16 ulp = 2^-12 relative)
        assert feats[l].dtype == (STORAGE[storage] or torch.float32)
        lim(f"feat{l}", rel_err(feats[l], t["feats"][l]), 2e-5 if storage == "fp32" else 5e-4)

    # stage inputs from the oracle: (stored) features, reference-faithful fp32 projections (module.py:77-90), depth range
    cl = {l: cu(t["feats_gathered"][l]).to(feats[l].dtype).contiguous(memory_format=torch.channels_last) for l in (1, 2, 3)}
    for l in (1, 2, 3):        # the oracle's rounded features are exactly representable in the storage type
        assert torch.equal(cl[l].float().cpu(), t["feats_gathered"][l])
    pv = {l: cl[l].view(b, v, *cl[l].shape[1:]) for l in (1, 2, 3)}
    src = {l: [pv[l][:, i] for i in range(1, v)] for l in (1, 2, 3)}
    ref = {l: pv[l][:, 0] for l in (1, 2, 3)}
    pm = [s["proj_matrices"][f"level_{l}"].float() for l in (1, 2, 3)]
    p12 = torch.stack([torch.stack([O.compose_projection(pm[i][:, k], pm[i][:, 0])[:, :3, :4].reshape(-1, 12)
                                    for k in range(1, v)], 1) for i in range(3)]).contiguous()      # [3,B,S,12]
    proj = cu(p12)
    inv_min, inv_max = cu(1.0 / s["depth_min"].float()), cu(1.0 / s["depth_max"].float())
    with torch.no_grad():
        ref_q = ops.ref_quarter(ref[1], ref[2], ref[3])

        # ---- initialisation (itermvs.py:270-276) ----
        tr = {}
        view_w = eng.stage_init(ws, src[3], ref[3], proj[2], inv_min, inv_max, trace=tr)
        lim("init.agg", rel_err(tr["init_agg"], t["init_agg"].permute(0, 2, 1, 3, 4)), 5e-5)
        lim("init.view_w", rel_err(view_w, t["view_weights"]), 1e-4)
        score0 = eng.stage_score0(cu(t["init_agg"].permute(0, 2, 1, 3, 4).contiguous()))
        lim("init.score", rel_err(score0, t["init_score"]), 1e-4)
        eng.stage_hidden0(ws, cu(t["init_score"]))
        lim("hidden0", rel_err(ws["hidden"], t["hidden0"]), 1e-4)
        need(torch.equal(hx[:, :HID], ws["hidden"]), "hidden0: the two copies differ")
        report["head0"], report["decided0"] = check_head(eng, ws, w_cpu, t["hidden0"], t["best0"], t["nd0"], "init")

        # ---- iterations (itermvs.py:288-324) ----
        hidden_in = t["hidden0"]
        view_w_o = cu(t["view_weights"])
        for it, ti in enumerate(t["iters"]):
            hx[:, :HID].copy_(cu(hidden_in))
            hx[:, HID:HID + 1].copy_(cu(ti["nd_in"]))
            hx2[:, HID:HID + 1].copy_(cu(ti["nd_in"]))
            aggs = eng.stage_corr(ws, src, ref_q, proj, view_w_o, inv_min, inv_max)
            for i, a in enumerate(aggs):
                want = ti["aggs"][i].permute(0, 2, 1, 3, 4)
                lim(f"it{it}.agg{i + 1}", rel_err(a, want), 5e-5)
                ws["agg"][i].copy_(cu(want))
            eng.stage_corrnets(ws)
            lim(f"it{it}.score", rel_err(hx[:, HID + 1:], ti["score"]), 1e-4)
            need(torch.equal(hx2[:, HID + 1:], hx[:, HID + 1:]), f"it{it}: score copies differ")
            hx[:, HID + 1:].copy_(cu(ti["score"]))
            hx2[:, HID + 1:].copy_(cu(ti["score"]))
            eng.stage_gru(ws)
            lim(f"it{it}.hidden", rel_err(ws["hidden"], ti["hidden"]), 1e-4)
            need(torch.equal(hx[:, :HID], ws["hidden"]), f"it{it}: hidden copies differ")
            if ti["conf"] is not None:
                ws["hidden"].copy_(cu(ti["hidden"]))
                conf = eng.confidence(ws["hidden"], ws["conf"])
                lim("conf", rel_err(conf, ti["conf"]), 1e-4)
            report[f"it{it}.head"], _ = check_head(eng, ws, w_cpu, ti["hidden"], ti["best"], ti["nd"], f"iter {it}")
            hidden_in = ti["hidden"]

        # ---- final up-sampling (itermvs.py:262-264, 321-324) on the oracle's depth, logits and confidence ----
        last = t["iters"][-1]
        u = "iter_mvs.upsample."
        ref2 = t["feats"][2].view(b, v, *t["feats"][2].shape[1:])[:, 0]
        up_logits_o = F.conv2d(F.relu(F.conv2d(ref2, w_cpu[u + "0.weight"], padding=1)), w_cpu[u + "2.weight"])
        up_logits = eng.upsample_logits(cu(ref2).contiguous(), ws)
        lim("up_logits", rel_err(up_logits, up_logits_o), 1e-4)
        hx[:, HID:HID + 1].copy_(cu(last["nd"]))
        depth_up = ops.convex_upsample(cu(up_logits_o), hx, inv_min, inv_max, nd_channel=HID)
        d, d_o = depth_up.cpu(), out_o["depths_upsampled"]
        lim("depth_up", float(((d - d_o).abs() / d_o).max()), 1e-5)
        conf_up = ops.bilinear_up(cu(last["conf"]), 4)
        lim("conf_up", float((conf_up.cpu() - out_o["confidence_upsampled"]).abs().max()), 1e-6)
    print(f"teacher-forced {wtag}/{kind} {width}x{height}: " + ", ".join(
        f"{k}={v:.1e}" if isinstance(v, float) else f"{k}={v}" for k, v in report.items()))
    assert not FAILS, "; ".join(FAILS)


# ---------------------------------------------------------------------------------------------------------------------
# Teacher-forced BACKWARD at BASELINE cfg 4's size (round 4).  The training graph is chaotic end to end (an arg-max flip moves
# a regression window), so the end-to-end gradient gates of tests/test_train_gpu.py carry a measured chaos floor.  Here the
# fused correlation autograd Functions (the only hand-written backward on the path besides BatchNorm) are fed the pinned
# ORACLE's own tensors of a cfg-4 training step -- its FeatureNet pyramids (bf16-rounded for cfg 4 "as stated"), its view
# weights, its normalised depth going into a GRU iteration, and the UPSTREAM gradient dL/d(aggregated correlation) that the
# oracle's full_loss backward delivers to that Evaluation call -- and their input gradients are compared with torch autograd
# through the oracle's warp + group correlation + view-weighted mean on the same tensors.  Nothing feeds back: the
# tolerance is 1e-4 x scale at 640x512 with B = 2, like the 24x40 kernel tests.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_corr_views(feat_pv, ref, p12_l, depth, size):
    """per-view group correlations [B,G,N,h,w] of one level: feat_pv [B,V,C,H1,W1], the oracle's differentiable pieces"""
    b, v = feat_pv.shape[:2]
    out = []
    for s in range(1, v):
        m = torch.cat([p12_l[:, s - 1].view(b, 3, 4), torch.zeros(b, 1, 4)], 1)
        with torch.no_grad():
            ix, iy, _ = O.warp_source_coords(m, depth, size[0], size[1])
        out.append(O.group_correlation(O.bilinear_gather(feat_pv[:, s], ix, iy), ref))
    return out


@pytest.fixture(scope="module")
def cfg4_backward_trace():
    """one cfg-4 shaped training step of the CPU oracle at B = 2 (5 views, 640x512, 4 iterations, seed-0 weights, the batch
    train.py --batch_size 2 builds), for fp32 and bf16 feature storage: traced tensors + the upstream gradients of every
    Evaluation call (retain_grad on the aggregated correlations)"""
    from itermvs_amd import synthetic
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    imgs, projs, dmin, dmax, gt, mk = synthetic.make_training_batch(2, num_views=5, height=512, width=640, seed=2, hole_fraction=0.1)
    out = {}
    for name, storage in (("fp32", None), ("bf16", torch.bfloat16)):
        w = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in load_weights("seed0").items()}
        tr = {}
        res = O.pipeline_forward(w, imgs, projs, dmin, dmax, iteration=4, test=False, training=True, trace=tr, feature_storage=storage)
        for it in tr["iters"]:
            for a in it["aggs"]:
                a.retain_grad()
        O.full_loss(res["depths"], res["depths_upsampled"], res["confidences"], gt, mk, dmin, dmax, True).backward()
        out[name] = dict(
            feats={l: tr["feats_gathered"][l].detach() for l in (1, 2, 3)},
            view_w=tr["view_weights"].detach(),
            iters=[dict(nd_in=it["nd_in"].detach(), aggs=[a.detach() for a in it["aggs"]], up=[a.grad.detach() for a in it["aggs"]])
                   for it in tr["iters"]])
    p = torch.stack([projs[f"level_{l}"] for l in (1, 2, 3)])
    p12 = torch.stack([torch.stack([O.compose_projection(p[i][:, s], p[i][:, 0])[:, :3, :4].reshape(-1, 12) for s in range(1, 5)], 1)
                       for i in range(3)])
    return out, p12, (1.0 / dmin), (1.0 / dmax)


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("it", [0, 2])
def test_corr_iter_backward_on_the_oracles_training_tensors(cfg4_backward_trace, storage, it):
    """itermvs_corr_iter_backward at 640x512, B = 2, 4 source views: dL/dsrc (scatter) and dL/dref_q (gather) for the
    oracle's upstream gradient of GRU iteration ``it`` (0: hypotheses around the first, noisy depth map; 2: a smooth one)"""
    from itermvs_amd import ops
    from itermvs_amd.engine import sample_offsets
    traces, p12, inv_min, inv_max = cfg4_backward_trace
    t = traces[storage]
    b, v = 2, 5
    feats = {l: t["feats"][l].clone().requires_grad_(True) for l in (1, 2, 3)}                  # [B*V,C,H,W], values as stored
    pv = {l: feats[l].view(b, v, *feats[l].shape[1:]) for l in (1, 2, 3)}
    rq = O.ref_feature_quarter({l: t["feats"][l].view(b, v, *feats[l].shape[1:])[:, 0] for l in (1, 2, 3)})   # (no graph: a leaf below)
    ref_q = torch.cat([rq[1], rq[2], rq[3]], 1).permute(0, 2, 3, 1).contiguous().requires_grad_(True)   # [B,h,w,96]
    assert ref_q.is_leaf
    h, w = ref_q.shape[1:3]
    nd, vw = t["iters"][it]["nd_in"], t["view_w"]
    samples = O.iteration_depth_samples(nd, inv_min.view(b, 1, 1, 1), inv_max.view(b, 1, 1, 1))
    off, chans, loss = {1: 0, 2: 16, 3: 48}, {1: 16, 2: 32, 3: 48}, 0
    for i, l in enumerate((1, 2, 3)):
        refl = ref_q[..., off[l]:off[l] + chans[l]].permute(0, 3, 1, 2)
        acc, wsum = 0, 1e-5
        for s, corr in enumerate(_oracle_corr_views(pv[l], refl, p12[i], samples[l], feats[l].shape[2:])):
            wv = vw[:, s].view(b, 1, 1, h, w)
            acc, wsum = acc + corr * wv, wsum + wv
        agg = acc / wsum
        want_agg = t["iters"][it]["aggs"][i]
        assert float((agg.detach() - want_agg).abs().max()) <= 1e-5 * max(1.0, float(want_agg.abs().max()))   # the traced call, restated
        loss = loss + (agg * t["iters"][it]["up"][i]).sum()
    loss.backward()
    dev = lambda x: x.to("cuda")
    fg = {l: dev(t["feats"][l]).contiguous(memory_format=torch.channels_last).requires_grad_(True) for l in (1, 2, 3)}
    stored = None if storage == "fp32" else {l: fg[l].detach().to(torch.bfloat16) for l in fg}
    if stored is not None:
        assert all(torch.equal(stored[l].float(), fg[l].detach()) for l in fg)                   # the oracle's storage model == bf16 values
    rqd = dev(ref_q.detach()).requires_grad_(True)
    outs = ops.corr_iter_train(fg, b, v, rqd, dev(p12), dev(vw), dev(inv_min), dev(inv_max), dev(nd), sample_offsets(), stored=stored)
    worst = 0.0
    for o, i in zip(outs, range(3)):
        want = t["iters"][it]["aggs"][i].permute(0, 2, 1, 3, 4)
        worst = max(worst, float((o.detach().cpu() - want).abs().max()) / max(1.0, float(want.abs().max())))
    assert worst <= 5e-5, worst                                                                  # forward, like the inference gate
    sum((o * dev(t["iters"][it]["up"][i].permute(0, 2, 1, 3, 4))).sum() for i, o in enumerate(outs)).backward()
    rep = {}
    for l in (1, 2, 3):
        g_ref = feats[l].grad
        scale = max(1e-30, float(g_ref.abs().max()))
        rep[l] = float((fg[l].grad.cpu() - g_ref).abs().max()) / scale
        assert rep[l] <= 1e-4, (l, rep[l], scale)
        assert float(fg[l].grad.reshape(b, v, -1)[:, 0].abs().max()) == 0.0                     # reference view: through ref_q only
    scale = max(1e-30, float(ref_q.grad.abs().max()))
    rep["ref_q"] = float((rqd.grad.cpu() - ref_q.grad).abs().max()) / scale
    assert rep["ref_q"] <= 1e-4, rep
    print(f"teacher-forced backward, iteration {it}, {storage}: forward {worst:.1e}, gradients (max abs / scale) {rep}")


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_corr_init_backward_on_the_oracles_training_features(cfg4_backward_trace, storage):
    """itermvs_corr_init_backward at the cfg-4 size (level 3: 64x80, 32 planes, B = 2, 4 source views) on the oracle's
    FeatureNet output; upstream gradient: seeded noise (the per-view volumes are not a traced seam of the oracle)"""
    from itermvs_amd import ops
    traces, p12, inv_min, inv_max = cfg4_backward_trace
    b, v = 2, 5
    f3 = traces[storage]["feats"][3].clone().requires_grad_(True)
    h3, w3 = f3.shape[2:]
    depth = O.initial_depth_samples(inv_min.view(b, 1, 1, 1), inv_max.view(b, 1, 1, 1), h3, w3)
    gw = torch.randn((b, v - 1, 32, 8, h3, w3), generator=torch.Generator().manual_seed(5))
    pv = f3.view(b, v, *f3.shape[1:])
    corrs = _oracle_corr_views(pv, pv[:, 0], p12[2], depth, (h3, w3))
    sum((c.permute(0, 2, 1, 3, 4) * gw[:, s]).sum() for s, c in enumerate(corrs)).backward()
    fg = f3.detach().to("cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    stored = None if storage == "fp32" else fg.detach().to(torch.bfloat16)
    out = ops.corr_init_train(fg, b, v, p12[2].to("cuda"), inv_min.to("cuda"), inv_max.to("cuda"), 32, stored=stored)
    want = torch.stack([c.detach().permute(0, 2, 1, 3, 4) for c in corrs], 1)
    assert float((out.detach().cpu() - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max()))
    (out * gw.to("cuda")).sum().backward()
    scale = float(f3.grad.abs().max())
    err = float((fg.grad.cpu() - f3.grad).abs().max()) / scale
    assert err <= 1e-4, (err, scale)
    print(f"teacher-forced init backward, {storage}: {err:.1e} of the gradient scale {scale:.3g}")
