#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container, where the upstream repository is mounted
read-only at /root/reference.  It imports ``models.net`` from there, feeds it
seeded synthetic inputs (itermvs_amd.synthetic) and records inputs / outputs at
the reference's own seams (forward hooks on its sub-modules, no source edits).
Only DATA is written: inputs, expected outputs, the seeded weights and the
numbers of the published DTU checkpoint.  The reference's code never ships.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("ITERMVS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from itermvs_amd import synthetic  # noqa: E402
from itermvs_amd.schema import check_state_dict, strip_module_prefix  # noqa: E402

import models.module as ref_module  # noqa: E402  (reference, read-only)
import models.net as ref_net  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


def grad_slice(g, n=256):
    """``n`` evenly spaced elements of a gradient tensor (all of it when smaller): what the full-tensor gradient checks of
    tests/test_train_gpu.py compare (direction and magnitude, not just the norm).  Same indices as tests/conftest.py:grad_slice."""
    flat = g.detach().reshape(-1)
    if flat.numel() <= n:
        return flat.clone()
    idx = torch.linspace(0, flat.numel() - 1, n).round().long()
    return flat[idx]


def record_gradients(model, arrays, tag):
    """per-parameter gradient norms (-1 = the reference leaves .grad None) and the concatenated gradient slices"""
    names, norms, slices, lens = [], [], [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        sl = grad_slice(p.grad) if p.grad is not None else torch.zeros(0)
        slices.append(sl.float())
        lens.append(sl.numel())
    arrays[f"{tag}.grad_names"] = np.array(names)
    arrays[f"{tag}.grad_norms"] = np.array(norms, dtype=np.float64)
    arrays[f"{tag}.grad_slices"] = npy(torch.cat(slices))
    arrays[f"{tag}.grad_slice_len"] = np.array(lens, dtype=np.int64)


def build_reference(weights, iteration, test):
    m = ref_net.Pipeline(iteration=iteration, test=test)
    m.load_state_dict(weights, strict=True)
    return m.eval() if test else m.train()


# ----------------------------------------------------------------------------
# 1. weights
# ----------------------------------------------------------------------------
def golden_weights():
    w0 = synthetic.random_state_dict(0)
    check_state_dict(w0)
    save("weights_seed0.npz", **{k: npy(v) for k, v in w0.items()})
    ckpt = torch.load(os.path.join(REF, "checkpoints/dtu/model_000015.ckpt"), map_location="cpu",
                      weights_only=False)
    wd = strip_module_prefix(ckpt["model"])
    check_state_dict(wd)
    save("weights_dtu.npz", **{k: npy(v) for k, v in wd.items()})
    return w0, wd


# ----------------------------------------------------------------------------
# 2. differentiable_warping seam (module.py:68)
# ----------------------------------------------------------------------------
def golden_warp():
    gen = torch.Generator().manual_seed(7)
    full_h, full_w = 64, 96
    out = {}
    cases = [
        # name, B, C, level of source map, N, sample-grid divisor, behind-camera view?
        ("l1", 1, 16, 1, 4, 4, False),
        ("l2_b2", 2, 32, 2, 4, 4, False),      # batch==2 branch, module.py:78-84
        ("l3", 1, 48, 3, 2, 4, False),
        ("init", 1, 48, 3, 32, 8, False),
        ("l1_behind", 1, 16, 1, 4, 4, True),   # negative-depth patch lands in-bounds at level 1
        ("l3_behind", 1, 48, 3, 2, 4, True),
    ]
    for name, b, c, lvl, n, div, behind in cases:
        sample = synthetic.make_sample(batch=b, num_views=3, height=full_h, width=full_w, seed=3)
        pm = sample["proj_matrices"][f"level_{lvl}"].clone()
        if behind:
            # turn the source camera by ~100 deg about y so a band of pixels projects behind it
            ang = np.radians(100.0)
            ry = torch.tensor([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0],
                               [-np.sin(ang), 0, np.cos(ang), 600.0], [0, 0, 0, 1]], dtype=torch.float32)
            pm[:, 1] = pm[:, 1] @ ry
        h1, w1 = full_h >> lvl, full_w >> lvl
        h, w = full_h // div, full_w // div
        src = torch.randn((b, c, h1, w1), generator=gen)
        depth = 425.0 + torch.rand((b, n, h, w), generator=gen) * 510.0
        warped, mask = ref_module.differentiable_warping(src, pm[:, 1], pm[:, 0], depth, return_mask=True)
        out.update({f"{name}.src": npy(src), f"{name}.src_proj": npy(pm[:, 1]), f"{name}.ref_proj": npy(pm[:, 0]),
                    f"{name}.depth": npy(depth), f"{name}.warped": npy(warped), f"{name}.mask": npy(mask)})
        frac0 = float((warped.abs().sum(1) == 0).float().mean())
        print(f"  warp {name}: zero-fraction {frac0:.3f}, valid-fraction {float(mask.float().mean()):.3f}")
    save("warp_cases.npz", **out)


# ----------------------------------------------------------------------------
# 2b. the reference's own SAMPLING GRIDS: tap indices of module.py:99-119
# ----------------------------------------------------------------------------
def tap_planes(grid, num_depth, height, width, h1, w1):
    """floor / bounds decisions of F.grid_sample(bilinear, zeros, align_corners=True) on the reference's grid
    (ATen GridSampler.h:31 un-normalisation ``((g + 1) / 2) * (size - 1)``, :205-207 floor, within_bounds_2d), in fp32 like
    the op: grid [B, N*H, W, 2] -> int32 [B,N,3,H,W] = (floor(ix), floor(iy), bits); bit 0: x0 inside, 1: x0+1, 2: y0, 3: y0+1.
    Same saturation as itermvs_tap_indices: NaN -> INT32_MIN, +-2^30."""
    b = grid.shape[0]
    g = grid.view(b, num_depth, height, width, 2)
    ix = ((g[..., 0] + 1) / 2) * (w1 - 1)
    iy = ((g[..., 1] + 1) / 2) * (h1 - 1)
    fx, fy = torch.floor(ix), torch.floor(iy)

    def to_int(f):
        o = torch.clamp(torch.nan_to_num(f, nan=0.0, posinf=2.0 ** 30, neginf=-2.0 ** 30), -2.0 ** 30, 2.0 ** 30).to(torch.int32)
        return torch.where(torch.isnan(f), torch.full_like(o, -2 ** 31), o)

    bits = ((fx >= 0) & (fx <= w1 - 1)).int() | (((fx + 1 >= 0) & (fx + 1 <= w1 - 1)).int() << 1) | \
           (((fy >= 0) & (fy <= h1 - 1)).int() << 2) | (((fy + 1 >= 0) & (fy + 1 <= h1 - 1)).int() << 3)
    return torch.stack([to_int(fx), to_int(fy), bits.to(torch.int32)], 2)


def reference_taps(pm, depth, h1, w1):
    """run the reference's differentiable_warping per source view with torch.matmul and F.grid_sample observed (no source
    edits): -> (its composed proj [B,S,4,4], tap planes int32 [B,S,N,3,H,W])"""
    b, n, h, w = depth.shape
    seen = {}
    real_mm, real_gs = torch.matmul, torch.nn.functional.grid_sample

    def mm(a, bb, *args, **kw):
        out = real_mm(a, bb, *args, **kw)
        if tuple(out.shape[-2:]) == (4, 4) and "proj" not in seen:
            seen["proj"] = out.clone()
        return out

    def gs(inp, grid, *args, **kw):
        seen["grid"] = grid.clone()
        return real_gs(inp, grid, *args, **kw)

    projs, planes = [], []
    src = torch.zeros((b, 1, h1, w1))
    for v in range(1, pm.shape[1]):
        seen.clear()
        torch.matmul, torch.nn.functional.grid_sample = mm, gs
        try:
            ref_module.differentiable_warping(src, pm[:, v], pm[:, 0], depth)
        finally:
            torch.matmul, torch.nn.functional.grid_sample = real_mm, real_gs
        projs.append(seen["proj"])
        planes.append(tap_planes(seen["grid"], n, h, w, h1, w1))
    return torch.stack(projs, 1), torch.stack(planes, 1)


def golden_taps():
    """tap_cases.npz: for DTU-like cameras, the floor / bounds decisions of the reference's sampling for the hypotheses the
    reference itself builds (itermvs.py:11-19 and :290-293) -- the `pixel indices bit-exact` pin of the fused kernels."""
    import models.itermvs as ref_itermvs
    gen = torch.Generator().manual_seed(23)
    out = {}
    core = ref_itermvs.IterMVS(iteration=1, feature_dim=32, hidden_dim=32, test=True)

    def iter_samples(nd, inv_min, inv_max, lvl):           # itermvs.py:290-293, the reference's own expressions
        ns = nd + core.corr_interval[f"level{lvl}"] * core.interval_scale
        ns = torch.clamp(ns, min=0, max=1)
        return ref_module.depth_unnormalization(ns, inv_min, inv_max)

    cases = [
        # name, batch, image H, W, views, level, kind
        ("cfg1_l1", 1, 512, 640, 5, 1, "noise"),
        ("cfg1_init", 1, 512, 640, 5, 3, "init"),
        ("mid_l2", 1, 192, 256, 5, 2, "noise"),
        ("mid_l3", 1, 192, 256, 5, 3, "noise"),
        ("mid_l1_smooth", 1, 192, 256, 5, 1, "smooth"),
        ("b2_l2", 2, 96, 128, 3, 2, "noise"),             # batch == 2 branch, module.py:78-84
        ("behind_l1", 1, 96, 128, 3, 1, "behind"),        # Z <= 1e-2 -> (W, H, 1), lands inside the level-1 map
        ("behind_l3", 1, 96, 128, 3, 3, "behind"),
    ]
    for name, b, hh, ww, views, lvl, kind in cases:
        sample = synthetic.make_sample(batch=b, num_views=views, height=hh, width=ww, seed=5)
        pm = sample["proj_matrices"][f"level_{lvl}"].clone()
        if kind == "behind":
            ang = np.radians(100.0)
            ry = torch.tensor([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0],
                               [-np.sin(ang), 0, np.cos(ang), 600.0], [0, 0, 0, 1]], dtype=torch.float32)
            pm[:, 1] = pm[:, 1] @ ry
        dmin, dmax = sample["depth_min"].float(), sample["depth_max"].float()
        inv_min, inv_max = (1.0 / dmin).view(b, 1, 1, 1), (1.0 / dmax).view(b, 1, 1, 1)       # itermvs.py:267-268
        h1, w1 = hh >> lvl, ww >> lvl
        arrays = {}
        if kind == "init":
            h, w = hh // 8, ww // 8
            depth = core.depth_initialization(inv_min, inv_max, h, w, torch.device("cpu"))   # itermvs.py:270
        else:
            h, w = hh // 4, ww // 4
            if kind == "smooth":
                yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
                nd = (0.3 + 0.3 * xx + 0.1 * yy + 0.002 * torch.randn((h, w), generator=gen)).view(1, 1, h, w).repeat(b, 1, 1, 1)
            else:
                nd = torch.rand((b, 1, h, w), generator=gen)
            depth = iter_samples(nd, inv_min, inv_max, lvl)
            arrays["nd"] = npy(nd)
        proj, planes = reference_taps(pm, depth.contiguous(), h1, w1)
        arrays.update(proj=npy(proj), depth=npy(depth), taps=npy(planes), inv_min=npy(inv_min.view(b)), inv_max=npy(inv_max.view(b)),
                      meta=np.array([lvl, h, w, h1, w1, 1 if kind == "init" else 0], dtype=np.int64))
        for k, v in arrays.items():
            out[f"{name}.{k}"] = v
        bits = planes[:, :, :, 2]
        print(f"  taps {name}: {planes[:, :, :, 0].numel()} footprints, all four taps inside {float((bits == 15).float().mean()):.3f}, "
              f"no column inside {float(((bits & 3) == 0).float().mean()):.3f}")
    save("tap_cases.npz", **out)


# ----------------------------------------------------------------------------
# 3. small end-to-end run with every seam recorded
# ----------------------------------------------------------------------------
def record_e2e(weights, tag, height=64, width=96, views=3, iteration=2, seed=11):
    model = build_reference(weights, iteration, test=True)
    ev, up = model.iter_mvs.evaluation, model.iter_mvs.update
    rec = {"corrnet_in": [], "pvw_in": [], "eval_out": [], "eval_samples": [], "logits": [],
           "update_in": [], "update_out": [], "hidden_init": [], "depth_init": []}
    hooks = []
    for i, net in enumerate(ev.corr_conv1):
        hooks.append(net.register_forward_pre_hook(lambda m, a, i=i: rec["corrnet_in"].append((i, a[0].clone()))))
    hooks.append(ev.pixel_view_weight.register_forward_pre_hook(lambda m, a: rec["pvw_in"].append(a[0].clone())))
    hooks.append(ev.register_forward_hook(lambda m, a, o: rec["eval_out"].append(o)))
    hooks.append(ev.register_forward_pre_hook(lambda m, a: rec["eval_samples"].append(a[4])))
    hooks.append(up.depth_head.register_forward_hook(lambda m, a, o: rec["logits"].append(o.clone())))
    hooks.append(up.register_forward_pre_hook(lambda m, a: rec["update_in"].append([x.clone() for x in a[:3]])))
    hooks.append(up.register_forward_hook(lambda m, a, o: rec["update_out"].append(o)))
    feats = {}
    hooks.append(model.feature_net.register_forward_hook(lambda m, a, o: feats.update(o)))
    upw = []
    hooks.append(model.iter_mvs.upsample.register_forward_hook(lambda m, a, o: upw.append(o.clone())))
    orig_hi, orig_di = up.hidden_init, up.depth_init
    up.hidden_init = lambda corr: rec["hidden_init"].append(orig_hi(corr)) or rec["hidden_init"][-1]
    up.depth_init = lambda hid: rec["depth_init"].append(orig_di(hid)) or rec["depth_init"][-1]

    sample = synthetic.make_sample(batch=1, num_views=views, height=height, width=width, seed=seed)
    with torch.no_grad():
        out = model(sample["imgs"], sample["proj_matrices"], sample["depth_min"], sample["depth_max"])
    for h in hooks:
        h.remove()

    arrays = {
        "imgs": npy(sample["imgs"]["level_0"]),
        "depth_min": npy(sample["depth_min"]), "depth_max": npy(sample["depth_max"]),
        "iteration": np.int64(iteration),
        "out.depths_upsampled": npy(out["depths_upsampled"]),
        "out.confidence_upsampled": npy(out["confidence_upsampled"]),
        "upsample_logits": npy(upw[0]),
    }
    for l in (1, 2, 3):
        arrays[f"proj.level_{l}"] = npy(sample["proj_matrices"][f"level_{l}"])
        arrays[f"feat.level{l}"] = np.stack([npy(f) for f in feats[f"level{l}"]], axis=1)  # [B,V,C,h,w]
    # Evaluation: call 0 = init branch, calls 1.. = iteration branch
    vw, score0, depth0 = rec["eval_out"][0]
    arrays.update({"init.samples": npy(rec["eval_samples"][0]), "init.view_weights": npy(vw),
                   "init.score": npy(score0), "init.depth": npy(depth0)})
    for s, x in enumerate(rec["pvw_in"]):
        arrays[f"init.corr_view{s}"] = npy(x)
    # CorrNet inputs: first the init call (index 2), then (0,1,2) per iteration
    arrays["init.agg"] = npy(rec["corrnet_in"][0][1])
    for it in range(iteration):
        for j in range(3):
            idx, x = rec["corrnet_in"][1 + it * 3 + j]
            assert idx == j
            arrays[f"iter{it}.agg.level{j + 1}"] = npy(x)
        for l in (1, 2, 3):
            arrays[f"iter{it}.samples.level{l}"] = npy(rec["eval_samples"][1 + it][f"level{l}"])
        arrays[f"iter{it}.score"] = npy(rec["eval_out"][1 + it])
        hid_in, nd_in, corr_in = rec["update_in"][it]
        hid, nd, prob, conf, conf0 = rec["update_out"][it]
        arrays.update({f"iter{it}.hidden_in": npy(hid_in), f"iter{it}.nd_in": npy(nd_in),
                       f"iter{it}.hidden": npy(hid), f"iter{it}.nd": npy(nd),
                       f"iter{it}.best": npy(torch.argmax(prob, dim=1, keepdim=True)),
                       f"iter{it}.logits": npy(rec["logits"][1 + it])})
        if conf is not None:
            arrays[f"iter{it}.conf"] = npy(conf)
    arrays["iter_last.prob"] = npy(rec["update_out"][-1][2])
    arrays["hidden0"] = npy(rec["hidden_init"][0])
    nd0, prob0 = rec["depth_init"][0]
    arrays.update({"nd0": npy(nd0), "best0": npy(torch.argmax(prob0, dim=1, keepdim=True)),
                   "logits0": npy(rec["logits"][0])})
    save(f"e2e_small_{tag}.npz", **arrays)


# ----------------------------------------------------------------------------
# 4. convex upsample seam (module.py:127)
# ----------------------------------------------------------------------------
def golden_upsample():
    gen = torch.Generator().manual_seed(5)
    x = torch.rand((2, 1, 6, 10), generator=gen)
    logits = torch.randn((2, 144, 6, 10), generator=gen)
    w = torch.softmax(logits.view(2, 1, 9, 4, 4, 6, 10), dim=2)
    y = ref_module.upsample(x, w)
    inv_min = torch.tensor([1 / 425.0, 1 / 300.0]).view(2, 1, 1, 1)
    inv_max = torch.tensor([1 / 935.0, 1 / 1200.0]).view(2, 1, 1, 1)
    d = ref_module.depth_unnormalization(y, inv_min, inv_max)
    nd = ref_module.depth_normalization(d, inv_min, inv_max)
    save("upsample.npz", x=npy(x), logits=npy(logits), up=npy(y), depth=npy(d), renorm=npy(nd),
         inv_min=npy(inv_min), inv_max=npy(inv_max))


# ----------------------------------------------------------------------------
# 5. BASELINE cfg-1 shape: seeded inputs are regenerated on the GPU box, only the
#    expected outputs (sub-sampled) are stored
# ----------------------------------------------------------------------------
def golden_cfg1(weights, tag, scene=False):
    model = build_reference(weights, 4, test=True)
    if scene:   # photo-consistent textured plane instead of noise images
        sample = synthetic.make_scene_sample(num_views=5, height=512, width=640, seed=0)
        tag += "_scene"
    else:
        sample = synthetic.make_sample(batch=1, num_views=5, height=512, width=640, seed=0)
    with torch.no_grad():
        out = model(sample["imgs"], sample["proj_matrices"], sample["depth_min"], sample["depth_max"])
    d, c = out["depths_upsampled"], out["confidence_upsampled"]
    save(f"cfg1_{tag}.npz", depth_sub=npy(d[:, :, ::4, ::4]), conf_sub=npy(c[:, :, ::4, ::4]),
         depth_sum=np.float64(d.double().sum().item()), conf_sum=np.float64(c.double().sum().item()),
         depth_row=npy(d[0, 0, 257]), conf_row=npy(c[0, 0, 257]))


# ----------------------------------------------------------------------------
# 6. training step: forward (train mode) + full_loss + backward
# ----------------------------------------------------------------------------
def golden_train(weights):
    gen = torch.Generator().manual_seed(21)
    b, v, h, w, iters = 2, 3, 64, 96, 2
    sample = synthetic.make_sample(batch=b, num_views=v, height=h, width=w, seed=4)
    gt0 = 500.0 + 300.0 * torch.rand((b, 1, h, w), generator=gen)
    gt2 = gt0[:, :, ::4, ::4].contiguous()
    m0 = (torch.rand((b, 1, h, w), generator=gen) > 0.2).float()
    m2 = m0[:, :, ::4, ::4].contiguous()
    arrays = {"imgs": npy(sample["imgs"]["level_0"]), "gt0": npy(gt0), "gt2": npy(gt2), "m0": npy(m0),
              "m2": npy(m2), "depth_min": npy(sample["depth_min"]), "depth_max": npy(sample["depth_max"]),
              "iteration": np.int64(iters)}
    for l in (1, 2, 3):
        arrays[f"proj.level_{l}"] = npy(sample["proj_matrices"][f"level_{l}"])
    for regress in (True, False):
        model = build_reference(weights, iters, test=False)
        out = model(sample["imgs"], sample["proj_matrices"], sample["depth_min"], sample["depth_max"])
        loss = ref_net.full_loss(out["depths"], out["depths_upsampled"], out["confidences"],
                                 {"level_0": gt0, "level_2": gt2}, {"level_0": m0, "level_2": m2},
                                 sample["depth_min"], sample["depth_max"], regress)
        loss.backward()
        tag = "regress" if regress else "noregress"
        arrays[f"{tag}.loss"] = np.float64(loss.item())
        record_gradients(model, arrays, tag)
        if regress:
            arrays["train.depths_upsampled"] = npy(out["depths_upsampled"][0])
            arrays["train.confidence_upsampled"] = npy(out["confidence_upsampled"])
            arrays["train.initial"] = npy(out["depths"]["initial"][0])
            for i, d in enumerate(out["depths"]["combine"]):
                arrays[f"train.combine{i}"] = npy(d)
            for i, cf in enumerate(out["confidences"]):
                arrays[f"train.conf{i}"] = npy(cf)
            arrays["train.best_last"] = npy(torch.argmax(out["depths"]["probability"][-1], 1, keepdim=True))
            arrays["train.running_mean_conv1"] = npy(model.feature_net.conv1.bn.running_mean)
        print(f"  train {tag}: loss {loss.item():.6f}")
    save("train_small.npz", **arrays)


# ----------------------------------------------------------------------------
# 6b. BASELINE cfg 4 at its real size: ONE training step (forward in train mode + full_loss + backward) at
#     B=1, V=5, 640x512, 4 iterations.  Inputs are regenerated from seeds on the GPU box; only the loss, the
#     per-parameter gradient norms and a few output statistics are stored.
# ----------------------------------------------------------------------------
def golden_train_cfg4(weights):
    sample, gt, mask = synthetic.make_training_sample(num_views=5, height=512, width=640, seed=2)
    arrays = {"iteration": np.int64(4)}
    for regress in (True, False):
        model = build_reference(weights, 4, test=False)
        out = model(sample["imgs"], sample["proj_matrices"], sample["depth_min"], sample["depth_max"])
        loss = ref_net.full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask,
                                 sample["depth_min"], sample["depth_max"], regress)
        loss.backward()
        tag = "regress" if regress else "noregress"
        arrays[f"{tag}.loss"] = np.float64(loss.item())
        record_gradients(model, arrays, tag)
        if regress:
            d = out["depths_upsampled"][0]
            arrays["train.depth_sub"] = npy(d[:, :, ::8, ::8])
            arrays["train.depth_abs_err_median"] = np.float64((d - gt["level_0"]).abs().median().item())
            arrays["train.initial_sub"] = npy(out["depths"]["initial"][0][:, :, ::4, ::4])
        print(f"  train cfg4 {tag}: loss {loss.item():.6f}")
    save("train_cfg4.npz", **arrays)


# ----------------------------------------------------------------------------
# 6c. The same step at B = 2 (two different scenes, 5 views, 640x512, 4 iterations): the batched form train_dtu.sh runs
#     (--batch_size 4 over the reference's DataParallel replicas = B >= 2 per forward), and the reference's `batch == 2`
#     branch of differentiable_warping (module.py:78-84) at full size.  Loss, gradient norms and gradient slices.
# ----------------------------------------------------------------------------
def golden_train_cfg4_b2(weights):
    imgs, projs, dmin, dmax, gt, mask = synthetic.make_training_batch(2, num_views=5, height=512, width=640, seed=2,
                                                                      hole_fraction=0.1)
    arrays = {"iteration": np.int64(4), "batch": np.int64(2)}
    for regress in (True, False):
        model = build_reference(weights, 4, test=False)
        out = model(imgs, projs, dmin, dmax)
        loss = ref_net.full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, regress)
        loss.backward()
        tag = "regress" if regress else "noregress"
        arrays[f"{tag}.loss"] = np.float64(loss.item())
        record_gradients(model, arrays, tag)
        if regress:
            d = out["depths_upsampled"][0]
            arrays["train.depth_sub"] = npy(d[:, :, ::8, ::8])
            arrays["train.depth_abs_err_median"] = np.float64((d - gt["level_0"]).abs().median().item())
            arrays["train.initial_sub"] = npy(out["depths"]["initial"][0][:, :, ::4, ::4])
        print(f"  train cfg4 B=2 {tag}: loss {loss.item():.6f}")
    save("train_cfg4_b2.npz", **arrays)


# ----------------------------------------------------------------------------
# 7. PFM bytes as written by the reference's datasets/data_io.py (run separately: this part needs only numpy)
# ----------------------------------------------------------------------------
def golden_pfm():
    import tempfile
    from datasets.data_io import read_pfm, save_pfm
    rng = np.random.RandomState(3)
    img = (rng.rand(5, 7) * 900).astype(np.float32)
    fd, path = tempfile.mkstemp(suffix=".pfm")
    os.close(fd)
    save_pfm(path, img)
    raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    back, scale = read_pfm(path)
    save("pfm.npz", image=img, file_bytes=raw, readback=np.ascontiguousarray(back), scale=np.float64(scale))


def main():
    """``make_golden.py`` rewrites every fixture; ``make_golden.py train_cfg4 pfm`` only the named ones"""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    w0, wd = golden_weights() if want("weights") else (synthetic.random_state_dict(0), None)
    if wd is None and (not only or only & {"e2e", "cfg1"}):
        wd = strip_module_prefix(torch.load(os.path.join(REF, "checkpoints/dtu/model_000015.ckpt"), map_location="cpu",
                                            weights_only=False)["model"])
    if want("warp"):
        golden_warp()
    if want("taps"):
        golden_taps()
    if want("upsample"):
        golden_upsample()
    if want("e2e"):
        record_e2e(w0, "seed0")
        record_e2e(wd, "dtu")
    if want("cfg1"):
        golden_cfg1(w0, "seed0")
        golden_cfg1(wd, "dtu")
        golden_cfg1(wd, "dtu", scene=True)
    if want("train"):
        golden_train(w0)
    if want("train_cfg4"):
        golden_train_cfg4(w0)
    if want("train_cfg4_b2"):
        golden_train_cfg4_b2(w0)
    if want("pfm"):
        golden_pfm()


if __name__ == "__main__":
    main()


