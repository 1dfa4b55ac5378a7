#!/usr/bin/env python3
"""Parity diagnostics on the GPU box (not collected by pytest).

For each BASELINE cfg-1 case it runs (a) the CPU oracle, (b) the SAME oracle code on PyTorch-ROCm
tensors (= what the reference's algorithm gives on this GPU with stock ops) and (c) the HIP
engine, and prints per-seam deviations and arg-max flip rates, so platform-induced chaos
(MIOpen vs MKL-DNN rounding) can be told from engine defects.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import golden, load_weights  # noqa: E402
from itermvs_amd import synthetic  # noqa: E402
from itermvs_amd.engine import InferenceEngine  # noqa: E402
from oracle import itermvs_oracle as O  # noqa: E402


def md(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


def rate(a, b, tol=1e-4):
    a, b = a.detach().cpu(), b.detach().cpu()
    rel = (a - b).abs() / b.abs().clamp(min=1e-12)
    return {"gt1e-4": float((rel > 1e-4).float().mean()), "gt1e-3": float((rel > 1e-3).float().mean()),
            "gt1e-2": float((rel > 1e-2).float().mean()), "max": float(rel.max()), "median": float(rel.median())}


def flips(a, b):
    return float((a.detach().cpu() != b.detach().cpu()).float().mean())


def run_case(tag):
    wtag = tag.split("_")[0]
    w = load_weights(wtag)
    if tag.endswith("scene"):
        s = synthetic.make_scene_sample(num_views=5, height=512, width=640, seed=0)
    else:
        s = synthetic.make_sample(batch=1, num_views=5, height=512, width=640, seed=0)
    t_cpu, t_gpu, t_eng = {}, {}, {}
    with torch.no_grad():
        out_cpu = O.pipeline_forward(w, s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"], 4, trace=t_cpu)
        wg = {k: v.cuda() for k, v in w.items()}
        imgs = {k: v.cuda() for k, v in s["imgs"].items()}
        pm = {k: v.cuda() for k, v in s["proj_matrices"].items()}
        dmin, dmax = s["depth_min"].cuda(), s["depth_max"].cuda()
        # oracle on the GPU: projection composed on the CPU like the CPU oracle (isolates conv/elementwise backends)
        out_gpu = O.pipeline_forward(wg, imgs, pm, dmin, dmax, 4, trace=t_gpu)
        eng = InferenceEngine(wg, 4)
        d_eng, c_eng = eng.run(imgs["level_0"], {l: pm[f"level_{l}"] for l in (1, 2, 3)}, dmin, dmax, trace=t_eng)
    g = golden(f"cfg1_{tag}.npz")
    rep = {"case": tag}
    rep["oracle_cpu_vs_reference_golden"] = rate(out_cpu["depths_upsampled"][:, :, ::4, ::4], g["depth_sub"])
    rep["oracle_gpu_vs_oracle_cpu.depth"] = rate(out_gpu["depths_upsampled"], out_cpu["depths_upsampled"])
    rep["engine_vs_oracle_cpu.depth"] = rate(d_eng, out_cpu["depths_upsampled"])
    rep["engine_vs_oracle_gpu.depth"] = rate(d_eng, out_gpu["depths_upsampled"])
    rep["engine_vs_oracle_cpu.conf_maxabs"] = md(c_eng, out_cpu["confidence_upsampled"])
    seams = {}
    for l in (1, 2, 3):
        seams[f"feat{l}.eng_cpu"] = md(t_eng["feats"][l], t_cpu["feats"][l])
        seams[f"feat{l}.gpu_cpu"] = md(t_gpu["feats"][l], t_cpu["feats"][l])
    seams["init_agg.eng_cpu"] = md(t_eng["init_agg"].permute(0, 2, 1, 3, 4), t_cpu["init_agg"])
    seams["init_agg.gpu_cpu"] = md(t_gpu["init_agg"], t_cpu["init_agg"])
    seams["view_w.eng_cpu"] = md(t_eng["view_weights"], t_cpu["view_weights"])
    seams["view_w.gpu_cpu"] = md(t_gpu["view_weights"], t_cpu["view_weights"])
    seams["score0.eng_cpu"] = md(t_eng["init_score"], t_cpu["init_score"])
    seams["score0.gpu_cpu"] = md(t_gpu["init_score"], t_cpu["init_score"])
    seams["hidden0.eng_cpu"] = md(t_eng["hidden0"], t_cpu["hidden0"])
    seams["hidden0.gpu_cpu"] = md(t_gpu["hidden0"], t_cpu["hidden0"])
    seams["best0.flips.eng_cpu"] = flips(t_eng["best0"], t_cpu["best0"])
    seams["best0.flips.gpu_cpu"] = flips(t_gpu["best0"], t_cpu["best0"])
    seams["best0.flips.eng_gpu"] = flips(t_eng["best0"], t_gpu["best0"])
    for it in range(4):
        e, c, gq = t_eng["iters"][it], t_cpu["iters"][it], t_gpu["iters"][it]
        seams[f"it{it}.nd_in.eng_cpu.gt1e-4"] = float(((e["nd_in"].cpu() - c["nd_in"]).abs() > 1e-4).float().mean())
        seams[f"it{it}.nd_in.gpu_cpu.gt1e-4"] = float(((gq["nd_in"].cpu() - c["nd_in"]).abs() > 1e-4).float().mean())
        for i in range(3):
            seams[f"it{it}.agg{i + 1}.eng_cpu"] = md(e["aggs"][i].permute(0, 2, 1, 3, 4), c["aggs"][i])
        seams[f"it{it}.score.eng_cpu"] = md(e["score"], c["score"])
        seams[f"it{it}.hidden.eng_cpu"] = md(e["hidden"], c["hidden"])
        seams[f"it{it}.hidden.gpu_cpu"] = md(gq["hidden"], c["hidden"])
        seams[f"it{it}.best.flips.eng_cpu"] = flips(e["best"], c["best"])
        seams[f"it{it}.best.flips.gpu_cpu"] = flips(gq["best"], c["best"])
        seams[f"it{it}.best.flips.eng_gpu"] = flips(e["best"], gq["best"])
        # how often is the flip a +-1 neighbour?
        dd = (e["best"].cpu() - c["best"]).abs()
        seams[f"it{it}.best.flip_gt1bin.eng_cpu"] = float((dd > 1).float().mean())
    rep["seams"] = seams
    if tag.endswith("scene"):
        rep["abs_err_mm_median.engine"] = float((d_eng.cpu() - s["depth_gt"]).abs().median())
        rep["abs_err_mm_median.oracle_cpu"] = float((out_cpu["depths_upsampled"] - s["depth_gt"]).abs().median())
    return rep


if __name__ == "__main__":
    for tag in sys.argv[1:] or ["dtu_scene", "seed0", "dtu"]:
        print(json.dumps(run_case(tag), indent=1), flush=True)
