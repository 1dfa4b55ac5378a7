"""End-to-end ``Pipeline(test=True)`` on the MI355X vs golden outputs of the reference."""
import pytest
import torch

from conftest import golden, load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_model(tag, iteration):
    from itermvs_amd.net import Pipeline
    m = Pipeline(iteration=iteration, test=True)
    # checkpoints carry the DataParallel 'module.' prefix (eval.py:119,125)
    m.load_checkpoint_state({"module." + k: v for k, v in load_weights(tag).items()})
    return m.to(DEV).eval()


def to_dev(sample):
    return ({k: v.to(DEV) for k, v in sample["imgs"].items()}, {k: v.to(DEV) for k, v in sample["proj_matrices"].items()},
            sample["depth_min"].to(DEV), sample["depth_max"].to(DEV))


def maxdiff(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


@pytest.mark.parametrize("tag", ["seed0", "dtu"])
def test_small_pipeline_every_seam(tag):
    """V=3, 64x96, 2 iterations: FeatureNet, init correlation, view weights, scores, hidden state,
    arg-max indices and the final maps against what the reference produced at the same seams."""
    g = golden(f"e2e_small_{tag}.npz")
    iters = int(g.np("iteration"))
    model = make_model(tag, iters)
    from itermvs_amd.engine import InferenceEngine
    eng = InferenceEngine(model.weights(), iters)
    imgs = g["imgs"].to(DEV)
    projs = {l: g[f"proj.level_{l}"].to(DEV) for l in (1, 2, 3)}
    trace = {}
    with torch.no_grad():
        depth, conf = eng.run(imgs, projs, g["depth_min"].to(DEV), g["depth_max"].to(DEV), trace=trace)
    rel = lambda got, ref: maxdiff(got, ref) / max(1.0, float(ref.abs().max()))
    m = {}
    for l in (1, 2, 3):
        m[f"feat{l}"] = rel(trace["feats"][l].view(g[f"feat.level{l}"].shape), g[f"feat.level{l}"])
    s = trace["corr_views"].shape[1]
    m["corr_view"] = max(rel(trace["corr_views"][:, i].permute(0, 2, 1, 3, 4), g[f"init.corr_view{i}"]) for i in range(s))
    m["view_w"] = maxdiff(trace["view_weights"], g["init.view_weights"])
    m["init_agg"] = rel(trace["init_agg"], g["init.agg"].permute(0, 2, 1, 3, 4))
    m["init_score"] = rel(trace["init_score"], g["init.score"])
    m["hidden0"] = maxdiff(trace["hidden0"], g["hidden0"])
    flips0 = float((trace["best0"].cpu() != g["best0"]).float().mean())
    flips = float((trace["iters"][-1]["best"].cpu() != g[f"iter{iters - 1}.best"]).float().mean())
    d, dr = depth.cpu(), g["out.depths_upsampled"]
    drel = (d - dr).abs() / dr.abs()
    # confidence: the sigmoid of the last iteration and its x4 up-sampling, wherever the depth agrees (an arg-max flip
    # upstream moves the hidden state of that pixel's neighbourhood and the confidence with it)
    c_lo, c_lo_ref = trace["iters"][-1]["conf"].cpu(), g[f"iter{iters - 1}.conf"]
    cu_, cr = conf.cpu(), g["out.confidence_upsampled"]
    m["conf_lo_median"] = float((c_lo - c_lo_ref).abs().median())
    m["conf_up_median"] = float((cu_ - cr).abs().median())
    conf_bad = float(((cu_ - cr).abs() > 1e-3).float().mean())
    print(f"e2e_small {tag}: " + ", ".join(f"{k}={v:.1e}" for k, v in m.items()) +
          f", flips0={flips0:.4f}, flips_last={flips:.4f}, depth median {float(drel.median()):.1e}, "
          f"depth>1e-4 {float((drel > 1e-4).float().mean()):.4f}, conf>1e-3 {conf_bad:.4f}")
    # un-forced run: each seam carries the deviations of the seams before it.  Bounds = ~5x what the kernels measure on
    # the MI355X (profiles/r02/r02b_parity_report.txt: features 1.4e-6, correlations 3.5e-6 / 5.2e-6, view weights 5.6e-6,
    # scores 4.0e-6, hidden state 1.2e-5 (seed0) / 9.4e-5 (DTU weights, tanh of larger pre-activations))
    for l in (1, 2, 3):
        assert m[f"feat{l}"] <= 1e-5
    assert m["corr_view"] <= 2e-5 and m["init_agg"] <= 3e-5
    assert m["view_w"] <= 5e-5 and m["init_score"] <= 3e-5 and m["hidden0"] <= (1e-4 if tag == "seed0" else 5e-4)
    lim = 0.02 if tag == "seed0" else 0.10
    assert flips0 <= lim, flips0
    assert flips <= 2 * lim, flips
    assert float(drel.median()) <= 1e-5
    assert float((drel > 1e-4).float().mean()) <= 2 * lim, float(drel.max())
    assert conf.shape == cr.shape and c_lo.shape == c_lo_ref.shape
    assert m["conf_lo_median"] <= 2e-5 and m["conf_up_median"] <= 2e-5 and conf_bad <= 2 * lim


def _rates(a, b):
    a, b = a.detach().cpu(), b.detach().cpu()
    rel = (a - b).abs() / b.abs()
    return float((rel > 1e-4).float().mean()), float(rel.median()), float(rel.max())


@pytest.mark.parametrize("tag", ["seed0", "dtu_scene", "dtu"])
def test_cfg1_against_reference_outputs(tag):
    """BASELINE cfg 1/2: V=5, 640x512, 4 iterations, seeded inputs regenerated here.

    End-to-end the network is chaotic: one arg-max flip (from a 1e-7 rounding difference between
    two conv back-ends) moves a pixel by >10 % and spreads through the GRU.  Measured: the
    reference's own CPU path on two different hosts (the golden comes from an 8-core Xeon, the GPU
    box has an EPYC) disagrees on 3 % of the pixels of the photo-consistent ``dtu_scene`` case and
    on 37 % with noise images.  So the gate is threefold:
      (1) before chaos sets in (first arg-max) the engine agrees with the CPU oracle;
      (2) the engine's mismatch rate against the CPU oracle / the reference golden is no worse
          than the platform floor = the SAME oracle code run with stock PyTorch-ROCm ops here;
      (3) with benign weights (seed0) engine and PyTorch-ROCm oracle agree at EVERY pixel.
    Kernel-level parity on identical inputs is asserted in test_kernels_gpu.py."""
    from itermvs_amd import synthetic
    from itermvs_amd.engine import InferenceEngine
    from oracle import itermvs_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g = golden(f"cfg1_{tag}.npz")
    wtag = tag.split("_")[0]
    w = load_weights(wtag)
    if tag.endswith("scene"):
        s = synthetic.make_scene_sample(num_views=5, height=512, width=640, seed=0)
    else:
        s = synthetic.make_sample(batch=1, num_views=5, height=512, width=640, seed=0)
    model = make_model(wtag, 4)
    out = model(*to_dev(s))                                                         # public API (net.py:78)
    assert set(out.keys()) == {"depths_upsampled", "confidence_upsampled"}          # net.py:125-128
    d, c = out["depths_upsampled"], out["confidence_upsampled"]
    assert d.shape == (1, 1, 512, 640) and c.shape == (1, 1, 512, 640)

    t_cpu, t_gpu, t_eng = {}, {}, {}
    imgs, pm, dmin, dmax = to_dev(s)
    with torch.no_grad():
        o_cpu = O.pipeline_forward(w, s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"], 4, trace=t_cpu)
        o_gpu = O.pipeline_forward({k: v.to(DEV) for k, v in w.items()}, imgs, pm, dmin, dmax, 4, trace=t_gpu)
        eng = InferenceEngine(model.weights(), 4)
        pj = {l: pm[f"level_{l}"] for l in (1, 2, 3)}
        d3, _ = eng.run(imgs["level_0"], pj, dmin, dmax)
        d2, _ = eng.run(imgs["level_0"], pj, dmin, dmax, trace=t_eng)
    assert maxdiff(d3, d) == 0.0                                                    # deterministic engine

    # (1) first arg-max, before any feedback
    flips0 = float((t_eng["best0"].cpu() != t_cpu["best0"]).float().mean())
    floor0 = float((t_gpu["best0"].cpu() != t_cpu["best0"]).float().mean())
    for l in (1, 2, 3):
        ref = t_cpu["feats"][l]
        assert maxdiff(t_eng["feats"][l], ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert flips0 <= max(2e-3, 2 * floor0), (flips0, floor0)

    # (2) end to end against the platform floor
    floor, _, _ = _rates(o_gpu["depths_upsampled"], o_cpu["depths_upsampled"])
    host_floor, _, _ = _rates(o_cpu["depths_upsampled"][:, :, ::4, ::4], g["depth_sub"])
    bad_cpu, med_cpu, max_cpu = _rates(d, o_cpu["depths_upsampled"])
    bad_gold, med_gold, _ = _rates(d[:, :, ::4, ::4], g["depth_sub"])
    bad_gpu, med_gpu, max_gpu = _rates(d, o_gpu["depths_upsampled"])
    print(f"cfg1 {tag}: engine-vs-oracleCPU {bad_cpu:.4f} (median {med_cpu:.1e}) | engine-vs-golden {bad_gold:.4f} | "
          f"engine-vs-oracleROCm {bad_gpu:.4f} (max {max_gpu:.1e}) | floors: ROCm-vs-CPU {floor:.4f}, "
          f"CPU(EPYC)-vs-golden(Xeon) {host_floor:.4f} | first-argmax flips {flips0:.5f} (floor {floor0:.5f})")
    med_lim = 1e-4 if tag == "dtu" else 2e-6
    assert med_cpu <= med_lim and med_gold <= med_lim
    assert bad_cpu <= 1.5 * max(floor, host_floor) + 0.005
    assert bad_gold <= 1.5 * max(floor, host_floor) + 0.005
    # the traced run evaluates the depth head unfused (logits in memory, other summation order): same gate
    bad_tr, med_tr, _ = _rates(d2, d)
    assert med_tr <= med_lim and bad_tr <= 1.5 * max(floor, host_floor) + 0.005, (bad_tr, med_tr)
    cbad = float(((c.cpu() - o_cpu["confidence_upsampled"]).abs() > 1e-3).float().mean())
    assert cbad <= 1.5 * max(floor, host_floor) + 0.01

    # (3) same platform, benign weights, same convolution back-end (MIOpen): every pixel
    if tag == "seed0":
        from miopen_engine import MiopenEngine
        eng_m = MiopenEngine(model.weights(), 4)
        with torch.no_grad():
            d_m, _ = eng_m.run(imgs["level_0"], {l: pm[f"level_{l}"] for l in (1, 2, 3)}, dmin, dmax)
        bad_m, _, max_m = _rates(d_m, o_gpu["depths_upsampled"])
        print(f"  MIOpen-backed engine vs PyTorch-ROCm oracle: mismatch {bad_m:.5f}, max rel {max_m:.2e}")
        assert bad_m == 0.0 and max_m <= 1e-5
        assert bad_gpu <= 1.5 * max(floor, host_floor) + 0.005
    if tag == "dtu_scene":   # and the engine really reconstructs the plane (mm)
        assert float((d.cpu() - s["depth_gt"]).abs().median()) < 1.0
        assert abs(float((d.cpu() - s["depth_gt"]).abs().median())
                   - float((o_cpu["depths_upsampled"] - s["depth_gt"]).abs().median())) < 0.02


def test_batch_of_two_equals_two_singles():
    from itermvs_amd import synthetic
    model = make_model("seed0", 2)
    s2 = synthetic.make_sample(batch=2, num_views=3, height=64, width=96, seed=2)
    out2 = model(*to_dev(s2))
    for b in range(2):
        s1 = {"imgs": {k: v[b:b + 1] for k, v in s2["imgs"].items()},
              "proj_matrices": {k: v[b:b + 1] for k, v in s2["proj_matrices"].items()},
              "depth_min": s2["depth_min"][b:b + 1], "depth_max": s2["depth_max"][b:b + 1]}
        out1 = model(*to_dev(s1))
        rel = (out2["depths_upsampled"][b] - out1["depths_upsampled"][0]).abs() / out1["depths_upsampled"][0]
        assert float((rel > 1e-4).float().mean()) <= 0.02
        assert float(rel.median()) <= 1e-6


def test_views_and_iterations_are_runtime_parameters():
    from itermvs_amd import synthetic
    for views, iters in ((2, 1), (7, 3), (11, 2)):
        model = make_model("seed0", iters)
        s = synthetic.make_sample(batch=1, num_views=views, height=64, width=96, seed=views)
        out = model(*to_dev(s))
        assert out["depths_upsampled"].shape == (1, 1, 64, 96)
        assert bool(torch.isfinite(out["depths_upsampled"]).all())
        d = out["depths_upsampled"]
        assert float(d.min()) >= 424.9 and float(d.max()) <= 935.1


def test_rejects_cpu_tensors_and_bad_sizes():
    from itermvs_amd import synthetic
    model = make_model("seed0", 1)
    s = synthetic.make_sample(batch=1, num_views=3, height=64, width=96, seed=0)
    with pytest.raises(RuntimeError, match="MI355X"):
        model(s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"])
    bad = {k: v[..., :60, :] for k, v in s["imgs"].items()}
    with pytest.raises(RuntimeError, match="multiples of 32"):
        model({k: v.to(DEV) for k, v in bad.items()}, {k: v.to(DEV) for k, v in s["proj_matrices"].items()},
              s["depth_min"].to(DEV), s["depth_max"].to(DEV))


def test_nan_projection_raises_like_the_reference_and_clears():
    """module.py:83,87: a singular reference camera trips the (deferred) NaN assert; the next good sample runs"""
    from itermvs_amd import synthetic
    model = make_model("seed0", 1)
    s = synthetic.make_sample(batch=1, num_views=3, height=64, width=96, seed=0)
    imgs, pm, dmin, dmax = to_dev(s)
    bad = {k: v.clone() for k, v in pm.items()}
    for k in bad:
        bad[k][:, 0] = 0.0
    with pytest.raises(AssertionError, match="nan in proj"):
        model(imgs, bad, dmin, dmax)
    out = model(imgs, pm, dmin, dmax)                        # the flag was cleared by the failed check
    assert bool(torch.isfinite(out["depths_upsampled"]).all())
    graphed = make_model("seed0", 1)
    graphed.use_graphs = True                                # graph mode: never stalls, the check is explicit
    graphed(imgs, pm, dmin, dmax)
    graphed(imgs, bad, dmin, dmax)
    with pytest.raises(AssertionError, match="nan in proj"):
        graphed.check_projection_finite()
    graphed.check_projection_finite()


@pytest.mark.parametrize("fdt", ["bf16", "fp16"])
def test_16bit_feature_storage_end_to_end(fdt):
    """BASELINE cfg 4 (bf16) / cfg 5 (fp16) feature storage at the cfg-1 shape, photo-consistent scene, DTU weights:
    the stated tolerance against the fp32 engine.  Storage rounding perturbs the features by 2^-9 (bf16) / 2^-12 (fp16)
    relative, the correlations by about as much, and a perturbation of that size flips arg-max bins on part of the
    pixels (the chaos of DESIGN.md section 2); what must hold: same reconstruction quality, median deviation small."""
    from itermvs_amd import synthetic
    s = synthetic.make_scene_sample(num_views=5, height=512, width=640, seed=0)
    ref_model = make_model("dtu", 4)
    m16 = make_model("dtu", 4)
    m16.feature_dtype = fdt
    d32 = ref_model(*to_dev(s))["depths_upsampled"]
    out = m16(*to_dev(s))
    d16, c16 = out["depths_upsampled"], out["confidence_upsampled"]
    assert d16.dtype == torch.float32 and c16.dtype == torch.float32
    assert m16._engine.feature_dtype == {"bf16": torch.bfloat16, "fp16": torch.float16}[fdt]
    rel = ((d16 - d32).abs() / d32).cpu()
    gt = s["depth_gt"]
    e32, e16 = float((d32.cpu() - gt).abs().median()), float((d16.cpu() - gt).abs().median())
    print(f"feature storage {fdt}: depth vs fp32 engine median {float(rel.median()):.2e}, >1e-3 on {float((rel > 1e-3).float().mean()):.4f} "
          f"of the pixels; median |depth - ground truth| {e16:.3f} mm (fp32: {e32:.3f} mm)")
    # measured on the MI355X (profiles/r02): median 4.3e-6 (bf16) / 5.4e-7 (fp16); 3.5 % / 2.1 % of the pixels move by more
    # than 1e-3 (arg-max flips); plane reconstruction 0.743 / 0.741 mm vs 0.742 mm in fp32
    assert float(rel.median()) <= (2e-5 if fdt == "bf16" else 5e-6)
    assert float((rel > 1e-3).float().mean()) <= 0.08
    assert float((rel > 1e-2).float().mean()) <= 0.05
    assert abs(e16 - e32) <= 0.1 and e16 < 1.0               # reconstructs the plane as well as fp32 does


def test_graph_replay_equals_eager():
    """hipGraph segments + eager corr_iter launches reproduce the eager engine bit for bit, for changing inputs."""
    from itermvs_amd import synthetic
    model = make_model("seed0", 3)
    graphed = make_model("seed0", 3)
    graphed.use_graphs = True
    for seed in (1, 2, 3):
        s = synthetic.make_sample(batch=1, num_views=4, height=64, width=96, seed=seed)
        a = model(*to_dev(s))
        b = graphed(*to_dev(s))
        torch.cuda.synchronize()
        assert torch.equal(a["depths_upsampled"], b["depths_upsampled"])
        assert torch.equal(a["confidence_upsampled"], b["confidence_upsampled"])


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


@pytest.mark.parametrize("cfg,fdt", [("cfg3", "fp32"), ("cfg3v6", "fp32"), ("cfg5", "fp32"), ("cfg5", "fp16")])
def test_full_size_configs_cross_backend(cfg, fdt):
    """BASELINE cfg 3 (1600x1152, 5 views, 4 iterations) and cfg 5 (1920x1280, 11 views, 8 iterations):
    too large for the CPU oracle in a test, so parity is checked through size-independent properties --
    determinism, the two independent convolution back-ends (hand-written MFMA kernels vs MIOpen) agreeing
    to the platform's chaos floor, first arg-max identical, outputs inside the depth range.
    cfg 5 also runs AS STATED in BASELINE.json: fp16 feature storage at the full 1920x1280 / 11 views / 8 iterations.
    With 16-bit storage the two back-ends' features (1e-6 apart in fp32) round to different fp16 values on ~0.2 % of
    the elements, so the chaos floor between them is higher than in fp32 -- the bounds below are the measured ones."""
    from itermvs_amd import synthetic
    from itermvs_amd.engine import InferenceEngine
    # cfg 3 is quoted with "5 src views" in BASELINE.json (6 images) while the reference's DTU script runs 5 images: both
    views, h, w, iters = {"cfg3": (5, 1152, 1600, 4), "cfg3v6": (6, 1152, 1600, 4), "cfg5": (11, 1280, 1920, 8)}[cfg]
    model = make_model("seed0", iters)
    s = synthetic.make_sample(batch=1, num_views=views, height=h, width=w, seed=3)
    imgs, pm, dmin, dmax = to_dev(s)
    projs = {l: pm[f"level_{l}"] for l in (1, 2, 3)}
    from miopen_engine import MiopenEngine
    hip = InferenceEngine(model.weights(), iters, fdt)
    mio = MiopenEngine(model.weights(), iters, fdt)
    t_hip, t_mio = {}, {}
    with torch.no_grad():
        d1, c1 = hip.run(imgs["level_0"], projs, dmin, dmax, trace=t_hip)
        d1b, _ = hip.run(imgs["level_0"], projs, dmin, dmax)                    # fused depth head (no trace)
        d1c = hip.run(imgs["level_0"], projs, dmin, dmax)[0].clone()
        d2, c2 = mio.run(imgs["level_0"], projs, dmin, dmax, trace=t_mio)
    assert d1.shape == (1, 1, h, w) and c1.shape == (1, 1, h, w)
    assert torch.equal(d1b, d1c)                                                 # deterministic
    relf = (d1 - d1b).abs() / d1b                                                # traced (unfused head) vs fused
    assert float(relf.median()) <= 1e-6 and float((relf > 1e-4).float().mean()) <= 0.03
    assert bool(torch.isfinite(d1).all()) and float(d1.min()) >= 424.9 and float(d1.max()) <= 935.1
    flips0 = float((t_hip["best0"] != t_mio["best0"]).float().mean())
    rel = (d1 - d2).abs() / d2
    bad = float((rel > 1e-4).float().mean())
    print(f"{cfg} {fdt}: first-argmax flips {flips0:.6f}; depth mismatch HIP-convs vs MIOpen {bad:.5f}, median {float(rel.median()):.1e}")
    if fdt == "fp32":
        assert flips0 <= 1e-3 and float(rel.median()) <= 1e-6 and bad <= 0.03
    else:
        assert hip.feature_net(imgs["level_0"][0, :1].contiguous())[1].dtype == torch.float16
        assert flips0 <= 5e-3 and float(rel.median()) <= 1e-5 and bad <= 0.15


def test_in_place_weight_updates_refresh_the_packed_engine():
    """test-mode Pipeline folds / packs its weights and captures hipGraphs once; an autograd-visible in-place update of a
    parameter (what an optimizer step does) must be noticed at the next forward (net.py: version counters), in eager and
    in graph mode; two graph runners of one engine own separate workspaces"""
    from itermvs_amd import synthetic
    from itermvs_amd.engine import GraphedRunner
    s = synthetic.make_sample(batch=1, num_views=3, height=64, width=96, seed=5)
    for graphs in (False, True):
        m = make_model("seed0", 2)
        m.use_graphs = graphs
        a = m(*to_dev(s))["depths_upsampled"].clone()
        eng = m._engine
        with torch.no_grad():
            m.get_parameter("iter_mvs.update.depth_head.4.bias").add_(torch.linspace(-1, 1, 256, device=DEV))
        b = m(*to_dev(s))["depths_upsampled"].clone()
        assert m._engine is not eng and not torch.equal(a, b)
        fresh = make_model("seed0", 2)
        fresh.load_state_dict(m.state_dict())
        fresh = fresh.to(DEV).eval()
        assert torch.equal(fresh(*to_dev(s))["depths_upsampled"], b)
        assert m(*to_dev(s))["depths_upsampled"].data_ptr() and m._engine is not eng
    imgs, pm, dmin, dmax = to_dev(s)
    pj = {l: pm[f"level_{l}"] for l in (1, 2, 3)}
    eng = make_model("seed0", 2)
    eng(imgs, pm, dmin, dmax)
    r1 = GraphedRunner(eng._engine, imgs["level_0"], pj, dmin, dmax)
    r2 = GraphedRunner(eng._engine, imgs["level_0"], pj, dmin, dmax)
    keys = [k for k in eng._engine._ws if k[3] is not None]
    assert len(keys) == 2 and eng._engine._ws[keys[0]]["hx"].data_ptr() != eng._engine._ws[keys[1]]["hx"].data_ptr()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        o1 = r1(imgs["level_0"], pj, dmin, dmax)
    with torch.cuda.stream(s2):
        o2 = r2(imgs["level_0"], pj, dmin, dmax)
    torch.cuda.synchronize()
    assert torch.equal(o1[0], o2[0]) and torch.equal(o1[1], o2[1])


def test_host_fp32_projection_mode_reproduces_the_reference_composition():
    """``projection="host_fp32"``: the engine composes ``src @ inverse(ref)`` on the host in fp32 like module.py:77-90 --
    bit for bit the pinned oracle's restatement of those lines on this host, so the kernels see the reference's own
    matrices (tap indices identical to a reference run here); the default device composition stays within the tap-index
    bounds of test_compose_proj_tap_indices.  Outputs of the two modes agree to the chaos floor; graph capture refuses
    the mode (it reads the cameras back)."""
    from itermvs_amd import synthetic
    from oracle import itermvs_oracle as O
    s = synthetic.make_sample(batch=2, num_views=4, height=64, width=96, seed=7)
    imgs, pm, dmin, dmax = to_dev(s)
    m_host = make_model("seed0", 2)
    m_host.projection = "host_fp32"
    m_dev = make_model("seed0", 2)
    t_host, t_dev = {}, {}
    projs = {l: pm[f"level_{l}"] for l in (1, 2, 3)}
    from itermvs_amd.engine import InferenceEngine
    e_host = InferenceEngine(m_host.weights(), 2, projection="host_fp32")
    e_dev = InferenceEngine(m_dev.weights(), 2)
    with torch.no_grad():
        d_h, _ = e_host.run(imgs["level_0"], projs, dmin, dmax, trace=t_host)
        d_d, _ = e_dev.run(imgs["level_0"], projs, dmin, dmax, trace=t_dev)
    for i, l in enumerate((1, 2, 3)):
        mats = s["proj_matrices"][f"level_{l}"].float()
        for k in range(1, mats.shape[1]):
            want = O.compose_projection(mats[:, k], mats[:, 0])[:, :3, :4].reshape(-1, 12)
            assert torch.equal(t_host["proj"][i][:, k - 1].cpu(), want)                     # the reference's own numbers
            assert maxdiff(t_dev["proj"][i][:, k - 1], want) <= 2e-5 * float(want.abs().max())
    rel = ((d_h - d_d).abs() / d_d).cpu()
    assert float(rel.median()) <= 1e-5 and float((rel > 1e-4).float().mean()) <= 0.02
    with torch.no_grad():
        d_h_fused, _ = e_host.run(imgs["level_0"], projs, dmin, dmax)          # (the traced run evaluates the head layer by layer)
    out = m_host(imgs, pm, dmin, dmax)["depths_upsampled"]
    assert torch.equal(out, d_h_fused)
    m_host.use_graphs = True
    with pytest.raises(RuntimeError, match="host_fp32"):
        m_host(imgs, pm, dmin, dmax)


def test_host_fp32_projection_is_capturable_with_host_resident_cameras():
    """``projection="host_fp32"`` with the cameras handed over as CPU tensors (where a data loader has them): composed on the
    host in fp32 like module.py:77-90 WITHOUT any device synchronisation, and capturable -- the hipGraph reads the composed
    matrices from a static buffer every replay refreshes through pinned memory.  Eager == graph replay bit for bit, both ==
    the synchronising device-camera form, and replays follow changing cameras."""
    from itermvs_amd import synthetic
    from itermvs_amd.engine import InferenceEngine
    from oracle import itermvs_oracle as O
    s = [synthetic.make_sample(batch=1, num_views=4, height=64, width=96, seed=k) for k in (3, 4, 5)]
    want = []
    m_ref = make_model("seed0", 2)
    m_ref.projection = "host_fp32"
    for smp in s:                                           # the eager, synchronising form with device cameras (round 3)
        imgs, pm, dmin, dmax = to_dev(smp)
        want.append(m_ref(imgs, pm, dmin, dmax)["depths_upsampled"].clone())
    pm0 = torch.stack([s[0]["proj_matrices"][f"level_{l}"].float() for l in (1, 2, 3)])
    comp = InferenceEngine.compose_host(pm0)
    for i in range(3):
        for k in range(1, 4):
            assert torch.equal(comp[i][:, k - 1], O.compose_projection(pm0[i][:, k], pm0[i][:, 0])[:, :3, :4].reshape(-1, 12))
    for graphs in (False, True):
        m = make_model("seed0", 2)
        m.projection = "host_fp32"
        m.use_graphs = graphs
        for rep in range(2):                                # second pass: replays of the captured graph with other cameras
            for smp, w in zip(s, want):
                imgs, _, dmin, dmax = to_dev(smp)
                got = m(imgs, smp["proj_matrices"], dmin, dmax)["depths_upsampled"]         # CPU cameras
                assert torch.equal(got, w), (graphs, rep)
        if graphs:
            assert len(m._runners) == 1
    bad = {k: v.clone() for k, v in s[0]["proj_matrices"].items()}
    bad["level_2"][0, 1, 0, 0] = float("nan")               # non-finite source camera: module.py:83,87 assert, raised on the spot
    imgs, _, dmin, dmax = to_dev(s[0])
    with pytest.raises(AssertionError, match="nan in proj"):
        m(imgs, bad, dmin, dmax)
    bad["level_2"][0, 1, 0, 0] = s[0]["proj_matrices"]["level_2"][0, 1, 0, 0]
    bad["level_2"][0, 0] = 0.0                              # singular reference camera: torch.inverse raises on the host, as in the
    with pytest.raises(RuntimeError):                       # reference run on a CPU (module.py:81)
        m(imgs, bad, dmin, dmax)
    m_dev = make_model("seed0", 2)                          # default composition: CPU cameras are simply uploaded
    imgs, pm, dmin, dmax = to_dev(s[1])
    assert torch.equal(m_dev(imgs, s[1]["proj_matrices"], dmin, dmax)["depths_upsampled"], m_dev(imgs, pm, dmin, dmax)["depths_upsampled"])
