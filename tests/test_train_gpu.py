"""Training form of the pipeline on the MI355X (HIP warp forward/backward inside autograd) vs
the reference's forward + full_loss + backward captured in tests/golden/train_small.npz."""
import pytest
import torch

from conftest import golden, load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("regress", [True, False])
def test_train_step_matches_reference(regress):
    from itermvs_amd.net import Pipeline, full_loss
    g = golden("train_small.npz")
    tag = "regress" if regress else "noregress"
    model = Pipeline(iteration=int(g.np("iteration")), test=False)
    model.load_state_dict(load_weights("seed0"))
    model = model.to(DEV).train()
    imgs = {"level_0": g["imgs"].to(DEV)}
    proj = {f"level_{l}": g[f"proj.level_{l}"].to(DEV) for l in (1, 2, 3)}
    dmin, dmax = g["depth_min"].to(DEV), g["depth_max"].to(DEV)
    out = model(imgs, proj, dmin, dmax)
    assert set(out.keys()) == {"depths", "depths_upsampled", "confidences", "confidence_upsampled"}   # net.py:115-120
    n = int(g.np("iteration")) + 1
    assert len(out["depths"]["combine"]) == n and len(out["depths"]["probability"]) == n
    assert len(out["depths"]["initial"]) == 1 and len(out["confidences"]) == n and len(out["depths_upsampled"]) == 1
    gt = {"level_0": g["gt0"].to(DEV), "level_2": g["gt2"].to(DEV)}
    mk = {"level_0": g["m0"].to(DEV), "level_2": g["m2"].to(DEV)}
    loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mk, dmin, dmax, regress)
    ref = float(g.np(f"{tag}.loss"))
    assert abs(loss.item() - ref) <= 2e-3 * abs(ref), (loss.item(), ref)
    loss.backward()
    names = [str(x) for x in g.np(f"{tag}.grad_names")]
    norms = g.np(f"{tag}.grad_norms")
    params = dict(model.named_parameters())
    worst = 0.0
    for name, want in zip(names, norms):
        got = params[name].grad
        if want < 0:          # parameters the reference never touches (inner3; confidence head w/o regress)
            assert got is None or float(got.norm()) == 0.0, name
        else:
            assert got is not None, name
            err = abs(float(got.norm()) - want) / max(want, 1e-3)
            worst = max(worst, err)
            assert err <= 6e-2, (name, float(got.norm()), want)
    print(f"train {tag}: loss {loss.item():.6f} (reference {ref:.6f}); worst grad-norm deviation {worst:.2e}")
    if regress:
        d = out["depths_upsampled"][0].detach().cpu()
        rel = (d - g["train.depths_upsampled"]).abs() / g["train.depths_upsampled"]
        assert float((rel > 1e-4).float().mean()) <= 0.02
        assert float((out["depths"]["initial"][0].detach().cpu() - g["train.initial"]).abs().max()) <= 1e-4 * 935
        best = torch.argmax(out["depths"]["probability"][-1], 1, keepdim=True).cpu()
        assert float((best != g["train.best_last"]).float().mean()) <= 0.02
        rm = model.state_dict()["feature_net.conv1.bn.running_mean"].cpu()          # BatchNorm statistics updated
        assert float((rm - g["train.running_mean_conv1"]).abs().max()) <= 1e-5
        assert int(model.state_dict()["feature_net.conv1.bn.num_batches_tracked"]) == 1


@pytest.mark.parametrize("regress", [True, False])
def test_train_step_cfg4_full_size(regress):
    """BASELINE cfg 4 at its workload on the MI355X: one HIP-backed training step at B=1, 5 views, 640x512, 4 iterations
    (train.py:194-243, net.py:131-190) against the reference's loss and 102 gradient norms (tests/golden/train_cfg4.npz)."""
    from itermvs_amd import synthetic
    from itermvs_amd.net import Pipeline, full_loss
    g = golden("train_cfg4.npz")
    tag = "regress" if regress else "noregress"
    sample, gt, mk = synthetic.make_training_sample(num_views=5, height=512, width=640, seed=2)
    model = Pipeline(iteration=int(g.np("iteration")), test=False)
    model.load_state_dict(load_weights("seed0"))
    model = model.to(DEV).train()
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    dmin, dmax = sample["depth_min"].to(DEV), sample["depth_max"].to(DEV)
    torch.cuda.reset_peak_memory_stats()
    out = model(dev(sample["imgs"]), dev(sample["proj_matrices"]), dmin, dmax)
    loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], dev(gt), dev(mk), dmin, dmax, regress)
    ref = float(g.np(f"{tag}.loss"))
    loss.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2**20
    assert abs(loss.item() - ref) <= 2e-3 * abs(ref), (loss.item(), ref)
    params = dict(model.named_parameters())
    worst = 0.0
    for name, want in zip([str(x) for x in g.np(f"{tag}.grad_names")], g.np(f"{tag}.grad_norms")):
        got = params[name].grad
        if want < 0:
            assert got is None or float(got.norm()) == 0.0, name
        else:
            assert got is not None, name
            err = abs(float(got.norm()) - want) / max(want, 1e-3)
            worst = max(worst, err)
            assert err <= 6e-2, (name, float(got.norm()), want)
    print(f"train cfg4 {tag}: loss {loss.item():.6f} (reference {ref:.6f}); worst grad-norm deviation {worst:.2e}; "
          f"peak device memory {peak:.0f} MiB")
    if regress:
        d = out["depths_upsampled"][0].detach().cpu()
        rel = (d[:, :, ::8, ::8] - g["train.depth_sub"]).abs() / g["train.depth_sub"]
        assert float(rel.median()) <= 1e-5 and float((rel > 1e-4).float().mean()) <= 0.05
