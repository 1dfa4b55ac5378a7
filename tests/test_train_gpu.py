"""Training form of the pipeline on the MI355X (HIP warp forward/backward inside autograd) vs
the reference's forward + full_loss + backward captured in tests/golden/train_small.npz."""
import pytest
import torch

from conftest import check_gradient_slices, golden, gradient_chaos_floor, load_weights, run_ranks

pytestmark = pytest.mark.gpu
DEV = "cuda"
# Full-tensor gradient agreement with the reference (sliced gradients of every differentiated parameter).  The training
# graph is chaotic like inference (an arg-max flip moves a pixel's regression window; some gradients are small sums of large
# cancelling terms): moving the images by 2e-6 -- the size of a convolution back-end's rounding -- changes the CPU oracle's
# own sliced gradients by 0.3 % (median) at the small shape and by 3 % (median) / 28 % (90th percentile) / ~45 % (CorrNet
# weights) at the cfg-4 shape, measured by conftest.gradient_chaos_floor inside each test.  A parameter is held to the
# base tolerance below unless 4x its measured floor is larger; at least 10 parameters must be held to the base tolerance.
GRAD_REL_L2 = 8e-2
GRAD_MIN_COS = 0.995
GRAD_REL_L2_CFG4 = 0.06       # measured on the MI355X (profiles/r03): worst 1.3 % (regress) / 2.2 % (no regress), cosine >= 0.99994
GRAD_MIN_COS_CFG4 = 0.999


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
    # the gradient TENSORS (256 evenly spaced elements of each of the ~100 differentiated parameters) against the
    # reference's: direction and magnitude.  Measured on the MI355X: see the printed line (profiles/r03).
    sample = {"imgs": {"level_0": g["imgs"]}, "proj_matrices": {f"level_{l}": g[f"proj.level_{l}"] for l in (1, 2, 3)},
              "depth_min": g["depth_min"], "depth_max": g["depth_max"]}
    floor, _, _ = gradient_chaos_floor(load_weights("seed0"), sample, {"level_0": g["gt0"], "level_2": g["gt2"]},
                                       {"level_0": g["m0"], "level_2": g["m2"]}, int(g.np("iteration")), regress)
    rep = check_gradient_slices(g, tag, {n: p.grad for n, p in params.items()}, rel_l2=GRAD_REL_L2, min_cos=GRAD_MIN_COS,
                                floor=floor, floor_factor=1.0, min_checked=60)
    print(f"train {tag}: loss {loss.item():.6f} (reference {ref:.6f}); worst grad-norm deviation {worst:.2e}; gradient slices {rep}")
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
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    floor, _, _ = gradient_chaos_floor(load_weights("seed0"), sample, gt, mk, int(g.np("iteration")), regress)
    rep = check_gradient_slices(g, tag, {n: p.grad for n, p in params.items()}, rel_l2=GRAD_REL_L2_CFG4, min_cos=GRAD_MIN_COS_CFG4,
                                floor=floor, floor_factor=1.0, min_checked=10)
    print(f"train cfg4 {tag}: loss {loss.item():.6f} (reference {ref:.6f}); worst grad-norm deviation {worst:.2e}; "
          f"gradient slices {rep}; peak device memory {peak:.0f} MiB")
    if regress:
        d = out["depths_upsampled"][0].detach().cpu()
        rel = (d[:, :, ::8, ::8] - g["train.depth_sub"]).abs() / g["train.depth_sub"]
        assert float(rel.median()) <= 1e-5 and float((rel > 1e-4).float().mean()) <= 0.05


@pytest.mark.parametrize("regress", [True, False])
def test_train_step_cfg4_batch2(regress):
    """The cfg-4 step BATCHED (train_dtu.sh: --batch_size 4 over DataParallel replicas, i.e. B >= 2 per forward;
    train.py:89-90): B = 2 different scenes, 5 views, 640x512, 4 iterations, against the reference's loss, gradient norms and
    sliced gradient tensors of the same batch (tests/golden/train_cfg4_b2.npz; the reference takes its `batch == 2` branch
    of differentiable_warping there, module.py:78-84).  The batch is the one `train.py --batch_size 2` builds."""
    from itermvs_amd import synthetic
    from itermvs_amd.net import Pipeline, full_loss
    g = golden("train_cfg4_b2.npz")
    tag = "regress" if regress else "noregress"
    b = int(g.np("batch"))
    imgs, projs, dmin, dmax, gt, mk = synthetic.make_training_batch(b, num_views=5, height=512, width=640, seed=2, hole_fraction=0.1)
    model = Pipeline(iteration=int(g.np("iteration")), test=False)
    model.load_state_dict(load_weights("seed0"))
    model = model.to(DEV).train()
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    out = model(dev(imgs), dev(projs), dmin.to(DEV), dmax.to(DEV))
    assert out["depths_upsampled"][0].shape == (b, 1, 512, 640)
    loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], dev(gt), dev(mk), dmin.to(DEV), dmax.to(DEV), regress)
    ref = float(g.np(f"{tag}.loss"))
    loss.backward()
    assert abs(loss.item() - ref) <= 2e-3 * abs(ref), (loss.item(), ref)
    params = dict(model.named_parameters())
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    sample = {"imgs": imgs, "proj_matrices": projs, "depth_min": dmin, "depth_max": dmax}
    floor, _, _ = gradient_chaos_floor(load_weights("seed0"), sample, gt, mk, int(g.np("iteration")), regress)
    worst = 0.0
    for name, want in zip([str(x) for x in g.np(f"{tag}.grad_names")], g.np(f"{tag}.grad_norms")):
        got = params[name].grad
        if want < 0:
            assert got is None or float(got.norm()) == 0.0, name
        else:
            assert got is not None, name
            err = abs(float(got.norm()) - want) / max(want, 1e-3)
            worst = max(worst, err)
            # a norm moves by at most the relative L2 change of the tensor: the parameter's own chaos floor bounds what
            # rounding-size input noise does to it on the reference itself (e.g. the lateral layers' biases: sums of
            # cancelling terms over every pixel)
            assert err <= max(6e-2, 2.0 * floor.get(name, 0.0)), (name, float(got.norm()), want, floor.get(name))
    # floor_factor 2: the floor is ONE sample of the reference's own sensitivity (a single 2e-6 perturbation); measured on the
    # MI355X (profiles/r04): the three ill-conditioned tensors of this batch (inner1.bias, output1.bias, one CorrNet weight)
    # deviate by 1.15x / 1.1x / 1.2x their sampled floor.  Held to the BASE tolerance (2x floor below it): 69-87 of 96 tensors
    # without --regress, 39 of 99 with it (the confidence loss makes most floors exceed 3 %; median deviation 1.8 %) -- the
    # count moves with the host that samples the floor, so only >= 10 are REQUIRED, like in the B = 1 test above
    rep = check_gradient_slices(g, tag, {n: p.grad for n, p in params.items()}, rel_l2=GRAD_REL_L2_CFG4, min_cos=GRAD_MIN_COS_CFG4,
                                floor=floor, floor_factor=2.0, min_checked=10)
    print(f"train cfg4 B={b} {tag}: loss {loss.item():.6f} (reference {ref:.6f}); worst grad-norm deviation {worst:.2e}; "
          f"gradient slices {rep}")
    if regress:
        d = out["depths_upsampled"][0].detach().cpu()
        rel = (d[:, :, ::8, ::8] - g["train.depth_sub"]).abs() / g["train.depth_sub"]
        assert float(rel.median()) <= 1e-5 and float((rel > 1e-4).float().mean()) <= 0.05


def _gpu_step(model, sample, gt, mk, regress=True):
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    dmin, dmax = sample["depth_min"].to(DEV), sample["depth_max"].to(DEV)
    from itermvs_amd.net import full_loss
    out = model(dev(sample["imgs"]), dev(sample["proj_matrices"]), dmin, dmax)
    loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], dev(gt), dev(mk), dmin, dmax, regress)
    loss.backward()
    return out, loss


def test_train_step_cfg4_bf16_feature_storage():
    """BASELINE cfg 4 AS STATED: 5-view 640x512 training step with bf16 feature storage (train.py --feature_dtype bf16).
    The fused correlation forward AND backward gather bf16 features (fp32 arithmetic, fp32 gradients).  Checks:
    (1) against the pinned CPU oracle evaluated with the same storage model (oracle ``feature_storage=torch.bfloat16``:
        features rounded to bf16 before the matching stages, straight-through) -- loss within 1.5 %, every sliced gradient
        within max(30 %, 4x its measured chaos floor under bf16 storage), the median within 15 %;
    (2) the stated tolerance against the REFERENCE's fp32 step (tests/golden/train_cfg4.npz): loss within 2 % (bf16
        storage perturbs the correlations by ~2^-9, which flips arg-max bins on a few per cent of the pixels)."""
    from itermvs_amd import synthetic
    from itermvs_amd.net import Pipeline
    g = golden("train_cfg4.npz")
    sample, gt, mk = synthetic.make_training_sample(num_views=5, height=512, width=640, seed=2)
    w0 = load_weights("seed0")
    model = Pipeline(iteration=4, test=False)
    model.load_state_dict(w0)
    model = model.to(DEV).train()
    model.feature_dtype = "bf16"
    out, loss = _gpu_step(model, sample, gt, mk)
    params = dict(model.named_parameters())
    ref32 = float(g.np("regress.loss"))
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    # the pinned oracle with the same storage model, and its chaos floor under that storage.  bf16 rounding makes the
    # training step far more chaotic than fp32: two CPU evaluations of this very oracle step (different hosts / thread counts)
    # gave losses 121.873 and 121.386, and the 2e-6 image perturbation moves its sliced gradients by 10 % (median) and up
    # to 200 % (CorrNet weights).  So: loss within 1.5 % of the bf16-storage oracle and 2 % of the fp32 reference, every
    # sliced gradient within max(30 %, 4x its floor), the MEDIAN within 15 %.  The tight gate of the bf16 backward is the
    # kernel-level test (test_corr_iter_backward_matches_autograd[bfloat16], 1e-4 on identical rounded features).
    floor, g_oracle, loss_oracle = gradient_chaos_floor(w0, sample, gt, mk, 4, True, feature_storage=torch.bfloat16)
    for name, p in params.items():
        if g_oracle.get(name) is None:
            assert p.grad is None or float(p.grad.norm()) == 0.0, name
    rep = check_gradient_slices(None, None, {n: p.grad for n, p in params.items()}, rel_l2=0.3, min_cos=0.9,
                                floor=floor, floor_factor=4.0, min_checked=5, want=g_oracle)
    print(f"train cfg4 bf16 storage: loss {loss.item():.6f}; oracle with bf16 storage {loss_oracle:.6f}; reference fp32 {ref32:.6f}; "
          f"gradient slices vs the bf16-storage oracle {rep}")
    assert rep["median_l2"] <= 0.15
    assert abs(loss.item() - loss_oracle) <= 1.5e-2 * abs(loss_oracle), (loss.item(), loss_oracle)
    assert abs(loss.item() - ref32) <= 2e-2 * abs(ref32), (loss.item(), ref32)      # stated tolerance against the fp32 reference


def _ddp_worker(rank, world, port, height, width, feature_dtype, q):
    """one rank of the two-rank training step: both ranks share cuda:0 (the GPU box has one GPU), gloo rendezvous"""
    import os
    import torch.distributed as dist
    from itermvs_amd import ddp, synthetic
    from itermvs_amd.net import Pipeline
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = Pipeline(iteration=2, test=False)
    model.load_state_dict(load_weights("seed0"))
    model = model.to(DEV).train()
    model.feature_dtype = feature_dtype
    ddp.broadcast_parameters(model)
    sample, gt, mk = synthetic.make_training_sample(num_views=3, height=height, width=width, seed=10 + rank)   # rank-local B=1
    _gpu_step(model, sample, gt, mk)
    # numpy arrays through the queue (torch tensors travel as shared-memory handles that die with the producer)
    local = {n: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for n, p in model.named_parameters()}
    n_red = ddp.flat_allreduce_gradients(model.parameters())
    torch.cuda.synchronize()
    reduced = {n: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for n, p in model.named_parameters()}
    q.put((rank, local, reduced, n_red))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("feature_dtype", ["fp32", "bf16"])
def test_two_rank_training_step_on_one_gpu(feature_dtype):
    """BASELINE cfg 4's data parallelism (train.py:194-243 with one process per GPU instead of DataParallel): two processes
    share cuda:0, each runs a rank-local B=1 HIP-backed training step on its own sample, then ONE flat all-reduce
    (ddp.flat_allreduce_gradients).  Parameter by parameter the result must be the mean of the two single-rank gradients,
    identical on both ranks, and parameters without gradient stay without."""
    res = run_ranks(_ddp_worker, 2, 128, 160, feature_dtype, timeout=600)
    (_, l0, r0, n0), (_, l1, r1, n1) = res
    assert n0 == n1 and n0 > 300_000
    tt = lambda d: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()}
    l0, l1, r0, r1 = tt(l0), tt(l1), tt(r0), tt(r1)
    checked = 0
    for name in l0:
        if l0[name] is None:
            assert l1[name] is None and r0[name] is None and r1[name] is None, name
            continue
        mean = (l0[name] + l1[name]) / 2
        assert torch.equal(r0[name], r1[name]), name                              # every rank holds the same averaged gradient
        assert torch.allclose(r0[name], mean, rtol=1e-6, atol=1e-12), name        # == mean of the two rank-local gradients
        # the ranks really saw different data (a gradient that is zero in exact arithmetic -- the bias in front of a softmax -- is
        # rounding noise of a few 2^-26 and may coincide)
        assert not torch.equal(l0[name], l1[name]) or float(l0[name].abs().max()) <= 1e-7, name
        checked += 1
    assert checked >= 95


def test_captured_training_step_equals_the_eager_step():
    """train_step.CapturedTrainStep (forward + full_loss + backward + clip + Adam as ONE hipGraph; reference train.py:194-243) against
    the eager step on the same batches from the same initial weights: the losses of six steps (three eager warm-ups, the
    capture, two replays on NEW batches copied into the static buffers) and the final parameters."""
    from itermvs_amd import synthetic
    from itermvs_amd.net import Pipeline, full_loss
    from itermvs_amd.train_step import CapturedTrainStep
    to = lambda d: {k: v.to(DEV) for k, v in d.items()}  # noqa: E731
    batches = []
    for i in range(6):
        imgs, projs, dmin, dmax, gt, mask = synthetic.make_training_batch(2, num_views=3, height=96, width=128, seed=10 * i, hole_fraction=0.1)
        batches.append((to(imgs), to(projs), dmin.to(DEV), dmax.to(DEV), to(gt), to(mask)))

    def fresh():
        m = Pipeline(iteration=2, test=False)
        m.load_state_dict(load_weights("seed0"))
        return m.to(DEV).train()

    eager, losses_e = fresh(), []
    # (a small rate: at 1e-3 the random-init network on random scenes leaves the stable regime within five steps -- the loss
    #  jumps from 95 to 800 -- and two trajectories that differ by Adam's device-side bias correction, 1e-7, drift percent apart)
    lr = 1e-5
    opt_e = torch.optim.Adam(eager.parameters(), lr=lr, betas=(0.9, 0.999))
    for imgs, projs, dmin, dmax, gt, mask in batches:
        opt_e.zero_grad(set_to_none=True)
        out = eager(imgs, projs, dmin, dmax)
        loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(eager.parameters(), 2.0)
        opt_e.step()
        losses_e.append(float(loss))

    graphed = fresh()
    opt_g = torch.optim.Adam(graphed.parameters(), lr=torch.tensor(lr, device=DEV), betas=(0.9, 0.999), capturable=True)
    cap = CapturedTrainStep(graphed, opt_g, regress=True, clip=2.0, warmup=3)
    losses_g = [float(cap.step(bt)[0]) for bt in batches]
    cap.check()
    assert cap.graph is not None and cap.calls == 6
    assert abs(losses_g[0] - losses_e[0]) <= 1e-5 * abs(losses_e[0])
    for a, b in zip(losses_g, losses_e):
        assert abs(a - b) <= 2e-3 * abs(b), (losses_g, losses_e)
    w0 = load_weights("seed0")
    moved = torch.cat([(p.detach().cpu() - w0[name]).abs().reshape(-1) for name, p in graphed.named_parameters()])
    apart = torch.cat([(p.detach() - q.detach()).abs().reshape(-1).cpu() for p, q in zip(graphed.parameters(), eager.parameters())])
    assert 2e-5 <= float(moved.max()) <= 6.1e-5, float(moved.max())          # six Adam steps of 1e-5 each really happened ...
    # ... and both trajectories took them together.  (Adam normalises every update to ~lr: an element whose gradient is rounding
    # noise may step the other way in one run -- a few elements apart by a step or two, none by the distance travelled)
    assert float(apart.mean()) <= 0.02 * float(moved.mean()), (float(apart.mean()), float(moved.mean()))
    assert float(apart.max()) <= 0.5 * float(moved.max()), (float(apart.max()), float(moved.max()))
    with pytest.raises(ValueError):
        CapturedTrainStep(graphed, torch.optim.Adam(graphed.parameters(), lr=1e-3), regress=True)


@pytest.mark.gpu
def test_resume_across_launch_modes(tmp_path):
    """reference train.py:103-117,152-157 (--resume) with this repo's --graph: a checkpoint written by an eager run resumes as a
    captured step and one written by a captured run resumes eagerly -- ``optimizer.load_state_dict`` replaces the param groups
    with the saved ones, ``train.restore_optimizer_mode`` puts this run's mode back (capturable flag, tensor / float learning
    rate, where Adam's counters live) and MultiStepLR still reaches the replayed step."""
    import train as T
    from itermvs_amd import synthetic
    from itermvs_amd.net import Pipeline
    from itermvs_amd.train_step import CapturedTrainStep
    to = lambda d: {k: v.to(DEV) for k, v in d.items()}  # noqa: E731
    batches = []
    for i in range(5):
        imgs, projs, dmin, dmax, gt, mask = synthetic.make_training_batch(1, num_views=3, height=64, width=96, seed=7 * i)
        batches.append((to(imgs), to(projs), dmin.to(DEV), dmax.to(DEV), to(gt), to(mask)))

    def fresh():
        m = Pipeline(iteration=1, test=False)
        m.load_state_dict(load_weights("seed0"))
        return m.to(DEV).train()

    # eager run -> checkpoint -> --graph resume
    m = fresh()
    opt = torch.optim.Adam(m.parameters(), lr=1e-5)
    T.train_step(m, opt, batches[0], True)
    T.save_checkpoint(str(tmp_path / "a" / "model_000000.ckpt"), 0, m, opt)
    state = torch.load(str(tmp_path / "a" / "model_000000.ckpt"), map_location="cpu", weights_only=False)
    g = state["optimizer"]["param_groups"][0]
    assert isinstance(g["lr"], float) and g.get("capturable", False) is False
    m2 = fresh()
    m2.load_checkpoint_state(state["model"], strict=False)
    opt2 = torch.optim.Adam(m2.parameters(), lr=torch.tensor(1e-5, device=DEV), capturable=True)
    opt2.load_state_dict(state["optimizer"])
    T.restore_optimizer_mode(opt2, True, DEV)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt2, [1], gamma=0.5)
    cap = CapturedTrainStep(m2, opt2, regress=True, warmup=1)
    before = [p.detach().clone() for p in m2.parameters()]
    for bt in batches[1:4]:
        cap.step(bt)
    cap.check()
    assert cap.graph is not None
    lr_t = opt2.param_groups[0]["lr"]
    sched.step()                                              # halves the rate IN the tensor the graph reads
    assert opt2.param_groups[0]["lr"] is lr_t and abs(float(lr_t) - 5e-6) < 1e-10
    mid = [p.detach().clone() for p in m2.parameters()]
    cap.step(batches[4])
    torch.cuda.synchronize()
    step_full = max(float((a - b).abs().max()) for a, b in zip(before, mid))
    step_half = max(float((a - b).abs().max()) for a, b in zip(mid, m2.parameters()))
    assert 0 < step_half <= 0.75e-5 and step_full >= 1e-5 * 0.9, (step_half, step_full)   # Adam moves ~lr per step
    steps = {float(st["step"]) for st in opt2.state.values()}
    assert steps == {5.0}, steps                              # 1 eager + 4 resumed steps, counted on the device

    # --graph run -> checkpoint -> eager resume
    T.save_checkpoint(str(tmp_path / "b" / "model_000001.ckpt"), 1, m2, opt2)
    state = torch.load(str(tmp_path / "b" / "model_000001.ckpt"), map_location="cpu", weights_only=False)
    g = state["optimizer"]["param_groups"][0]
    assert isinstance(g["lr"], float) and g["capturable"] is False
    assert all(st["step"].device.type == "cpu" for st in state["optimizer"]["state"].values())
    m3 = fresh()
    m3.load_checkpoint_state(state["model"], strict=False)
    opt3 = torch.optim.Adam(m3.parameters(), lr=1e-5)
    opt3.load_state_dict(state["optimizer"])
    T.restore_optimizer_mode(opt3, False, DEV)
    g3 = opt3.param_groups[0]
    assert isinstance(g3["lr"], float) and abs(g3["lr"] - 5e-6) < 1e-10 and g3["capturable"] is False
    loss, _ = T.train_step(m3, opt3, batches[0], True)
    assert loss == loss and {float(st["step"]) for st in opt3.state.values()} == {6.0}
