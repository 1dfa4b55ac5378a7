"""CPU checks of the formats and drivers either side of the hot path: PFM bytes identical to the
reference writer's, checkpoint naming / format, learning-rate recipe, the flat gradient all-reduce on
two gloo ranks, and the eval driver's flag surface."""
import os
import sys

import numpy as np
import torch

from conftest import ROOT, golden

sys.path.insert(0, ROOT)


def test_pfm_bytes_equal_reference_writer(tmp_path):
    from itermvs_amd.data_io import read_pfm, save_pfm
    g = golden("pfm.npz")
    img = g.np("image")
    path = str(tmp_path / "sub" / "x.pfm")
    save_pfm(path, img)
    assert open(path, "rb").read() == g.np("file_bytes").tobytes()          # byte-identical file
    back, scale = read_pfm(path)
    assert scale == float(g.np("scale")) and np.array_equal(back, g.np("readback"))
    assert np.array_equal(back[..., 0], img)
    rgb = np.random.RandomState(0).rand(4, 6, 3).astype(np.float32)
    save_pfm(path, rgb)
    assert np.array_equal(read_pfm(path)[0], rgb)


def test_train_recipe_and_checkpoint_format(tmp_path):
    import train as T
    assert T.parse_lrepochs("4,8,12:2") == ([4, 8, 12], 0.5) and T.GRAD_CLIP == 2.0
    from itermvs_amd.net import Pipeline
    m = Pipeline(iteration=1, test=False)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for e in (0, 3, 11):
        T.save_checkpoint(str(tmp_path / "model_{:0>6}.ckpt".format(e)), e, m, opt)
    last = T.latest_checkpoint(str(tmp_path))
    assert last.endswith("model_000011.ckpt")
    state = torch.load(last, map_location="cpu", weights_only=False)
    assert set(state) == {"epoch", "model", "optimizer"} and state["epoch"] == 11
    assert all(k.startswith("module.") for k in state["model"]) and len(state["model"]) == 150   # eval.py:124-125
    m2 = Pipeline(iteration=1, test=True)
    m2.load_checkpoint_state(state["model"])
    args = T.build_parser().parse_args(["--regress", "--lr", "0.002", "--resume"])
    assert args.regress and args.resume and args.lr == 0.002 and args.epochs == 16


def test_checkpoint_optimizer_state_is_mode_independent(tmp_path):
    """train.py:152-157 / :103-117: a checkpoint's optimizer state must not carry the saving run's launch mode.  ``--graph``
    keeps the rate and Adam's counters in tensors and sets ``capturable``; ``optimizer.load_state_dict`` REPLACES the fresh
    groups with the saved ones, so without ``restore_optimizer_mode`` a resumed run inherits the wrong mode (an eager checkpoint
    makes CapturedTrainStep raise, a --graph checkpoint freezes the learning-rate schedule inside the graph)."""
    import train as T
    w = torch.nn.Parameter(torch.ones(3))
    # what a --graph run's optimizer looks like (CPU stand-in: tensor rate, capturable flag in the group)
    opt = torch.optim.Adam([w], lr=1e-3)
    w.grad = torch.ones(3)
    opt.step()
    opt.param_groups[0]["lr"] = torch.tensor(5e-4)
    opt.param_groups[0]["capturable"] = True
    sd = T.portable_optimizer_state(opt)
    g = sd["param_groups"][0]
    assert isinstance(g["lr"], float) and abs(g["lr"] - 5e-4) < 1e-9 and g["capturable"] is False
    assert all(st["step"].device.type == "cpu" and float(st["step"]) == 1.0 for st in sd["state"].values())
    torch.save({"optimizer": sd}, str(tmp_path / "o.ckpt"))
    sd = torch.load(str(tmp_path / "o.ckpt"), map_location="cpu", weights_only=False)["optimizer"]
    # eager resume from it: plain float rate, host counters, the scheduler moves the rate
    w2 = torch.nn.Parameter(torch.ones(3))
    opt2 = torch.optim.Adam([w2], lr=1e-3)
    opt2.load_state_dict(sd)
    T.restore_optimizer_mode(opt2, graph=False, dev="cpu")
    g2 = opt2.param_groups[0]
    assert isinstance(g2["lr"], float) and g2["capturable"] is False
    sched = torch.optim.lr_scheduler.MultiStepLR(opt2, [1], gamma=0.5)
    w2.grad = torch.ones(3)
    opt2.step()
    sched.step()
    assert abs(opt2.param_groups[0]["lr"] - 2.5e-4) < 1e-9
    assert float(next(iter(opt2.state.values()))["step"]) == 2.0
    # a "graph" resume: the rate becomes a tensor the scheduler fills IN PLACE (the captured step reads that tensor)
    w3 = torch.nn.Parameter(torch.ones(3))
    opt3 = torch.optim.Adam([w3], lr=1e-3)
    opt3.load_state_dict(sd)
    T.restore_optimizer_mode(opt3, graph=True, dev="cpu")
    lr_t = opt3.param_groups[0]["lr"]
    assert torch.is_tensor(lr_t) and opt3.param_groups[0]["capturable"] is True
    sched = torch.optim.lr_scheduler.MultiStepLR(opt3, [1], gamma=0.5)
    sched.step()
    assert opt3.param_groups[0]["lr"] is lr_t and abs(float(lr_t) - 2.5e-4) < 1e-10


def test_train_driver_builds_the_batch_it_is_asked_for():
    """train.py --batch_size N (train.py:89-90, train_dtu.sh: 4): B = N different samples per rank, the reference's collated
    training schema (datasets/dtu_yao.py:227-232); ranks and steps never see the same scene"""
    import train as T
    args = T.build_parser().parse_args(["--batch_size", "3", "--n_views", "3", "--img_wh", "96", "64"])
    imgs, projs, dmin, dmax, gt, mask = T.synthetic_batch(args, 0, 0, torch.device("cpu"), world=2)
    assert imgs["level_0"].shape == (3, 3, 3, 64, 96) and imgs["level_3"].shape == (3, 3, 3, 8, 12)
    assert projs["level_1"].shape == (3, 3, 4, 4) and dmin.shape == dmax.shape == (3,)
    assert gt["level_0"].shape == mask["level_0"].shape == (3, 1, 64, 96) and gt["level_2"].shape == (3, 1, 16, 24)
    assert not torch.equal(imgs["level_0"][0], imgs["level_0"][1])                # different scenes inside the batch
    other = T.synthetic_batch(args, 0, 1, torch.device("cpu"), world=2)[0]["level_0"]
    nxt = T.synthetic_batch(args, 1, 0, torch.device("cpu"), world=2)[0]["level_0"]
    for a in imgs["level_0"]:
        assert not any(torch.equal(a, b) for b in list(other) + list(nxt))       # nor across ranks / steps
    one = T.synthetic_batch(T.build_parser().parse_args(["--n_views", "3", "--img_wh", "96", "64"]), 0, 0, torch.device("cpu"))
    assert one[0]["level_0"].shape[0] == 1                                       # the default stays --batch_size 1


def test_eval_flags_and_synthetic_dataset_schema():
    import eval as E
    args = E.build_parser().parse_args(["--n_views", "3", "--img_wh", "96", "64", "--iteration", "2", "--num_samples", "3"])
    ds = E.make_dataset(args)
    assert len(ds) == 3
    s = E.collate([ds[0], ds[1]])
    assert s["imgs"]["level_0"].shape == (2, 3, 3, 64, 96) and s["proj_matrices"]["level_2"].shape == (2, 3, 4, 4)
    assert s["filename"][1].format("depth_est", ".pfm") == "scan_synthetic/depth_est/00000001.pfm"   # eval.py:141-151
    assert s["depth_min"].shape == (2,)


def _ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from itermvs_amd import ddp, shard
    shard.init_distributed(backend="gloo")
    torch.manual_seed(rank)
    lin = torch.nn.Linear(5, 3)
    unused = torch.nn.Parameter(torch.zeros(4))                 # like feature_net.inner3: never gets a gradient
    params = list(lin.parameters()) + [unused]
    ddp.broadcast_parameters(lin)
    w0 = lin.weight.detach().clone()
    x = torch.full((2, 5), float(rank + 1))
    lin(x).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    n = ddp.flat_allreduce_gradients(params)
    # numpy arrays travel by value; tensors would be shared through file descriptors that die with this process
    q.put((rank, w0.numpy(), [g.numpy() for g in local], [p.grad.numpy().copy() for p in lin.parameters()], n, unused.grad))
    torch.distributed.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks():
    from conftest import run_ranks
    res = run_ranks(_ddp_worker, 2)
    t = torch.from_numpy
    res = [(r, t(w0), [t(g) for g in loc], [t(g) for g in red], n, un) for r, w0, loc, red, n, un in res]
    assert torch.equal(res[0][1], res[1][1])                                     # same start weights on both ranks
    for i in range(2):                                                           # weight and bias
        mean = (res[0][2][i] + res[1][2][i]) / 2
        assert torch.allclose(res[0][3][i], mean) and torch.allclose(res[1][3][i], mean)
    assert res[0][4] == res[1][4] == 5 * 3 + 3 and res[0][5] is None             # one bucket, unused parameter skipped


def test_synthetic_batch_seeds_stay_in_range_for_validation_steps_at_scale():
    """train.py's validation uses steps from 10_000_019 on: with train_dtu.sh's --batch_size 4, or 8 ranks, the scene seed
    used to exceed numpy's 2**32 limit and crash at the end of epoch 0"""
    import argparse
    import train
    args = argparse.Namespace(batch_size=4, n_views=3, img_wh=[64, 64])
    imgs, projs, dmin, dmax, gt, mask = train.synthetic_batch(args, 10_000_019, 7, "cpu", world=8)
    assert imgs["level_0"].shape == (4, 3, 3, 64, 64) and gt["level_0"].shape == (4, 1, 64, 64)
    args.batch_size = 1
    a = train.synthetic_batch(args, 10_000_020, 7, "cpu", world=8)
    b = train.synthetic_batch(args, 10_000_020, 6, "cpu", world=8)
    assert not torch.equal(a[0]["level_0"], b[0]["level_0"])          # ranks still see different scenes
