#!/usr/bin/env python3
"""Training driver -- counterpart of the reference's ``train.py`` (train.py:54-243) on MI355X.

Same flags and constants (Adam lr 1e-3, MultiStepLR milestones 4,8,12 with gamma 1/2, gradient clip 2.0,
checkpoint ``{'epoch','model','optimizer'}`` with ``module.``-prefixed keys per epoch), but one process
per GPU with a single flat gradient all-reduce over RCCL (itermvs_amd/ddp.py) instead of
``nn.DataParallel``.  Data: ``--dataset synthetic`` (photo-consistent planes with exact depth) or
``module:Class`` yielding the reference's training sample dict (datasets/dtu_yao.py:227-232).
"""
from __future__ import annotations

import argparse
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from itermvs_amd import ddp, shard, synthetic  # noqa: E402
from itermvs_amd.net import Pipeline, full_loss  # noqa: E402

GRAD_CLIP = 2.0            # train.py:213
LR_GAMMA = 0.5             # train.py:124-127 ("lrepochs 4,8,12:2")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="IterMVS training (MI355X)")
    p.add_argument("--mode", default="train", choices=["train", "val"])
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--trainpath"); p.add_argument("--valpath"); p.add_argument("--trainlist"); p.add_argument("--vallist")
    p.add_argument("--epochs", type=int, default=16)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--lrepochs", type=str, default="4,8,12:2")
    p.add_argument("--wd", type=float, default=0.0)
    p.add_argument("--batch_size", type=int, default=1, help="per GPU")
    p.add_argument("--loadckpt", default=None)
    p.add_argument("--logdir", default="./checkpoints/debug")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--regress", action="store_true")
    p.add_argument("--small_image", action="store_true")
    p.add_argument("--summary_freq", type=int, default=20)
    p.add_argument("--save_freq", type=int, default=1)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--iteration", type=int, default=4)
    p.add_argument("--n_views", type=int, default=5)
    p.add_argument("--img_wh", nargs="+", type=int, default=[640, 512])
    p.add_argument("--steps_per_epoch", type=int, default=8, help="synthetic dataset only")
    p.add_argument("--graph", action="store_true",
                   help="replay the whole step (forward, loss, backward, all-reduce, clip, Adam) as ONE hipGraph after three eager "
                        "warm-up steps (itermvs_amd.train_step.CapturedTrainStep): the eager step is launch-bound")
    p.add_argument("--feature_dtype", default="fp32", choices=["fp32", "bf16", "fp16"],
                   help="storage type of the feature pyramids the fused correlation kernels gather from (BASELINE cfg 4: bf16); "
                        "arithmetic, gradients and weights stay fp32")
    return p


def parse_lrepochs(spec: str):
    """'4,8,12:2' -> ([4, 8, 12], 0.5)   (train.py:124-127)"""
    epochs, factor = spec.split(":")
    return [int(e) for e in epochs.split(",")], 1.0 / float(factor)


def synthetic_batch(args, step: int, rank: int, dev, world: int = 1):
    """``--batch_size`` samples per GPU (train.py:89-90; train_dtu.sh: 4), every (step, rank, slot) a different scene."""
    # np.random.RandomState (synthetic._texture) takes seeds below 2**32: validation steps start at 10_000_019, which times
    # world * batch_size * 131 leaves that range from world * batch_size >= 4 on -- wrap (the slots of a batch add < 4096)
    seed = (((step * world + rank) * args.batch_size) * 131) % (2 ** 32 - 4096)
    imgs, projs, dmin, dmax, gt, mask = synthetic.make_training_batch(
        args.batch_size, num_views=args.n_views, height=args.img_wh[1], width=args.img_wh[0], seed=seed)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}  # noqa: E731
    return to(imgs), to(projs), dmin.to(dev), dmax.to(dev), to(gt), to(mask)


def portable_optimizer_state(optimizer) -> dict:
    """``optimizer.state_dict()`` in the form the reference's checkpoints have (train.py:152-157), whatever mode produced it:
    the learning rate a Python float and ``capturable`` off in every group, Adam's step counters on the host.  (A ``--graph``
    run keeps the rate and the counters in device tensors; written as they are they would be baked into -- or rejected by --
    the next run's optimizer.)"""
    sd = optimizer.state_dict()
    groups = []
    for g in sd["param_groups"]:
        g = dict(g)
        g["lr"] = float(g["lr"])
        if "capturable" in g:
            g["capturable"] = False
        groups.append(g)
    state = {}
    for k, st in sd["state"].items():
        st = dict(st)
        if torch.is_tensor(st.get("step")):
            st["step"] = st["step"].detach().to("cpu", torch.float32)
        state[k] = st
    return {"state": state, "param_groups": groups}


def restore_optimizer_mode(optimizer, graph: bool, dev) -> None:
    """after ``optimizer.load_state_dict``: the loaded groups REPLACE the freshly built ones, so ``capturable`` and the type of
    ``lr`` would come from the checkpoint.  Put back what this run asked for: ``--graph`` = capturable, the rate and Adam's step
    counters in device tensors (MultiStepLR fills the rate in place and the captured step reads it); eager = plain float rate,
    host counters."""
    for g in optimizer.param_groups:
        g["capturable"] = bool(graph)
        lr = float(g["lr"])
        g["lr"] = torch.tensor(lr, device=dev) if graph else lr
        if "initial_lr" in g:
            g["initial_lr"] = float(g["initial_lr"])
    for st in optimizer.state.values():
        if torch.is_tensor(st.get("step")):
            st["step"] = st["step"].detach().to(dev if graph else "cpu", torch.float32)


def save_checkpoint(path: str, epoch: int, model: torch.nn.Module, optimizer) -> None:
    """train.py:152-157: keys carry the DataParallel 'module.' prefix so the reference's eval.py loads them."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"epoch": epoch, "model": {"module." + k: v.cpu() for k, v in model.state_dict().items()},
                "optimizer": portable_optimizer_state(optimizer)}, path)


def latest_checkpoint(logdir: str):
    """train.py:103-112: newest ``model_XXXXXX.ckpt`` by numeric suffix."""
    if not os.path.isdir(logdir):
        return None
    found = [f for f in os.listdir(logdir) if re.fullmatch(r"model_\d+\.ckpt", f)]
    return os.path.join(logdir, max(found, key=lambda f: int(re.findall(r"\d+", f)[-1]))) if found else None


def train_step(model, optimizer, batch, regress: bool):
    """train.py:194-243 (train_sample)."""
    imgs, projs, dmin, dmax, gt, mask = batch
    model.train()
    optimizer.zero_grad(set_to_none=True)
    out = model(imgs, projs, dmin, dmax)
    loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, regress)
    loss.backward()
    ddp.flat_allreduce_gradients(model.parameters())
    torch.nn.utils.clip_grad_norm_(model.parameters(), GRAD_CLIP)
    optimizer.step()
    err = (out["depths_upsampled"][0].detach() - gt["level_0"]).abs().mean()
    return float(loss.detach()), float(err)


@torch.no_grad()
def val_step(model, batch, regress: bool, iteration: int):
    """train.py:245-295 (test_sample): eval-mode BatchNorm, the training-form outputs, loss and the reference's scalars."""
    imgs, projs, dmin, dmax, gt, mask = batch
    model.eval()
    out = model(imgs, projs, dmin, dmax)
    loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, regress)
    m0, m2 = mask["level_0"] > 0.5, mask["level_2"] > 0.5
    abs_err = lambda est, g, m: float((est[m] - g[m]).abs().mean())                       # utils.py AbsDepthError_metrics
    thres = lambda est, g, m, t: float(((est[m] - g[m]).abs() > t).float().mean())        # utils.py Thres_metrics
    final, first = out["depths_upsampled"][-1], out["depths"]["combine"][0]
    scalars = {"loss": float(loss), "abs_error_initial": abs_err(first, gt["level_2"], m2),
               "thres1mm_initial": thres(first, gt["level_2"], m2, 1.0),
               "abs_error_final_full": abs_err(final, gt["level_0"], m0)}
    for t in (1, 2, 4, 8):
        scalars[f"thres{t}mm_final_full"] = thres(final, gt["level_0"], m0, float(t))
    for j in range(1, iteration + 1):
        scalars[f"thres1mm_gru_{j}"] = thres(out["depths"]["combine"][j], gt["level_2"], m2, 1.0)
        scalars[f"abs_error_gru_{j}"] = abs_err(out["depths"]["combine"][j], gt["level_2"], m2)
    return scalars


def validate(model, args, rank: int, world: int, dev) -> dict:
    """train.py:177-190 (test): every rank scores its share of the validation steps; the means are averaged over ranks"""
    total, n = {}, 0
    for step in range(args.steps_per_epoch):
        sc = val_step(model, synthetic_batch(args, 10_000_019 + step, rank, dev, world), args.regress, args.iteration)
        for k, v in sc.items():
            total[k] = total.get(k, 0.0) + v
        n += 1
        if rank == 0:
            print("Iter {}/{}, test loss = {:.3f}".format(step, args.steps_per_epoch, sc["loss"]))
    keys = sorted(total)
    t = torch.tensor([total[k] / max(n, 1) for k in keys], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t)
        t /= world
    return dict(zip(keys, t.tolist()))


def main() -> None:
    args = build_parser().parse_args()
    rank, local_rank, world = shard.init_distributed()
    torch.manual_seed(args.seed)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    model = Pipeline(iteration=args.iteration, test=False).to(dev)
    model.feature_dtype = args.feature_dtype
    # train.py:98.  --graph: Adam's step counter and the learning rate live on the device (capturable; MultiStepLR fills the
    # rate tensor in place)
    optimizer = torch.optim.Adam(model.parameters(), lr=torch.tensor(args.lr, device=dev) if args.graph else args.lr, betas=(0.9, 0.999),
                                 weight_decay=args.wd, capturable=args.graph)
    start_epoch = 0
    ckpt = latest_checkpoint(args.logdir) if args.resume else args.loadckpt
    if ckpt:
        state = torch.load(ckpt, map_location="cpu", weights_only=False)
        model.load_checkpoint_state(state["model"], strict=False)
        if args.resume:
            optimizer.load_state_dict(state["optimizer"])
            restore_optimizer_mode(optimizer, args.graph, dev)          # the loaded groups carry the SAVING run's mode
            start_epoch = state["epoch"] + 1
    ddp.broadcast_parameters(model)
    milestones, gamma = parse_lrepochs(args.lrepochs)
    sched = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=gamma, last_epoch=start_epoch - 1)
    if args.dataset != "synthetic":
        raise SystemExit("only --dataset synthetic is built in; plug a dataset module in via itermvs_amd")
    if args.mode == "val":                                                               # train.py:297-300
        means = validate(model, args, rank, world, dev)
        if rank == 0:
            print("final", means)
        shard.barrier()
        return
    captured = None
    if args.graph:
        from itermvs_amd.train_step import CapturedTrainStep
        captured = CapturedTrainStep(model, optimizer, args.regress, clip=GRAD_CLIP)
    for epoch in range(start_epoch, args.epochs):
        for step in range(args.steps_per_epoch):
            t0 = time.time()
            batch = synthetic_batch(args, epoch * args.steps_per_epoch + step, rank, dev, world)
            if captured is not None:
                loss, err = (float(t) for t in captured.step(batch))
                captured.check()
            else:
                loss, err = train_step(model, optimizer, batch, args.regress)
            if rank == 0 and step % args.summary_freq == 0:
                print("Epoch {}/{}, Iter {}/{}, train loss = {:.3f}, abs depth error = {:.3f} mm, time = {:.3f}".format(
                    epoch, args.epochs, step, args.steps_per_epoch, loss, err, time.time() - t0))
        sched.step()
        if rank == 0 and (epoch + 1) % args.save_freq == 0:
            save_checkpoint("{}/model_{:0>6}.ckpt".format(args.logdir, epoch), epoch, model, optimizer)
        means = validate(model, args, rank, world, dev)                                   # train.py:160-175
        if captured is not None:
            captured.check()                       # a projection flagged by a validation forward belongs to THIS phase
        if rank == 0:
            print("avg_test_scalars:", means)
    shard.barrier()


if __name__ == "__main__":
    main()
