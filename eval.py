#!/usr/bin/env python3
"""Depth inference driver -- counterpart of the reference's ``eval.py`` ``save_depth()`` (eval.py:104-151)
for the MI355X engine.  Same flags, checkpoint format and output layout
(``<outdir>/<scan>/depth_est/<view:08d>.pfm`` and ``.../confidence/...``), one process per GPU:

    python eval.py --dataset synthetic --n_views 5 --img_wh 640 512 --outdir ./outputs [--loadckpt model.ckpt]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 eval.py ...

Each rank takes the reference views ``rank, rank + world, ...`` (no collective on the data path).
``--dataset module:Class`` accepts any dataset yielding the reference's sample dict
(datasets/dtu_yao_eval.py:154-158); the reference's own loaders (cv2 / PIL based) are outside this
repository's scope.  ``--filter`` runs the reference's filter / fusion stage (eval.py:215-325) afterwards: for every scan
folder under ``--testpath`` that has a ``pair.txt``, ``cams_1/`` and ``images/``, the depth / confidence PFMs written above
are fused into ``<outdir>/<scan>.ply`` by ``itermvs_amd.fusion.filter_depth`` (one HIP launch per reference view); the
camera intrinsics are rescaled by ``--img_wh`` / original image size and the points coloured from the resized images like
eval.py:231-232,251-252,295.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from itermvs_amd import shard, synthetic  # noqa: E402
from itermvs_amd.data_io import save_pfm  # noqa: E402
from itermvs_amd.net import Pipeline  # noqa: E402


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Predict depth maps (MI355X engine)")
    p.add_argument("--model", default="IterMVS")
    p.add_argument("--dataset", default="synthetic",
                   help="'synthetic', 'folder' (scan folders under --testpath: pair.txt, cams_1/, images/; decode on the host, "
                        "normalise / resize / pyramid on the GPU, prefetched) or module:Class of an MVSDataset")
    p.add_argument("--testpath")
    p.add_argument("--testlist")
    p.add_argument("--split", default="intermediate")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--n_views", type=int, default=5)
    p.add_argument("--img_wh", nargs="+", type=int, default=[640, 512], help="width height")
    p.add_argument("--loadckpt", default=None)
    p.add_argument("--outdir", default="./outputs")
    p.add_argument("--display", action="store_true")
    p.add_argument("--iteration", type=int, default=4)
    p.add_argument("--projection", default="device_fp64", choices=["device_fp64", "host_fp32"],
                   help="how src_proj @ inverse(ref_proj) (module.py:77-90) is composed: on the GPU in fp64 rounded once (default), or "
                        "on the host in fp32 operation for operation like the reference (tap indices of a reference run on this "
                        "host); the cameras then stay on the host, no synchronisation")
    p.add_argument("--conv_arithmetic", default="bf16x3", choices=["bf16x3", "fp32"],
                   help="3x3 convolutions with more than 8 input channels: exact three-term bf16 split of both operands on the bf16 "
                        "matrix instructions (default, fp32-rounding-class error) or the exact fp32 MFMA (bit-for-bit an fmaf chain)")
    p.add_argument("--feature_dtype", default="fp32", choices=["fp32", "bf16", "fp16"],
                   help="storage type of the feature pyramids (BASELINE cfg 5: fp16); arithmetic stays fp32")
    p.add_argument("--no_graphs", action="store_true",
                   help="launch every kernel from Python (~85 launches per depth map) instead of replaying one hipGraph per depth map "
                        "(captured once per image shape; results are bit-identical)")
    p.add_argument("--geo_pixel_thres", type=float, default=1)
    p.add_argument("--geo_depth_thres", type=float, default=0.01)
    p.add_argument("--photo_thres", type=float, default=0.3)
    p.add_argument("--num_samples", type=int, default=8, help="synthetic dataset: number of reference views")
    p.add_argument("--filter", action="store_true", help="fuse the saved depth maps of every scan into a point cloud (eval.py:311-325)")
    return p


class SyntheticMVSDataset(torch.utils.data.Dataset):
    """Photo-consistent synthetic scenes in the reference's sample-dict schema."""

    def __init__(self, n_views: int, img_wh, count: int):
        self.n_views, self.w, self.h, self.count = n_views, img_wh[0], img_wh[1], count

    def __len__(self):
        return self.count

    def __getitem__(self, idx):
        s = synthetic.make_scene_sample(num_views=self.n_views, height=self.h, width=self.w, seed=idx)
        return {"imgs": {k: v[0] for k, v in s["imgs"].items()},
                "proj_matrices": {k: v[0] for k, v in s["proj_matrices"].items()},
                "depth_min": s["depth_min"][0], "depth_max": s["depth_max"][0],
                "filename": "scan_synthetic/{}/" + "{:0>8}".format(idx) + "{}"}


def scan_list(args):
    """--testlist (one scan folder name per line, eval.py:47 of the reference) or every folder of --testpath with a pair.txt"""
    if args.testlist:
        with open(args.testlist) as f:
            return [line.rstrip() for line in f if line.strip()]
    return sorted(d for d in os.listdir(args.testpath) if os.path.isfile(os.path.join(args.testpath, d, "pair.txt")))


def make_dataset(args):
    if args.dataset == "synthetic":
        return SyntheticMVSDataset(args.n_views, args.img_wh, args.num_samples)
    if args.dataset == "folder":
        from itermvs_amd.scan_dataset import ScanFolderDataset
        return ScanFolderDataset(args.testpath, scan_list(args), args.n_views, tuple(args.img_wh))
    mod, cls = args.dataset.split(":")
    return getattr(importlib.import_module(mod), cls)(args.testpath, args.testlist, args.n_views, tuple(args.img_wh))


def collate(samples):
    out = {"imgs": {}, "proj_matrices": {}}
    for key in ("imgs", "proj_matrices"):
        for lvl in samples[0][key]:
            out[key][lvl] = torch.stack([torch.as_tensor(s[key][lvl]) for s in samples])
    out["depth_min"] = torch.stack([torch.as_tensor(s["depth_min"], dtype=torch.float32) for s in samples])
    out["depth_max"] = torch.stack([torch.as_tensor(s["depth_max"], dtype=torch.float32) for s in samples])
    out["filename"] = [s["filename"] for s in samples]
    return out


def tocuda(x, dev):
    """utils.py:60-67"""
    if isinstance(x, torch.Tensor):
        return x.to(dev, non_blocking=True)
    if isinstance(x, dict):
        return {k: tocuda(v, dev) for k, v in x.items()}
    return x


def load_model(args, dev) -> Pipeline:
    model = Pipeline(iteration=args.iteration, test=True)
    model.feature_dtype = getattr(args, "feature_dtype", "fp32")
    model.projection = getattr(args, "projection", "device_fp64")
    model.conv_arithmetic = getattr(args, "conv_arithmetic", "bf16x3")
    # one hipGraph replay per depth map (bit-identical to the eager launches); host_fp32 with device-resident cameras cannot be
    # captured -- save_depth keeps the cameras on the host for that mode
    model.use_graphs = not getattr(args, "no_graphs", False)
    # which arithmetic produced the PFMs of this run (device_fp64 moves ~1e-5 of the tap floors against the reference's host fp32
    # composition; host_fp32 follows module.py:77-90 operation for operation)
    print("depth maps: projection = {}, conv arithmetic = {}, feature storage = {}".format(
        model.projection, model.conv_arithmetic, model.feature_dtype))
    if args.loadckpt:
        print("loading model {}".format(args.loadckpt))
        state = torch.load(args.loadckpt, map_location="cpu", weights_only=False)
        model.load_checkpoint_state(state["model"])            # eval.py:124-125 ('module.' prefixed keys)
    else:
        model.load_state_dict(synthetic.random_state_dict(0))
    return model.to(dev).eval()


def save_depth(args) -> int:
    rank, local_rank, world = shard.init_distributed()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dataset = make_dataset(args)
    mine = shard.shard_indices(len(dataset), rank, world)
    model = load_model(args, dev)
    done = 0
    if args.dataset == "folder":
        return save_depth_folder(args, dataset, mine, model, dev)
    with torch.no_grad():
        for i in range(0, len(mine), args.batch_size):
            t0 = time.time()
            sample = collate([dataset[j] for j in mine[i:i + args.batch_size]])
            cu = tocuda(sample, dev)
            if model.projection == "host_fp32":
                cu["proj_matrices"] = sample["proj_matrices"]      # composed on the host: the cameras never need the device
            out = model(cu["imgs"], cu["proj_matrices"], cu["depth_min"], cu["depth_max"])
            depth = out["depths_upsampled"].cpu().numpy()      # D2H + sync, like tensor2numpy (eval.py:135)
            conf = out["confidence_upsampled"].cpu().numpy()
            model.check_projection_finite()                    # module.py:83,87 (deferred; the device is already idle)
            print("Iter {}/{}, time = {:.3f}".format(i // args.batch_size, (len(mine) + args.batch_size - 1) // args.batch_size,
                                                      time.time() - t0))
            for name, d, c in zip(sample["filename"], depth, conf):
                save_pfm(os.path.join(args.outdir, name.format("depth_est", ".pfm")), np.squeeze(d, 0))
                save_pfm(os.path.join(args.outdir, name.format("confidence", ".pfm")), np.squeeze(c, 0))
                done += 1
    shard.barrier()
    return done


def save_depth_folder(args, dataset, mine, model, dev) -> int:
    """the folder input side: a host thread decodes the next views while the GPU works; uint8 upload + pyramid kernel on a
    side stream (itermvs_amd.scan_dataset.Prefetcher); one reference view per forward like the reference's batch_size 1.
    The loop is software-pipelined by one depth map on ONE stream: forward n and the asynchronous download of its two maps
    (and of the engine's NaN flag, module.py:83,87) are enqueued, then the host writes the PFMs of map n-1 while the GPU
    works on n -- the reference's loop (eval.py:128-151) leaves the GPU idle for every file it writes."""
    from itermvs_amd.scan_dataset import Prefetcher
    done = 0
    host = [None, None]                                        # two sets of pinned result buffers, used alternately
    events = [torch.cuda.Event(), torch.cuda.Event()]
    model.check_nan = False                                    # the flag travels with the results instead of stalling the stream

    def finish(job) -> None:
        sample, k, n, t0 = job
        events[k].synchronize()                                # forward n-1 and its downloads are done; forward n is running
        depth, conf, flag = host[k]
        if int(flag[0]) != 0:
            raise AssertionError("nan in proj (singular or non-finite camera matrix, module.py:83,87)")
        print("Iter {}/{}, time = {:.3f}".format(n, len(mine), time.time() - t0))
        name = sample["filename"]
        save_pfm(os.path.join(args.outdir, name.format("depth_est", ".pfm")), np.squeeze(depth.numpy()[0], 0))
        save_pfm(os.path.join(args.outdir, name.format("confidence", ".pfm")), np.squeeze(conf.numpy()[0], 0))

    with torch.no_grad():
        prev = None
        for n, (sample, (imgs, projs, dmin, dmax)) in enumerate(Prefetcher(dataset, mine, dev)):
            t0 = time.time()
            if model.projection == "host_fp32":
                projs = {key: v.unsqueeze(0) for key, v in sample["proj_matrices"].items()}      # CPU cameras
            out = model(imgs, projs, dmin, dmax)
            k = n % 2
            d, c = out["depths_upsampled"], out["confidence_upsampled"]
            if host[k] is None or host[k][0].shape != d.shape:
                host[k] = (torch.empty(d.shape, dtype=d.dtype).pin_memory(), torch.empty(c.shape, dtype=c.dtype).pin_memory(),
                           torch.zeros((1,), dtype=torch.int32).pin_memory())
            host[k][0].copy_(d, non_blocking=True)
            host[k][1].copy_(c, non_blocking=True)
            flag = model.projection_flag()
            if flag is not None:
                host[k][2].copy_(flag, non_blocking=True)
            events[k].record()
            if prev is not None:
                finish(prev)
                done += 1
            prev = (sample, k, n, t0)
        if prev is not None:
            finish(prev)
            done += 1
    shard.barrier()
    return done


def fuse_scans(args) -> int:
    """eval.py:311-325: one point cloud per scan; scans are sharded over the ranks like the reference views"""
    from itermvs_amd import fusion
    rank, local_rank, world = shard.init_distributed()
    dev = "cuda:%d" % local_rank
    scans = sorted(d for d in os.listdir(args.testpath)
                   if os.path.isfile(os.path.join(args.testpath, d, "pair.txt")) and os.path.isdir(os.path.join(args.outdir, d)))
    n = 0
    for i in shard.shard_indices(len(scans), rank, world):
        scan = scans[i]
        stats = fusion.filter_depth(os.path.join(args.testpath, scan), os.path.join(args.outdir, scan),
                                    os.path.join(args.outdir, scan + ".ply"), args.geo_pixel_thres, args.geo_depth_thres,
                                    args.photo_thres, device=dev, img_wh=tuple(args.img_wh))
        for v, (g, ph, f) in stats.items():
            print("processing {}, ref-view{:0>2}, geo_mask:{:3f} photo_mask:{:3f} final_mask: {:3f}".format(scan, v, g, ph, f))
        n += 1
    shard.barrier()
    return n


if __name__ == "__main__":
    a = build_parser().parse_args()
    print("argv:", sys.argv[1:])
    save_depth(a)
    if a.filter:
        fuse_scans(a)
