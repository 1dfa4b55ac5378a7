"""Training-step measurement at BASELINE cfg 4's per-GPU workload (train_dtu.sh: 5 views, 640x512, 4 GRU iterations,
--batch_size per GPU, Adam + gradient clip 2.0; train.py:194-243): forward (training graph, fused correlation kernels),
full_loss, backward, flat gradient all-reduce (a no-op on one rank), clip, Adam -- `train.train_step` without its two
host read-backs inside the timed region.  One JSON line: ms per step, samples/s, peak device memory.

    python tools/train_bench.py [--batch 1] [--feature_dtype bf16] [--regress] [--steps 10] [--warmup 3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import ddp, synthetic  # noqa: E402
from itermvs_amd.net import Pipeline, full_loss  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--wh", nargs=2, type=int, default=[640, 512])
    ap.add_argument("--iteration", type=int, default=4)
    ap.add_argument("--feature_dtype", default="bf16", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--regress", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile", type=int, default=0, help="after the timing: torch.profiler over this many steady-state steps, top kernels by device time")
    ap.add_argument("--phases", action="store_true", help="also time forward / backward / optimizer separately (synchronising)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Pipeline(iteration=args.iteration, test=False).to(dev)
    model.feature_dtype = args.feature_dtype
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    # the batch train.py builds for --batch_size (B different scenes / reference views)
    imgs, projs, dmin, dmax, gt, mask = synthetic.make_training_batch(args.batch, num_views=args.views, height=args.wh[1], width=args.wh[0])
    to = lambda d: {k: v.to(dev) for k, v in d.items()}  # noqa: E731
    imgs, projs, gt, mask, dmin, dmax = to(imgs), to(projs), to(gt), to(mask), dmin.to(dev), dmax.to(dev)
    params = list(model.parameters())

    def fwd():
        out = model(imgs, projs, dmin, dmax)
        return full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, args.regress)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = fwd()
        loss.backward()
        ddp.flat_allreduce_gradients(params)
        torch.nn.utils.clip_grad_norm_(params, 2.0)
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    res = {"metric": "training step (cfg 4 per-GPU workload)", "ms_per_step": ms, "samples_per_s": args.batch * 1e3 / ms,
           "batch": args.batch, "views": args.views, "wh": args.wh, "iteration": args.iteration, "feature_dtype": args.feature_dtype,
           "regress": args.regress, "steps": args.steps, "warmup": args.warmup, "loss": float(loss.detach()),
           "peak_mem_MiB": torch.cuda.max_memory_allocated() / 2 ** 20}
    if args.phases:
        ph = {"forward": 0.0, "backward": 0.0, "clip+adam": 0.0}
        for _ in range(args.steps):
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize(); t = time.perf_counter()
            loss = fwd()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            torch.nn.utils.clip_grad_norm_(params, 2.0)
            opt.step()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            ph["forward"] += (t1 - t) * 1e3 / args.steps
            ph["backward"] += (t2 - t1) * 1e3 / args.steps
            ph["clip+adam"] += (t3 - t2) * 1e3 / args.steps
        res["phases_ms"] = ph
    print(json.dumps(res))
    if args.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(args.profile):
                step()
            torch.cuda.synchronize()
        rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages()
                if "CUDA" in str(e.device_type) and e.self_device_time_total > 0]
        rows.sort(key=lambda r: -r[2])
        total = sum(r[2] for r in rows)
        print(f"device kernels: {total / args.profile / 1e3:.2f} ms per step in {sum(r[1] for r in rows) // args.profile} launches")
        for name, count, us in rows[:45]:
            print(f"{us / args.profile / 1e3:8.3f} ms {count // args.profile:5d} x  {name[:150]}")


if __name__ == "__main__":
    main()
