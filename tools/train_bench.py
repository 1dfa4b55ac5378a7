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

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import benchmarks  # noqa: E402


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
    ap.add_argument("--cudnn-benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True (reference train.py:23): MIOpen times its "
                    "applicable solvers per convolution shape once instead of taking the immediate-mode pick.  On a box without a MIOpen "
                    "kernel cache the search compiles every candidate: > 11 minutes before the first step at B = 4 (profiles/r05/r05af)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (train_step.CapturedTrainStep)")
    ap.add_argument("--phases", action="store_true", help="also time forward / backward / optimizer separately (synchronising)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = args.cudnn_benchmark
    res = benchmarks.train_step_leg(dev, batch=args.batch, views=args.views, height=args.wh[1], width=args.wh[0], iteration=args.iteration,
                                    feature_dtype=args.feature_dtype, regress=args.regress, warmup=args.warmup, steps=args.steps,
                                    phases=args.phases, graph=args.graph)
    step = res.pop("_step")
    res = {"metric": "training step (cfg 4 per-GPU workload)", **res}
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
