#!/usr/bin/env python3
"""Benchmark of the IterMVS hot path on MI355X: depth-maps/s at BASELINE cfg 1/2
(1 reference + 4 source views, 640x512, 4 GRU iterations, test mode).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of ``Pipeline(test=True).forward`` over one batch of synthetic reference
views that is already resident in HBM.  One process per GPU, reference views sharded over the
ranks with no data-path collective (weak scaling); the timed region is bracketed by a barrier +
device synchronise and the maximum over ranks is reported.  Rank 0 prints ONE JSON line with
the throughput, the HBM roofline of the dominant hand-written kernel (itermvs_corr_iter, timed
with HIP events on its launch stream inside the timed region) and a CPU baseline (the oracle,
``kind: port``, timed on this box's host cores over a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")   # errors only: keeps MIOpen workspace warnings out of the JSON log
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def pmc_summary_file():
    """newest profiles/rNN_pmc_kernels.json (written by tools/pmc_kernels.sh on the GPU box, committed per round)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_kernels.json")))
    return found[-1] if found else None


def algorithmic_bytes(s: int, h: int, w: int, batch: int, iters: int, e: int = 4):
    """DESIGN.md 'Algorithmic bytes' (shared with the other legs: itermvs_amd/benchmarks.py)"""
    from itermvs_amd.benchmarks import algorithmic_bytes as impl
    return impl(s, h, w, batch, iters, e)


def _round_floats(x, digits: int = 6):
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _round_floats(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round_floats(v, digits) for v in x]
    return x


def workload_name(views: int, height: int, width: int, iters: int) -> str:
    """which BASELINE.json config the shape is (the bench line must name what actually ran)"""
    key = (views, height, width, iters)
    if key == (5, 512, 640, 4):
        return "BASELINE cfg 2 (= cfg 1 on 1xMI355X)"
    if key in ((5, 1152, 1600, 4), (6, 1152, 1600, 4)):
        return "BASELINE cfg 3 shape (1600x1152: the reference crops 1200 to a multiple of 32)"
    if key == (11, 1280, 1920, 8):
        return "BASELINE cfg 5 shape"
    return "custom shape (not a BASELINE config)"


def self_launch_command(n_gpus: int, argv):
    """the torch.distributed.run command line `python bench.py --gpus N` turns itself into when no launcher set
    WORLD_SIZE: one process per GPU over RCCL, rendezvous on 127.0.0.1 (the container hostname may not resolve)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def cpu_baseline(args, target_seconds: float = 15.0):
    """Time the CPU oracle (port of the reference's PyTorch-CPU path) on the same workload."""
    import torch
    from itermvs_amd import synthetic
    from oracle import itermvs_oracle as O

    weights = synthetic.random_state_dict(0)
    s = synthetic.make_sample(batch=1, num_views=args.views, height=args.height, width=args.width, seed=0)
    run = lambda: O.pipeline_forward(weights, s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"],
                                     iteration=args.iters, test=True)
    with torch.no_grad():
        run()                                   # warm-up
        # torch's default (one thread per core) is far from the best choice for these small ops on a
        # many-core host: give the CPU path its best of a few thread counts
        best_t, first = None, None
        for nt in sorted({8, 16, 32, min(64, os.cpu_count() or 8)}):
            if nt > (os.cpu_count() or 8):
                continue
            torch.set_num_threads(nt)
            run()
            t0 = time.perf_counter()
            run()
            dt = time.perf_counter() - t0
            if first is None or dt < first:
                best_t, first = nt, dt
        torch.set_num_threads(best_t)
        n = max(2, min(20, int(target_seconds / max(first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "depth-maps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} depth maps of the same workload, best of 8/16/32/64 torch threads (oracle, {os.cpu_count()} logical CPUs)",
            "s_per_depth_map": dt / n}


def transfers_leg(args, dev, samples, world, u8: bool = False, in_flight: int = 1):
    """depth-maps/s with PCIe in the loop: H2D of the sample (imgs level_0, cameras, depth range) and D2H of the two output
    maps, overlapped with compute through the static buffers of ``in_flight + 1`` graph runners.
    ``u8``: the images travel as the decoded uint8 RGB arrays (what eval.py --dataset folder uploads, 5x fewer bytes) and
    itermvs_image_pyramid normalises them into the runner's static input on the copy stream.
    ``in_flight``: depth maps queued on the GPU while the host waits for the oldest one; 1 = the protocol form (one compute
    stream), more = one compute stream per runner (the serving configuration, reported under `pipelined`)."""
    import torch
    from itermvs_amd import ops, shard, synthetic
    from itermvs_amd.engine import GraphedRunner, InferenceEngine
    from itermvs_amd.net import Pipeline
    m = Pipeline(iteration=args.iters, test=True)
    m.load_state_dict(synthetic.random_state_dict(0))
    m = m.to(dev).eval()
    eng = InferenceEngine(m.weights(), args.iters, args.feature_dtype, conv_arithmetic=args.conv_arithmetic)
    imgs0, projs0, dmin0, dmax0 = samples[0]
    pj = {l: projs0[f"level_{l}"].float() for l in (1, 2, 3)}
    nr = in_flight + 1
    runners = [GraphedRunner(eng, imgs0["level_0"].float(), pj, dmin0.float(), dmax0.float()) for _ in range(nr)]
    host_in = []
    for imgs, projs, dmin, dmax in samples:
        img = imgs["level_0"].float().cpu()
        if u8:      # [1,V,3,H,W] in -1..1 -> [V,H,W,3] uint8, like a decoded image file
            img = ((img[0].permute(0, 2, 3, 1) + 1) * 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
        host_in.append((img.pin_memory(),
                        torch.stack([projs[f"level_{l}"].float() for l in (1, 2, 3)]).cpu().pin_memory(),
                        dmin.float().cpu().pin_memory(), dmax.float().cpu().pin_memory()))
    host_out = [tuple(torch.empty(o.shape, dtype=o.dtype).pin_memory() for o in r.out) for r in runners]
    raw_dev = [torch.empty(host_in[0][0].shape, dtype=torch.uint8, device=dev) for _ in runners] if u8 else None
    s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    s_cmp = [torch.cuda.Stream(device=dev) for _ in range(nr if in_flight > 1 else 1)]
    ev_in = [torch.cuda.Event() for _ in range(nr)]
    ev_done = [torch.cuda.Event() for _ in range(nr)]
    ev_out = [torch.cuda.Event() for _ in range(nr)]
    state = {"primed": False}

    def upload(i: int) -> None:
        k = i % nr
        r = runners[k]
        h_img, h_proj, h_min, h_max = host_in[i % len(host_in)]
        with torch.cuda.stream(s_in):
            if u8:
                raw_dev[k].copy_(h_img, non_blocking=True)
                ops.image_pyramid(raw_dev[k], args.height, args.width, all_levels=False, out0=r.imgs[0])
            else:
                r.imgs.copy_(h_img, non_blocking=True)
            r.proj_stack.copy_(h_proj, non_blocking=True)
            r.depth_min.copy_(h_min, non_blocking=True)
            r.depth_max.copy_(h_max, non_blocking=True)
            ev_in[k].record(s_in)

    # The HOST orders the copies -- no stream ever waits on another stream's event.  (With hipStreamWaitEvent in front of every
    # replay, and the copy streams waiting on the replay's event, the same work measured 1.25-1.35 ms per map against 1.13 ms
    # resident; upload and replay side by side WITHOUT any ordering cost nothing: tools/transfer_lab.py, profiles/r03.)
    # Per step: enqueue replay i (its inputs were confirmed on the device earlier); wait for the OLDEST replay in flight,
    # i - in_flight; only then issue its download and the upload of sample i+1 into the runner it has released; wait for both.
    # The GPU always has `in_flight` replays queued and each compute stream carries one event record per replay.
    def step(i: int) -> None:
        k = i % nr
        r = runners[k]
        if not state["primed"]:                       # very first step: the first sample has to be on the device
            torch.cuda.synchronize()
            upload(i)
            ev_in[k].synchronize()
            state.update(primed=True, first=i)
        with torch.cuda.stream(s_cmp[k % len(s_cmp)]):
            r(r.imgs, r.projs, r.depth_min, r.depth_max)    # static inputs: no staging copies, one graph launch
            ev_done[k].record()
        p = i - in_flight                              # the oldest replay in flight
        if p >= state["first"]:
            pk = p % nr
            ev_done[pk].synchronize()                  # replay p is done (the younger ones are running / queued)
            with torch.cuda.stream(s_out):
                for h, d in zip(host_out[pk], runners[pk].out):
                    h.copy_(d, non_blocking=True)
                ev_out[pk].record(s_out)
        upload(i + 1)                                  # runner (i + 1) % nr == p % nr: released by the wait above (or never used yet)
        ev_in[(i + 1) % nr].synchronize()
        if p >= state["first"]:
            ev_out[p % nr].synchronize()               # depth + confidence of replay p are on the host
    regions = shard.timed_regions(step, args.steps, max(args.warmup, 4), args.repeats)
    elapsed = shard.median(regions)
    eng.check_projection_finite()
    h2d = sum(t.numel() * t.element_size() for t in host_in[0])
    d2h = sum(t.numel() * t.element_size() for t in host_out[0])
    return {"value": world * args.steps * args.batch / elapsed, "unit": "depth-maps/s", "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": min(regions) / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3,
            "h2d_bytes_per_step": h2d, "d2h_bytes_per_step": d2h, "depth_maps_in_flight": in_flight,
            "images": "uint8 RGB, normalised on the GPU (itermvs_image_pyramid)" if u8 else "float32 level_0 tensor",
"protocol": "pinned host in/out, copies on their own streams, ordered by the host (see transfers_leg)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1, help="reference views per step and GPU")
    ap.add_argument("--feature-dtype", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="storage type of the feature pyramids (BASELINE cfg 4 bf16 / cfg 5 fp16); arithmetic stays fp32")
    ap.add_argument("--projection", default="device_fp64", choices=["device_fp64", "host_fp32"],
                    help="host_fp32: the cameras stay on the host and src @ inverse(ref) is composed there in fp32 like module.py:77-90 "
                         "(the reference's tap indices on this host), per step, inside the timed region; the graphs read the composed "
                         "matrices from a static buffer refreshed through pinned memory")
    ap.add_argument("--conv-arithmetic", default="bf16x3", choices=["bf16x3", "fp32"],
                    help="3x3 convolutions with more than 8 input channels: exact three-term bf16 split on the bf16 MFMA (default) or the "
                         "exact fp32 MFMA (A/B on one box)")
    ap.add_argument("--side-branch", action="store_true",
                    help="A/B: ref_quarter and the up-sampling-weight launch as a parallel graph branch (measured slower: profiles/r06)")
    ap.add_argument("--repeats", type=int, default=9,
                    help="the --steps long timed region (barrier + device synchronise on both sides) is run this many times back "
                         "to back; `value` / `ms_per_step` are the MEDIAN region, min / max are reported beside them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-transfers", action="store_true", help="skip the host-buffers-in / host-buffers-out leg")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` legs (BASELINE cfg 3 / cfg 5 shapes, cfg 4 training step) run after the timed region")
    ap.add_argument("--minimal", action="store_true",
                    help="only the timed region (no CPU baseline, pipelined / transfer legs, conv roofline pass): profiling runs")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent reference views in flight per GPU (one HIP stream + engine workspace each)")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel instead of replaying hipGraph segments")
    ap.add_argument("--pipeline-streams", type=int, default=4,
                    help="extra measurement: throughput with this many reference views in flight (0/1 = skip)")
    args = ap.parse_args()
    if args.minimal:
        args.no_cpu_baseline = args.no_transfers = args.no_other_configs = True
        args.pipeline_streams = 0
    if args.projection == "host_fp32":          # an A/B of the headline region only: the side legs keep the default composition
        args.no_transfers = True
        args.pipeline_streams = 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same flags>`
        sys.stdout.flush()
        os.execv(sys.executable, self_launch_command(args.gpus, sys.argv[1:]))

    import torch
    from itermvs_amd import ops, shard, synthetic
    from itermvs_amd.net import Pipeline

    rank, local_rank, world = shard.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev_index = local_rank % max(1, torch.cuda.device_count())      # (ranks share a GPU only in the gloo test of the N > 1 path)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # this rank's host threads and (first-touch) pinned buffers next to its GPU: before any pinned allocation below
    affinity = shard.bind_to_gpu_numa(dev_index)
    # box calibration (two fixed micro-kernels, <= 0.3 s): what THIS chip delivers, printed beside `value` so that driver
    # lines from different boxes of the pool can be compared (`value_normalised`)
    from itermvs_amd import benchmarks as _bm
    box = _bm.box_probe(dev) if rank == 0 else None

    # graph mode, one stream: EIGHT runners replayed round-robin on that stream, each bound to one of the four resident samples
    # (see `resident` below).  Runners 0 and 1 carry event-record nodes around one / two of their fused correlation launches;
    # while step i runs, the pairs of step i-1 are read, so the launches are measured inside the timed region without ever
    # draining the stream.  A bracket costs ~5 us of graph time: with three brackets spread over eight runners the timing
    # costs `value` ~1.9 us per step (it was 3.75 us with four runners in rounds 4-5).
    ab = 8 if (args.streams == 1 and not args.eager) else 1
    models, streams = [], []
    for _ in range(args.streams * ab):
        m = Pipeline(iteration=args.iters, test=True)
        m.load_state_dict(synthetic.random_state_dict(0))
        m.use_graphs = not args.eager
        m.feature_dtype = args.feature_dtype
        m.projection = args.projection
        m.conv_arithmetic = args.conv_arithmetic
        m.side_branch = args.side_branch
        models.append(m.to(dev).eval())
        streams.append(torch.cuda.Stream(device=dev) if args.streams > 1 else torch.cuda.current_stream(dev))
    model = models[0]
    cam_dev = "cpu" if args.projection == "host_fp32" else dev

    # this rank's shard of the synthetic reference views, resident in HBM before timing starts
    n_resident = 4
    samples = []
    for i in range(n_resident):
        s = synthetic.make_sample(batch=args.batch, num_views=args.views, height=args.height, width=args.width,
                                  seed=rank * n_resident + i)
        samples.append(({k: v.to(dev) for k, v in s["imgs"].items()},
                        {k: v.to(cam_dev) for k, v in s["proj_matrices"].items()},
                        s["depth_min"].to(dev), s["depth_max"].to(dev)))

    sink = []
    graph_prof = []                                    # (kind, ms) of the timed region's captured launches
    n_models = len(models)

    def read_pairs(i: int) -> None:
        """event pairs of the graph replayed at step i (waits for that replay only)"""
        runner = next(iter(models[i % n_models]._runners.values()))
        first, count = runner.profile_pairs
        got = ops.profile_graph_read(first, count)
        if i >= args.warmup:
            graph_prof.extend(got)

    # graph mode, one stream: runner k's STATIC input buffers ARE resident sample k (filled once at capture), and the step
    # hands exactly those tensors to Pipeline.forward, which then replays without a staging copy -- the sample is resident
    # in HBM where the graph reads it.  (Other modes pass the resident sample and pay one 20 MB device copy per step.)
    resident = {}

    def step(i: int) -> None:
        k = i % n_models
        imgs, projs, dmin, dmax = resident.get(k) or samples[i % n_resident]
        with torch.cuda.stream(streams[k]):
            out = models[k](imgs, projs, dmin, dmax)      # graph mode: one hipGraph replay on this stream
        sink[:] = [out["depths_upsampled"], out["confidence_upsampled"]]
        if ab > 1 and i >= 1:
            read_pairs(i - 1)

    # HIP-event pairs around the fused kernels' launches (on their launch stream).  Graph mode: external
    # event-record nodes captured with the launches (enabled BEFORE the capture below); eager mode: plain records.
    per_step = args.iters + 1
    # (graph mode: an event-record node costs ~5-6 us of graph time on either side of the launch it brackets, so each replay
    # carries ONE timed launch: runner A a corr_iter (GRU iteration 1), runner B the corr_init)
    total_steps = args.steps * args.repeats
    ops.profile_enable((total_steps + args.warmup) * per_step + 8 * per_step, mask=0x3)
    for k in range(n_models):                          # set-up, not a step: capture every runner's hipGraph
        with torch.cuda.stream(streams[k]):
            if ab > 1:
                # an event-record node costs ~5 us of graph time: runner A carries them on iterations 0, 2, ...,
                # runner B on 1, 3, ... -- every iteration position is sampled in every second replay
                from itermvs_amd.engine import InferenceEngine
                # runner 0 brackets the corr_iter launch of GRU iteration 0, runner 1 that of iteration 2 and the corr_init launch,
                # runners 2 and 3 carry no timing nodes at all (a bracket costs ~5 us of graph time)
                models[k]._engine = InferenceEngine(models[k].weights(), models[k].iteration, args.feature_dtype, args.projection,
                                                    args.conv_arithmetic, args.side_branch)
                models[k]._engine_version = models[k]._weights_version()      # (the engine belongs to the current weights)
                models[k]._engine.profile_iterations = {0} if k == 0 else ({min(2, args.iters - 1)} if k == 1 else set())
                models[k]._engine.profile_init = (k == 1)
            models[k](*samples[k % n_resident])
            if ab > 1:
                r = next(iter(models[k]._runners.values()))
                cams = samples[k % n_resident][1] if args.projection == "host_fp32" else {f"level_{l}": r.projs[l] for l in (1, 2, 3)}
                resident[k] = ({"level_0": r.imgs}, cams, r.depth_min, r.depth_max)
    torch.cuda.synchronize()
    ops.profile_collect(max_samples=4096)              # drop the set-up launches' samples

    regions = shard.timed_regions(step, args.steps, args.warmup, args.repeats)
    elapsed = shard.median(regions)                    # the contract's K-step region; median of --repeats of them
    # this rank's own clock around the same regions (timed_regions reports the max over ranks): a straggler GPU shows up as
    # min < max across ranks in SCALE_r*.json
    own_ms = shard.median(shard.last_local_regions()) / args.steps * 1e3
    rank_ms = shard.gather_over_ranks(own_ms)
    if ab > 1:
        read_pairs(args.warmup + total_steps - 1)
        prof = graph_prof
    else:
        prof = ops.profile_collect(max_samples=(total_steps + args.warmup) * per_step + 8 * per_step)[-total_steps * per_step:]
    ops.profile_enable(0)
    # the shader clock this chip holds under the timed workload itself (20 more steps between two counter stamps, rank 0)
    sclk_workload = None
    if rank == 0:
        counter = iter(range(10 ** 9))

        def plain_step() -> None:                      # a step without the event read-back of the timed region
            k = next(counter) % n_models
            imgs, projs, dmin, dmax = resident.get(k) or samples[k % n_resident]
            with torch.cuda.stream(streams[k]):
                models[k](imgs, projs, dmin, dmax)
        if args.streams == 1:
            sclk_workload = _bm.workload_clock(dev, plain_step, 20)
    maps = world * args.steps * args.batch
    value = maps / elapsed

    s_views = args.views - 1
    b_iter, b_init, b_map = algorithmic_bytes(s_views, args.height, args.width, args.batch, args.iters,
                                               4 if args.feature_dtype == "fp32" else 2)
    t_iter = [ms for kind, ms in prof if kind == 1]
    t_init = [ms for kind, ms in prof if kind == 2]
    roofline = None
    if t_iter:
        avg_ms = sum(t_iter) / len(t_iter)
        achieved = b_iter / (avg_ms * 1e-3) / 1e9
        kernel_name = ops.corr_iter_kernel_name()
        roofline = {"bound": "hbm", "kernel": f"itermvs_corr_iter ({kernel_name})", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": b_iter, "avg_launch_ms": avg_ms, "launches_timed": len(t_iter),
                    "iterations_sampled": sorted({0, min(2, args.iters - 1)}) if ab > 1 else list(range(args.iters)),
                    # (graph mode: external hipEvent record nodes around the launch inside the replayed hipGraph, read for every replay of
                    #  the timed regions; runner 0 brackets the corr_iter launch of GRU iteration 0 -- hypotheses around the first, noisy
                    #  depth map --, runner 1 that of iteration 2 and the corr_init launch, runners 2 and 3 carry no timing nodes: a
                    #  bracket costs ~5 us of graph time, ~4 us per step on average, included in `value`)
                    "timing": "hipEvent nodes inside the replayed graphs" if ab > 1 else "hipEvent pairs on the launch stream"}
        # HBM bytes per launch: rocprofv3 --pmc passes of THIS command (tools/pmc_kernels.sh -> tools/pmc_summary.py ->
        # profiles/<round>_pmc_kernels.json); taken only if the summary names the kernel that ran here and the same workload
        pmc_file = pmc_summary_file()
        if pmc_file:
            pmc = json.load(open(pmc_file))
            entry = next((v for k, v in pmc.get("kernels", {}).items() if k.startswith(kernel_name + "<") or k == kernel_name), None)
            if (entry and pmc.get("workload") == [args.views, args.height, args.width, args.batch] and args.feature_dtype == "fp32"
                    and "traffic_bytes_per_launch" in entry):
                roofline["traffic"] = entry["traffic_bytes_per_launch"]
                roofline["traffic_source"] = f"profiles/{os.path.basename(pmc_file)}: " + pmc["source"]
        if t_init:
            init_ms = sum(t_init) / len(t_init)
            roofline["corr_init"] = {"kernel": "itermvs_corr_init (corr_init_kernel<32>)", "avg_launch_ms": init_ms,
                                     "launches_timed": len(t_init), "algorithmic_bytes_per_launch": b_init,
                                     "achieved": b_init / (init_ms * 1e-3) / 1e9, "frac": b_init / (init_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # the r02 form of the step, kept beside `value` so that round-over-round deltas are attributable: the sample is NOT the
    # runner's static input, every step pays the ~20 MB device-to-device staging copy (one itermvs_copy_multi launch) in front
    # of the replay -- what a consumer that receives fresh device tensors per depth map sees.  Runners 2 and 3 (no timing nodes).
    staged = None
    if ab > 1 and not args.minimal:
        def sstep(i: int) -> None:
            k = 2 + (i & 1)
            with torch.cuda.stream(streams[k]):
                out = models[k](*samples[i % n_resident])
            sink[:] = [out["depths_upsampled"], out["confidence_upsampled"]]
        sel = shard.timed_steps(sstep, args.steps, 4)
        staged = {"value": world * args.steps * args.batch / sel, "unit": "depth-maps/s", "steps": args.steps,
                  "ms_per_step": sel / args.steps * 1e3,
                  "what": "inputs copied device-to-device into the static buffers each step"}

    # extra: independent reference views pipelined on several HIP streams of the same GPU (each stream replays
    # its own hipGraph segments).  Reported separately: with concurrent streams the HIP-event bracket of a
    # single kernel also measures queueing behind the other streams, so the contract's `value` / `roofline`
    # above stay single-stream and verifiable against rocprofv3.
    pipelined = None
    if args.streams == 1 and not args.eager and args.pipeline_streams > 1:
        ns = args.pipeline_streams
        pm, pstreams = [], []
        for _ in range(ns):
            m = Pipeline(iteration=args.iters, test=True)
            m.load_state_dict(synthetic.random_state_dict(0))
            m.use_graphs = True
            m.feature_dtype = args.feature_dtype
            m.conv_arithmetic = args.conv_arithmetic
            pm.append(m.to(dev).eval())
            pstreams.append(torch.cuda.Stream(device=dev))
        for k in range(ns):
            with torch.cuda.stream(pstreams[k]):
                pm[k](*samples[0])
        torch.cuda.synchronize()

        def pstep(i: int) -> None:
            with torch.cuda.stream(pstreams[i % ns]):
                pm[i % ns](*samples[i % n_resident])

        psteps = max(args.steps, 4 * ns)
        pel = shard.timed_steps(pstep, psteps, 2 * ns)
        pipelined = {"streams": ns, "value": world * psteps * args.batch / pel, "unit": "depth-maps/s",
                     "steps": psteps, "ms_per_step": pel / psteps * 1e3}
        del pm
        if not args.no_transfers and args.batch == 1:
            # the serving configuration: host buffers in / out with `ns` depth maps in flight (one compute stream each)
            pipelined["with_transfers"] = transfers_leg(args, dev, samples, world, u8=True, in_flight=ns)

    # the SURVEY section 8(d) form of the metric: host buffers in, host buffers out (eval.py:130-137).  Pinned host
    # samples are copied into the static inputs of two alternating graph runners on a copy stream while the previous depth
    # map computes; both outputs go back to pinned host memory on a third stream.  Reported beside `value`, never as it.
    with_transfers = None
    if args.streams == 1 and not args.eager and not args.no_transfers:
        with_transfers = transfers_leg(args, dev, samples, world)
        if args.batch == 1:
            with_transfers["uint8_images"] = transfers_leg(args, dev, samples, world, u8=True)

    # second roofline: the matrix-core convolutions (FeatureNet, CorrNet, ConvGRU, heads) -- timed with
    # HIP-event pairs around every itermvs_conv2d launch in a short EXTRA pass after the timed region
    # (event pairs around ~100 launches per step would perturb the throughput measurement)
    conv_roofline = None
    if rank == 0 and not args.minimal:
        n_extra = 3
        ops.profile_enable(n_extra * 160 + 8, mask=0x4)
        ops.CONV_FLOP_COUNTER.update(enabled=True, flops=0.0, launches=0)
        eager_model = Pipeline(iteration=args.iters, test=True)
        eager_model.feature_dtype = args.feature_dtype
        eager_model.conv_arithmetic = args.conv_arithmetic
        eager_model.load_state_dict(synthetic.random_state_dict(0))
        eager_model = eager_model.to(dev).eval()
        eager_model(*samples[0])                       # warm-up (not timed: profiling collects below)
        torch.cuda.synchronize()
        ops.profile_collect(max_samples=4096)
        ops.CONV_FLOP_COUNTER.update(flops=0.0, launches=0)
        for i in range(n_extra):
            eager_model(*samples[i % n_resident])
        torch.cuda.synchronize()
        ops.CONV_FLOP_COUNTER["enabled"] = False
        conv_ms = [ms for kind, ms in ops.profile_collect(max_samples=n_extra * 160 + 8) if kind == 3]
        ops.profile_enable(0)
        if conv_ms and len(conv_ms) == ops.CONV_FLOP_COUNTER["launches"]:
            tf = ops.CONV_FLOP_COUNTER["flops"] / (sum(conv_ms) * 1e-3) / 1e12
            conv_roofline = {"bound": "mfma", "kernel": "itermvs_conv2d (conv_tile3_kernel / conv_tile_kernel / conv_mfma_kernel / lateral_up2_kernel) + itermvs_res_chain16 + itermvs_lateral_conv3x3 + itermvs_gru_conv",
                             "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                             "launches_per_step": len(conv_ms) // n_extra, "ms_per_step": sum(conv_ms) / n_extra,
                             "gflop_per_step": ops.CONV_FLOP_COUNTER["flops"] / n_extra / 1e9,
                             "timing": "hipEvent pairs around every launch, 3 extra eager steps",
                             "peak_note": "fp32-input MFMA peak; the 3x3 layers with > 8 input channels run the bf16x3 split (3 bf16 MFMAs "
                                          "per 4 fp32 ones): the FLOPs counted are the convolution's, not the instructions'"}

    # the other BASELINE configurations on the driver's record (rank 0 of a 1-GPU run, after everything timed above):
    # cfg-3 shape (5 views 1600x1152, fp32 and fp16 feature storage), cfg-5 shape (11 views 1920x1280, 8 iterations, fp16
    # storage) -- one captured hipGraph each, 5 warm-up + 10 timed replays -- and cfg 4's per-GPU training step (B = 4, bf16
    # feature storage, --regress, 3 warm-up + 5 timed steps)
    other = None
    if rank == 0 and world == 1 and not args.no_other_configs and (args.views, args.height, args.width, args.iters) == (5, 512, 640, 4):
        from itermvs_amd import benchmarks
        del models[:], streams[:]
        resident.clear()
        torch.cuda.empty_cache()
        other = {}
        for name, (v_, h_, w_, it_, ft_) in {"cfg3_fp32": (5, 1152, 1600, 4, "fp32"), "cfg3_fp16": (5, 1152, 1600, 4, "fp16"),
                                             "cfg5_fp16": (11, 1280, 1920, 8, "fp16")}.items():
            try:
                other[name] = benchmarks.shape_leg(dev, v_, h_, w_, it_, ft_, warmup=5, steps=10, conv_arithmetic=args.conv_arithmetic)
            except Exception as e:  # noqa: BLE001  (a leg that fails must not take the headline line with it)
                other[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
        for name, graph in (("train_step_cfg4", False), ("train_step_cfg4_graph", True)):      # eager, and the step as one hipGraph
            try:
                t = benchmarks.train_step_leg(dev, batch=4, feature_dtype="bf16", regress=True, warmup=3, steps=5, graph=graph)
                t.pop("_step", None)
                other[name] = t
            except Exception as e:  # noqa: BLE001
                other[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            torch.cuda.empty_cache()

    if rank == 0:
        # the probe once more after everything timed above: the chip's clocks drift with its thermal state within a run
        box_end = _bm.box_probe(dev)
        box_start = box
        box = {k: 0.5 * (box_start[k] + box_end[k]) for k in box_start}
        box["sclk_workload_MHz"] = sclk_workload
        result = {
            "metric": f"depth-maps/sec (ref-views/s) at {args.views}-view {args.width}x{args.height}, {args.iters} iters",
            "value": value, "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "repeats": args.repeats, "ms_per_step_min": min(regions) / args.steps * 1e3,
            "ms_per_step_max": max(regions) / args.steps * 1e3,
            "ms_per_step_regions": [round(r / args.steps * 1e3, 5) for r in regions],     # every timed region, in order (`value` = their median)
            # SURVEY 8(d) / eval.py:130-137 form of the same metric: pinned host buffers in (uint8 images as decoded from
            # disk, cameras, depth range), pinned host buffers out (depth + confidence), copies overlapped with compute
            "value_with_transfers": None,
            "vs_baseline": None, "dtype": "f32" if args.feature_dtype == "fp32" else f"f32 arithmetic, {args.feature_dtype} feature storage",
            "data": "synthetic",
            "config": {"workload": f"{workload_name(args.views, args.height, args.width, args.iters)}: 1 ref + {s_views} src views, "
                                   f"{args.width}x{args.height}, {args.iters} GRU iterations, test mode, "
                                   f"random-init weights, {args.batch} ref view(s) per step and GPU",
                       "views": args.views, "height": args.height, "width": args.width, "iterations": args.iters,
                       "value_is": "inputs resident in HBM; `value_with_transfers` = host buffers in / out",
                       "batch_per_gpu": args.batch, "streams_per_gpu": args.streams, "feature_dtype": args.feature_dtype,
                       "launch": "eager" if args.eager else "one hipGraph per depth map",
                       "parallelism": f"ref-view sharding x{world}, no collective",
                       "projection": args.projection, "conv_arithmetic": args.conv_arithmetic,
                       "side_branch": args.side_branch,
                       "process_group": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                       "cpu_affinity": affinity,
                       "algorithmic_MB_per_depth_map": b_map / 1e6},
            "ms_per_step_ranks": {"min": min(rank_ms), "max": max(rank_ms)},   # each rank's own median region (straggler check)
            # this box against the pool (benchmarks.POOL_MEDIAN): `value` x (pool median / this box), per probe
            "box": box, "box_start_end": [box_start, box_end], "value_normalised": _bm.normalised(value, box, roofline["avg_launch_ms"] if roofline and (args.views, args.height, args.width, args.iters, args.feature_dtype, args.batch) == (5, 512, 640, 4, "fp32", 1) else None),
            "box_pool_median": _bm.POOL_MEDIAN,
            "roofline": roofline,
            "other_configs": other,
            "roofline_conv": conv_roofline,
            "staged_inputs": staged,
            "pipelined": pipelined,
            "with_transfers": with_transfers,
        }
        if with_transfers is not None:
            result["value_with_transfers"] = with_transfers.get("uint8_images", with_transfers)["value"]
        if world == 1 and not args.no_cpu_baseline:
            shard.restore_affinity()                   # the CPU path gets the whole host, not the GPU's NUMA node
            result["cpu_baseline"] = cpu_baseline(args)
            result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
        elif world > 1:
            # (the CPU port, the other BASELINE shapes and the training legs are timed on rank 0 of a 1-GPU run only)
            result["cpu_baseline"] = "world>1: see the N=1 line"
            result["other_configs"] = "world>1: see the N=1 line"
        # the driver keeps the tail of stdout: 6 significant digits on everything but the headline numbers keeps the line short
        keep = {k: result[k] for k in ("value", "ms_per_step")}
        result = _round_floats(result)
        result.update(keep)
        print(json.dumps(result), flush=True)
    shard.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
