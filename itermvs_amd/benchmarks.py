"""Measurement legs shared by bench.py and the tools/ scripts (GPU box only): one depth map per hipGraph replay at any of the
BASELINE shapes, and the cfg-4 training step.  Nothing here is on the product path."""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(s: int, h: int, w: int, batch: int, iters: int, e: int = 4):
    """DESIGN.md 'Algorithmic bytes' (hypotheses built in-kernel): every required tensor moved once; ``e`` =
    bytes per stored feature element (4, or 2 with bf16 / fp16 feature storage), everything else fp32.
    Returns (bytes per corr_iter launch, bytes per corr_init launch, bytes per depth map)."""
    p1, p2, p3 = (h // 2) * (w // 2), (h // 4) * (w // 4), (h // 8) * (w // 8)
    it = batch * (s * (16 * p1 + 32 * p2 + 48 * p3) * e     # source pyramids, each view once
                  + 96 * p2 * 4                             # packed reference features at 1/4 res (kept fp32)
                  + p2 * 4                                  # normalised depth
                  + s * p2 * 4                              # view weights
                  + 80 * p2 * 4)                            # [B,10,8,H/4,W/4] aggregated correlations out
    init = batch * (s * 48 * p3 * e + 48 * p3 * e           # level-3 source + reference features
                    + s * 8 * 32 * p3 * 4)                  # per-view correlation volume out
    return it, init, (init + iters * it) / batch


# Pool reference of the box probe: medians over the fresh MI355X boxes this repository was measured on (profiles/r06_box_probe.md
# lists every box).  `value_normalised` = value x (POOL_MEDIAN / this box), one figure per probe -- `value` itself is never touched.
POOL_MEDIAN = {"mfma_f32_tflops": None, "copy_GBps": None, "sclk_MHz": None, "graph_node_us": None, "l2_latency_ns": None,
               "hbm_latency_ns": None, "sclk_idle_MHz": None, "sclk_workload_MHz": None}
# The yardstick that DOES track a box's speed (profiles/r06_box_probe.md): the event-bracketed launch time of itermvs_corr_iter inside
# the timed region (`roofline.avg_launch_ms`).  That kernel's code has not changed since round 4, every kernel of a depth map scales
# with it from box to box, and value x launch time is constant within 0.5 % over boxes that differ by 6 % in value.  Median over
# the ten boxes of round 6, cfg 1, fp32 feature storage:
POOL_MEDIAN_CORR_ITER_MS = 0.0262
# probes where a SMALLER figure means a faster box (value_normalised multiplies by box / pool instead of pool / box)
LOWER_IS_FASTER = {"graph_node_us", "l2_latency_ns", "hbm_latency_ns"}


def box_probe(dev, repeats: int = 5) -> Dict[str, float]:
    """Two fixed micro-kernels that say what THIS box (chip, clocks, HBM stack) delivers, run before the timed regions
    (<= 0.3 s together): (1) itermvs_box_probe -- 2048 workgroups x 4 waves x 4 independent fp32 MFMAs, 3000 rounds: sustained
    fp32-MFMA TFLOP/s and the shader clock it ran at (s_memtime ticks per 100 MHz s_memrealtime tick); (2) a 256 MB float4
    stream copy (itermvs_copy_multi: read + write = 512 MB of traffic).  Best of ``repeats`` event-bracketed launches each."""
    from . import ops
    blocks, iters = 2048, 3000
    sink = torch.empty((blocks * 256,), device=dev)
    clocks = torch.zeros((2,), device=dev, dtype=torch.int64)
    src = torch.empty((64 * 1024 * 1024,), device=dev)            # 256 MB
    src.normal_()
    dst = torch.empty_like(src)
    ops.box_probe(sink, blocks, 50, clocks)
    ops.copy_multi([dst], [src])
    torch.cuda.synchronize(dev)
    best_m, best_c, mhz = None, None, 0.0
    for _ in range(repeats):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ops.box_probe(sink, blocks, iters, clocks)
        e1.record()
        ops.copy_multi([dst], [src])
        e2.record()
        torch.cuda.synchronize(dev)
        tm, tc = e0.elapsed_time(e1), e1.elapsed_time(e2)
        if best_m is None or tm < best_m:
            best_m = tm
            c = clocks.tolist()
            mhz = c[0] / max(c[1], 1) * 100.0
        best_c = tc if best_c is None else min(best_c, tc)
    flops = 2.0 * 16 * 16 * 4 * 4 * iters * blocks * 4
    out = {"mfma_f32_tflops": flops / (best_m * 1e-3) / 1e12, "copy_GBps": 2.0 * src.numel() * 4 / (best_c * 1e-3) / 1e9,
           "sclk_MHz": mhz}
    # (3) what a kernel boundary costs inside a hipGraph: 200 dependent one-workgroup launches per replay (a depth map is ~50
    # nodes); (4) the latency of a dependent load, ring of 1 MB (L2) and of 512 MB (HBM): the gather kernels wait on these
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        ops.box_probe(sink, 1, 1)
        torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        gr.capture_begin()
        for _ in range(200):
            ops.box_probe(sink, 1, 1)
        gr.capture_end()
        best_g = None
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize(dev)
            best_g = e0.elapsed_time(e1) if best_g is None else min(best_g, e0.elapsed_time(e1))
    out["graph_node_us"] = best_g / 200 * 1e3
    del gr
    for name, n_lines in (("l2_latency_ns", 8192), ("hbm_latency_ns", 3 * 1024 * 1024)):
        order = torch.randperm(n_lines, device=dev)
        ring = torch.zeros((n_lines * 32,), dtype=torch.int32, device=dev)            # one index per 128-byte line
        ring[order * 32] = (order.roll(-1) * 32).to(torch.int32)
        _, _, at = ops.box_chase(ring, 2000)
        best = None
        for _ in range(3):                      # every walk continues where the last one ended: lines not touched before
            ns, mhz_idle, at = ops.box_chase(ring, 4000, at)
            best = ns if best is None else min(best, ns)
        out[name] = best
        out["sclk_idle_MHz"] = mhz_idle
        del ring
    del sink, src, dst
    torch.cuda.empty_cache()
    return out


def workload_clock(dev, replay, n: int = 20) -> Optional[float]:
    """shader clock (MHz) the chip sustains while ``replay()`` -- one depth map's hipGraph -- runs ``n`` times: two one-lane
    stamps of (100 MHz counter, shader-clock counter) bracket the replays on the stream.  The boxes of the pool differ by up to
    6 % in depth-maps/s with equal matrix-pipe and copy rates under sustained load (profiles/r06_box_probe.md): the clock a chip
    holds under THIS bursty, latency-bound workload is what differs.
    The ratio is only a clock when both counters ran through the bracket undisturbed: on some boxes of the pool it comes out at
    3 400 or 10 000 "MHz" (above anything the chip can clock, profiles/r06_box_probe.md) -- such a figure is reported as None."""
    from . import ops
    a, b = torch.zeros((16, 2), device=dev, dtype=torch.int64), torch.zeros((16, 2), device=dev, dtype=torch.int64)
    replay()
    torch.cuda.synchronize(dev)
    ops.clock_stamp(a)
    for _ in range(n):
        replay()
    ops.clock_stamp(b)
    torch.cuda.synchronize(dev)
    a, b = a.tolist(), b.tolist()
    # per XCD (the counters of different XCDs are not aligned): slots both stamps filled
    mhz = sorted(float(b[i][1] - a[i][1]) / (b[i][0] - a[i][0]) * 100.0 for i in range(16) if a[i][0] and b[i][0] > a[i][0])
    med = mhz[len(mhz) // 2] if mhz else None
    return med if med is not None and 500.0 <= med <= 2600.0 else None


def normalised(value: float, box: Dict[str, float], corr_iter_ms: Optional[float] = None) -> Optional[Dict[str, float]]:
    """``value`` scaled to the pool-median box: by the launch time of the unchanged corr_iter kernel (cfg 1 / fp32 only), and per
    probe where a pool median is filled in"""
    out = {}
    if corr_iter_ms:
        out["by_corr_iter_launch"] = value * corr_iter_ms / POOL_MEDIAN_CORR_ITER_MS
    for k, ref in POOL_MEDIAN.items():
        if ref and box.get(k):
            out["by_" + k] = value * (box[k] / ref if k in LOWER_IS_FASTER else ref / box[k])
    return out or None


def shape_leg(dev, views: int, height: int, width: int, iters: int, feature_dtype: str = "fp32", warmup: int = 5,
              steps: int = 10, conv_arithmetic: str = "bf16x3") -> Dict[str, object]:
    """depth-maps/s of ONE captured hipGraph (one reference view per replay, inputs resident) at another shape than the
    headline's, with HIP-event brackets around every fused correlation launch inside the graph:
    ``warmup`` + ``steps`` replays -> value, ms per depth map, mean corr_iter / corr_init launch time and their fraction of
    the HBM roof on the algorithmic bytes of that shape."""
    from . import ops, synthetic
    from .engine import GraphedRunner, InferenceEngine
    from .net import Pipeline
    m = Pipeline(iteration=iters, test=True)
    m.load_state_dict(synthetic.random_state_dict(0))
    m = m.to(dev).eval()
    eng = InferenceEngine(m.weights(), iters, feature_dtype, conv_arithmetic=conv_arithmetic)
    s = synthetic.make_sample(batch=1, num_views=views, height=height, width=width, seed=0)
    pj = {l: s["proj_matrices"][f"level_{l}"].float().to(dev) for l in (1, 2, 3)}
    per_step = iters + 1
    ops.profile_enable((warmup + steps + 4) * per_step + 8, mask=0x3)
    r = GraphedRunner(eng, s["imgs"]["level_0"].float().to(dev), pj, s["depth_min"].float().to(dev), s["depth_max"].float().to(dev))
    first, count = r.profile_pairs
    for _ in range(warmup):
        r.replay()
    torch.cuda.synchronize(dev)
    prof = []
    t0 = time.perf_counter()
    for _ in range(steps):
        r.replay()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    for _ in range(3):                                   # the brackets of three more replays, read one by one (each read waits
        r.replay()                                 # for its replay: outside the timed loop)
        prof.extend(ops.profile_graph_read(first, count))
    ops.profile_enable(0)
    eng.check_projection_finite()
    e = 4 if feature_dtype == "fp32" else 2
    b_iter, b_init, b_map = algorithmic_bytes(views - 1, height, width, 1, iters, e)
    t_iter = [ms for kind, ms in prof if kind == 1]
    t_init = [ms for kind, ms in prof if kind == 2]
    out = {"views": views, "height": height, "width": width, "iterations": iters, "feature_dtype": feature_dtype,
           "conv_arithmetic": conv_arithmetic, "value": steps / dt, "unit": "depth-maps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup}
    if t_iter:
        ms = sum(t_iter) / len(t_iter)
        out["corr_iter"] = {"avg_launch_ms": ms, "launches_timed": len(t_iter), "frac": b_iter / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if t_init:
        ms = sum(t_init) / len(t_init)
        out["corr_init"] = {"avg_launch_ms": ms, "launches_timed": len(t_init), "frac": b_init / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del r, eng, m
    torch.cuda.empty_cache()
    return out


def train_step_leg(dev, batch: int = 4, views: int = 5, height: int = 512, width: int = 640, iteration: int = 4,
                   feature_dtype: str = "bf16", regress: bool = True, warmup: int = 3, steps: int = 5,
                   phases: bool = False, graph: bool = False) -> Dict[str, object]:
    """BASELINE cfg 4's per-GPU training step (train_dtu.sh: 5 views, 640x512, 4 GRU iterations, ``batch`` per GPU, Adam +
    gradient clip 2.0; reference train.py:194-243): forward (training graph on the fused correlation kernels), full_loss,
    backward, flat gradient all-reduce (a no-op on one rank), clip, Adam -- ``train.train_step`` without its two host
    read-backs inside the timed region.  ``graph``: the same step replayed as ONE hipGraph (train_step.CapturedTrainStep)."""
    from . import ddp, synthetic
    from .net import Pipeline, full_loss
    torch.manual_seed(1)
    model = Pipeline(iteration=iteration, test=False).to(dev)
    model.feature_dtype = feature_dtype
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3, device=dev) if graph else 1e-3, betas=(0.9, 0.999), capturable=graph)
    imgs, projs, dmin, dmax, gt, mask = synthetic.make_training_batch(batch, num_views=views, height=height, width=width)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}  # noqa: E731
    imgs, projs, gt, mask, dmin, dmax = to(imgs), to(projs), to(gt), to(mask), dmin.to(dev), dmax.to(dev)
    params = list(model.parameters())

    def fwd():
        out = model(imgs, projs, dmin, dmax)
        return full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, regress)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = fwd()
        loss.backward()
        ddp.flat_allreduce_gradients(params)
        torch.nn.utils.clip_grad_norm_(params, 2.0)
        opt.step()
        return loss

    if graph:
        from .train_step import CapturedTrainStep
        cap = CapturedTrainStep(model, opt, regress, clip=2.0, warmup=max(1, warmup))
        the_batch = (imgs, projs, dmin, dmax, gt, mask)
        step = lambda: cap.step(the_batch)[0]  # noqa: E731
        for _ in range(max(1, warmup) + 2):     # the eager warm-up steps, the capture, one replay
            loss = step()
        cap.check()
    else:
        for _ in range(warmup):
            loss = step()
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / steps
    res = {"ms_per_step": ms, "samples_per_s": batch * 1e3 / ms, "batch": batch, "views": views, "wh": [width, height],
           "iteration": iteration, "feature_dtype": feature_dtype, "regress": regress, "steps": steps, "warmup": warmup, "graph": graph,
           "loss": float(loss.detach()), "peak_mem_MiB": torch.cuda.max_memory_allocated(dev) / 2 ** 20}
    if phases:
        ph = {"forward": 0.0, "backward": 0.0, "clip+adam": 0.0}
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize(dev); t = time.perf_counter()
            loss = fwd()
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize(dev); t2 = time.perf_counter()
            torch.nn.utils.clip_grad_norm_(params, 2.0)
            opt.step()
            torch.cuda.synchronize(dev); t3 = time.perf_counter()
            ph["forward"] += (t1 - t) * 1e3 / steps
            ph["backward"] += (t2 - t1) * 1e3 / steps
            ph["clip+adam"] += (t3 - t2) * 1e3 / steps
        res["phases_ms"] = ph
    res["_step"] = step              # (tools/train_bench.py --profile reuses the prepared step)
    return res
