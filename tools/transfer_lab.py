#!/usr/bin/env python3
"""Where does the host-buffers-in / host-buffers-out form of the metric lose its time?  (GPU box only)

Runs the cfg-1 depth map with uint8 images from pinned host memory in several choreographies and prints ms per depth map
for each, so the cost of each ingredient (H2D, D2H, cross-stream events, Python) is visible in one table:

    resident        two alternating graph runners on one stream, no copies            (= bench `value`)
    events_only     the three-stream event choreography of bench.transfers_leg with the copies removed
    h2d_only        + H2D (uint8 images, cameras, depth range) and the GPU pyramid kernel
    d2h_only        + D2H of depth and confidence
    full            both                                                               (= bench `with_transfers.uint8_images`)
    full_pyr_cmp    like full, but the uint8 -> float pyramid kernel runs on the compute stream in front of the replay
    serial          everything on ONE stream, no events: H2D -> pyramid -> replay -> D2H
    one_graph       (only with --one-graph) ONE captured graph per runner: replay || upload of the next sample, download as the tail
    abl_*           resident + one ingredient: an event record; a record another stream waits on; an unordered upload; a timing event
    host_sync       like full, but no cross-stream wait on the compute stream: the host waits for the (long finished) upload
    host_driven     no cross-stream wait anywhere: the host waits for replay i-1, then issues its download and the next upload
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import ops, synthetic  # noqa: E402
from itermvs_amd.engine import GraphedRunner, InferenceEngine  # noqa: E402
from itermvs_amd.net import Pipeline  # noqa: E402

H, W, V, IT = 512, 640, 5, 4
dev = torch.device("cuda")
m = Pipeline(iteration=IT, test=True)
m.load_state_dict(synthetic.random_state_dict(0))
m = m.to(dev).eval()
eng = InferenceEngine(m.weights(), IT)
samples = [synthetic.make_sample(1, V, H, W, seed=i) for i in range(4)]
s0 = samples[0]
pj0 = {l: s0["proj_matrices"][f"level_{l}"].float().to(dev) for l in (1, 2, 3)}
runners = [GraphedRunner(eng, s0["imgs"]["level_0"].float().to(dev), pj0, s0["depth_min"].float().to(dev),
                         s0["depth_max"].float().to(dev)) for _ in range(2)]
host_in = []
for s in samples:
    img = s["imgs"]["level_0"].float()
    u8 = ((img[0].permute(0, 2, 3, 1) + 1) * 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
    host_in.append((u8.pin_memory(), torch.stack([s["proj_matrices"][f"level_{l}"].float() for l in (1, 2, 3)]).pin_memory(),
                    s["depth_min"].float().pin_memory(), s["depth_max"].float().pin_memory()))
host_out = [tuple(torch.empty(o.shape, dtype=o.dtype).pin_memory() for o in r.out) for r in runners]
raw_dev = [torch.empty(host_in[0][0].shape, dtype=torch.uint8, device=dev) for _ in runners]
s_in, s_cmp, s_out = (torch.cuda.Stream(device=dev) for _ in range(3))


def timed(step, n=200, warm=20):
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(warm + i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3


def make_three_stream(h2d: bool, d2h: bool, pyramid_on_compute: bool = False):
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]
    started = [False, False]

    def step(i):
        k = i % 2
        r = runners[k]
        h_img, h_proj, h_min, h_max = host_in[i % len(host_in)]
        with torch.cuda.stream(s_in):
            if started[k]:
                s_in.wait_event(ev_done[k])
            if h2d:
                raw_dev[k].copy_(h_img, non_blocking=True)
                if not pyramid_on_compute:
                    ops.image_pyramid(raw_dev[k], H, W, all_levels=False, out0=r.imgs[0])
                r.proj_stack.copy_(h_proj, non_blocking=True)
                r.depth_min.copy_(h_min, non_blocking=True)
                r.depth_max.copy_(h_max, non_blocking=True)
            ev_in[k].record(s_in)
        with torch.cuda.stream(s_cmp):
            s_cmp.wait_event(ev_in[k])
            if started[k]:
                s_cmp.wait_event(ev_out[k])
            if h2d and pyramid_on_compute:      # only SDMA copies run beside the graph; the normalise / resize kernel is in line
                ops.image_pyramid(raw_dev[k], H, W, all_levels=False, out0=r.imgs[0])
            r(r.imgs, r.projs, r.depth_min, r.depth_max)
            ev_done[k].record(s_cmp)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_done[k])
            if d2h:
                for h, d in zip(host_out[k], r.out):
                    h.copy_(d, non_blocking=True)
            ev_out[k].record(s_out)
        started[k] = True
    return step


def make_host_sync():
    """like full, but the COMPUTE stream never waits on another stream: the host waits for the upload of sample i (issued
    one step earlier, long done) and for the download of replay i-2 before it enqueues replay i; only the copy streams wait
    on the compute stream's events"""
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]
    state = {"n": 0}

    def upload(i):
        k = i % 2
        r = runners[k]
        h_img, h_proj, h_min, h_max = host_in[i % len(host_in)]
        with torch.cuda.stream(s_in):
            if i >= 2:
                s_in.wait_event(ev_done[k])        # replay i-2 has consumed these static inputs
            raw_dev[k].copy_(h_img, non_blocking=True)
            ops.image_pyramid(raw_dev[k], H, W, all_levels=False, out0=r.imgs[0])
            r.proj_stack.copy_(h_proj, non_blocking=True)
            r.depth_min.copy_(h_min, non_blocking=True)
            r.depth_max.copy_(h_max, non_blocking=True)
            ev_in[k].record(s_in)

    def step(i):
        k = i % 2
        r = runners[k]
        if state["n"] == 0:
            upload(i)
        state["n"] += 1
        upload(i + 1)                              # the next sample starts travelling now
        ev_in[k].synchronize()                     # host: sample i is on the device
        if i >= 2:
            ev_out[k].synchronize()                # host: outputs of replay i-2 are on the host
        with torch.cuda.stream(s_cmp):
            r(r.imgs, r.projs, r.depth_min, r.depth_max)
            ev_done[k].record(s_cmp)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_done[k])
            for h, d in zip(host_out[k], r.out):
                h.copy_(d, non_blocking=True)
            ev_out[k].record(s_out)
    return step


_abl_ev = [torch.cuda.Event() for _ in range(4)]


def abl_record(i):              # resident + one event record on the compute stream after the replay
    r = runners[i % 2]
    with torch.cuda.stream(s_cmp):
        r(r.imgs, r.projs, r.depth_min, r.depth_max)
        _abl_ev[i % 2].record(s_cmp)


def abl_record_wait(i):         # ... + another stream waits on it and records its own event (no copies anywhere)
    abl_record(i)
    with torch.cuda.stream(s_out):
        s_out.wait_event(_abl_ev[i % 2])
        _abl_ev[2 + i % 2].record(s_out)


def abl_h2d_noevents(i):        # resident + the uploads on another stream with NO ordering at all (timing only)
    k = i % 2
    h_img, h_proj, h_min, h_max = host_in[i % len(host_in)]
    with torch.cuda.stream(s_in):
        raw_dev[k].copy_(h_img, non_blocking=True)
    resident(i)


def abl_timing_event(i):        # resident + a TIMING event record (what torch.cuda.Event(enable_timing=True) costs)
    r = runners[i % 2]
    with torch.cuda.stream(s_cmp):
        r(r.imgs, r.projs, r.depth_min, r.depth_max)
        torch.cuda.Event(enable_timing=True).record(s_cmp)


def make_host_driven():
    """NO cross-stream wait at all: the host orders the copies.  Per step: enqueue replay i (its inputs were confirmed on
    the device one step earlier); wait for replay i-1; only then issue its download and the upload of sample i+1 into the
    runner that replay i-1 has just released; wait for both.  The compute stream always has the next replay queued ~0.7 ms
    ahead and carries one event record per replay."""
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]
    state = {"primed": False}

    def upload(i):
        k = i % 2
        r = runners[k]
        h_img, h_proj, h_min, h_max = host_in[i % len(host_in)]
        with torch.cuda.stream(s_in):
            raw_dev[k].copy_(h_img, non_blocking=True)
            ops.image_pyramid(raw_dev[k], H, W, all_levels=False, out0=r.imgs[0])
            r.proj_stack.copy_(h_proj, non_blocking=True)
            r.depth_min.copy_(h_min, non_blocking=True)
            r.depth_max.copy_(h_max, non_blocking=True)
            ev_in[k].record(s_in)

    def step(i):
        k = i % 2
        r = runners[k]
        if not state["primed"]:
            torch.cuda.synchronize()
            upload(i)
            ev_in[k].synchronize()
            state["primed"] = True
            state["prev"] = None
        with torch.cuda.stream(s_cmp):
            r(r.imgs, r.projs, r.depth_min, r.depth_max)
            ev_done[k].record(s_cmp)
        p = state["prev"]
        if p is not None:
            ev_done[p].synchronize()               # replay i-1 is done (replay i runs now)
            with torch.cuda.stream(s_out):
                for h, d in zip(host_out[p], runners[p].out):
                    h.copy_(d, non_blocking=True)
                ev_out[p].record(s_out)
        upload(i + 1)                              # into runner (i+1) % 2: released by the wait above (first step: never used yet)
        ev_in[(i + 1) % 2].synchronize()
        if p is not None:
            ev_out[p].synchronize()
        state["prev"] = k
    return step


def resident(i):
    r = runners[i % 2]
    with torch.cuda.stream(s_cmp):
        r(r.imgs, r.projs, r.depth_min, r.depth_max)


def serial(i):
    k = i % 2
    r = runners[k]
    h_img, h_proj, h_min, h_max = host_in[i % len(host_in)]
    with torch.cuda.stream(s_cmp):
        raw_dev[k].copy_(h_img, non_blocking=True)
        ops.image_pyramid(raw_dev[k], H, W, all_levels=False, out0=r.imgs[0])
        r.proj_stack.copy_(h_proj, non_blocking=True)
        r.depth_min.copy_(h_min, non_blocking=True)
        r.depth_max.copy_(h_max, non_blocking=True)
        r(r.imgs, r.projs, r.depth_min, r.depth_max)
        for h, d in zip(host_out[k], r.out):
            h.copy_(d, non_blocking=True)


# ---- everything in ONE captured graph per runner: [replay of runner k] || [H2D of the NEXT sample into runner 1-k's static
# inputs + pyramid kernel], then D2H of runner k's outputs; one graph launch per step on one stream, no events on the host side
FILL_STAGING = "--fill-staging" in sys.argv
stage_in = [tuple(t.clone().pin_memory() for t in host_in[0]) for _ in range(2)]      # fixed pinned staging buffers (captured addresses)
big_graphs = []


def build_big_graphs():
    cap = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)
    for k in range(2):
        r, o = runners[k], runners[1 - k]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            torch.cuda.synchronize()
            g.capture_begin()
            side.wait_stream(cap)
            with torch.cuda.stream(side):          # upload of the next sample beside the compute chain
                h_img, h_proj, h_min, h_max = stage_in[1 - k]
                raw_dev[1 - k].copy_(h_img, non_blocking=True)
                ops.image_pyramid(raw_dev[1 - k], H, W, all_levels=False, out0=o.imgs[0])
                o.proj_stack.copy_(h_proj, non_blocking=True)
                o.depth_min.copy_(h_min, non_blocking=True)
                o.depth_max.copy_(h_max, non_blocking=True)
            out = eng.run(r.imgs, r.proj_stack, r.depth_min, r.depth_max)
            for h, d in zip(host_out[k], out):
                h.copy_(d, non_blocking=True)
            cap.wait_stream(side)
            g.capture_end()
        big_graphs.append(g)


def one_graph(i):
    k = i % 2
    src = host_in[(i + 1) % len(host_in)]
    if FILL_STAGING:                               # the loader's job: the next sample lands in the pinned staging buffer
        for d, t in zip(stage_in[1 - k], src):     # (a single-threaded 4.9 MB torch copy costs ~3 ms of host time: off by default)
            d.copy_(t)
    with torch.cuda.stream(s_cmp):
        big_graphs[k].replay()


have_big = False
if "--one-graph" in sys.argv:
    try:
        eng._ws_owner = "lab"
        build_big_graphs()
        eng._ws_owner = None
        have_big = True
    except Exception as e:  # noqa: BLE001
        print("one-graph variant not capturable here:", repr(e)[:300])

rows = [("resident", resident), ("events_only", make_three_stream(False, False)), ("h2d_only", make_three_stream(True, False)),
        ("d2h_only", make_three_stream(False, True)), ("full", make_three_stream(True, True)),
        ("full_pyr_cmp", make_three_stream(True, True, True)), ("serial", serial), ("resident", resident), ("full", make_three_stream(True, True))]
if have_big:
    rows += [("one_graph", one_graph)]
rows += [("host_driven", make_host_driven()), ("resident", resident), ("host_driven", make_host_driven()), ("abl_record", abl_record), ("abl_rec_wait", abl_record_wait), ("abl_h2d_noev", abl_h2d_noevents), ("abl_timing_ev", abl_timing_event),
         ("resident", resident), ("host_sync", make_host_sync()), ("full", make_three_stream(True, True)), ("host_sync", make_host_sync())]
print(f"{'choreography':<14} {'ms/map':>8} {'host ms/step':>13} {'maps/s':>8}")
for name, fn in rows:
    ms, host = timed(fn)
    print(f"{name:<14} {ms:8.3f} {host:13.3f} {1e3 / ms:8.1f}", flush=True)
eng.check_projection_finite()
