#!/usr/bin/env python3
"""Do the depth head (matrix-core bound) and the fused correlation kernel (vector-ALU / vector-L1 bound) overlap when they
run SIDE BY SIDE on one MI355X?  They are consecutive, dependent launches of a GRU iteration (head -> corr_iter, 20 + 22 us
of a 108 us iteration), so a fused "head, then the correlations of the same pixel tile" kernel could hide one behind the
other -- if the two do not contend for the same pipe.  This probe times, at cfg 1 on the engine's own buffers:
    N x head back to back,  N x corr_iter back to back,  and both chains concurrently on two streams (one hipGraph with a
    fork and a join around the N-launch chains),
and prints the three times: concurrent ~ max(a, b) means they overlap, ~ a + b means they serialise.

    python tools/overlap_probe.py [--n 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import ops, synthetic  # noqa: E402
from itermvs_amd.engine import HIDDEN, InferenceEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    dev = torch.device("cuda")
    w = {k: v.to(dev) for k, v in synthetic.random_state_dict(0).items()}
    eng = InferenceEngine(w, 4, "fp32")
    s = synthetic.make_sample(1, 5, 512, 640, seed=0)
    imgs = s["imgs"]["level_0"].to(dev)
    projs = {l: s["proj_matrices"][f"level_{l}"].to(dev) for l in (1, 2, 3)}
    dmin, dmax = s["depth_min"].to(dev), s["depth_max"].to(dev)
    # one ordinary run leaves a consistent workspace behind (hidden state, normalised depth, view weights)
    b, v = 1, 5
    feats = eng.feature_net(imgs.reshape(b * v, 3, 512, 640).contiguous())
    per_view = {l: feats[l].view(b, v, *feats[l].shape[1:]) for l in (1, 2, 3)}
    src = {l: [per_view[l][:, i] for i in range(1, v)] for l in (1, 2, 3)}
    ref = {l: per_view[l][:, 0] for l in (1, 2, 3)}
    ws = eng._workspace(b, *feats[2].shape[2:])
    pstack = torch.stack([projs[1], projs[2], projs[3]])
    ref_q, proj, inv_min, inv_max = ops.ref_quarter_compose(ref[1], ref[2], ref[3], pstack.reshape(3 * b, v, 4, 4), eng.nan_flag, (dmin, dmax))
    proj = proj.view(3, b, v - 1, 12)
    view_w = eng.stage_init(ws, src[3], ref[3], proj[2], inv_min, inv_max, None)
    eng.stage_head(ws)
    torch.cuda.synchronize()

    head = lambda: eng.stage_head(ws)
    corr = lambda: eng.stage_corr(ws, src, ref_q, proj, view_w, inv_min, inv_max, timed=False)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def graph_of(fa, fb):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(sa):
            for f in (fa, fb):
                if f:
                    f()
            torch.cuda.synchronize()
            g.capture_begin()
            if fb:
                sb.wait_stream(sa)
                with torch.cuda.stream(sb):
                    for _ in range(args.n):
                        fb()
            if fa:
                for _ in range(args.n):
                    fa()
            if fb:
                sa.wait_stream(sb)
            g.capture_end()
        return g

    def time_graph(g):
        best = 1e9
        with torch.cuda.stream(sa):
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / args.n)
        return best

    # the same chains as two SEPARATE graphs replayed on two streams (different hardware queues): what the machine can
    # overlap when nothing serialises the launches
    def two_streams(ga, gb):
        best = 1e9
        for _ in range(args.reps):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(sa):
                e0.record()
            sb.wait_stream(sa)
            with torch.cuda.stream(sa):
                ga.replay()
                e1.record()
            with torch.cuda.stream(sb):
                gb.replay()
                e2.record()
            torch.cuda.synchronize()
            best = min(best, max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3 / args.n)
        return best

    def graph_on(stream, f):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            f()
            torch.cuda.synchronize()
            g.capture_begin()
            for _ in range(args.n):
                f()
            g.capture_end()
        return g

    t_head = time_graph(graph_of(head, None))
    t_corr = time_graph(graph_of(corr, None))
    t_both = time_graph(graph_of(head, corr))
    t_two = two_streams(graph_on(sa, head), graph_on(sb, corr))
    print(f"two separate graphs on two streams: {t_two:.2f} us per launch pair")
    print(f"per launch pair (us), {args.n} launches per chain: head alone {t_head:.2f}, corr_iter alone {t_corr:.2f}, "
          f"both chains concurrently {t_both:.2f}  (sum {t_head + t_corr:.2f}, max {max(t_head, t_corr):.2f})")


if __name__ == "__main__":
    main()
