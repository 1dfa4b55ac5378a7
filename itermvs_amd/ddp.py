"""Data-parallel training glue: one process per GPU, ONE flat-bucket gradient all-reduce per step.

The reference trains with single-process ``nn.DataParallel`` (train.py:95: replicate / scatter /
gather through GPU 0 every step).  On MI355X the natural form is one process per GPU with RCCL over
xGMI: the model has 343 685 parameters (1.37 MB of fp32 gradients), so a single bucket holds every
gradient and the collective is latency-bound (tens of microseconds next to a step of tens of
milliseconds) -- no overlap machinery, no bucketing heuristics, nothing to tune.
Parameters that never receive a gradient (``feature_net.inner3.*`` always; ``confidence_head`` without
``--regress``) are the same on every rank, so skipping them keeps ranks in step.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def flat_allreduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int = None, force: bool = False) -> int:
    """Average ``p.grad`` over all ranks with one all-reduce over one flat buffer.
    Returns the number of gradient elements reduced (0 when not running distributed, or in a world of one rank unless
    ``force`` -- the 1-rank RCCL test runs the collective anyway)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = world_size or dist.get_world_size()
    grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
    if not grads or (world == 1 and not force):
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    if flat.is_cuda and dist.get_backend() == "gloo":
        # test rigs without RCCL between the ranks (two processes on one GPU): stage the 1.37 MB bucket through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return off


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s weights and BatchNorm statistics."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    via_host = dist.get_backend() == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if via_host and t.is_cuda:
            host = t.data.cpu()
            dist.broadcast(host, src)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src)
