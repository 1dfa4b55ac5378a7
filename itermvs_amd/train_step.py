"""The training step of reference train.py:194-243 (``train_sample``) as ONE hipGraph.

Eagerly the step is ~2 800 kernel launches driven from Python (MIOpen / ATen convolutions and element-wise kernels around
the fused correlation Functions): 26 ms at B = 1 and still launch-bound at train_dtu.sh's B = 4.  Every piece is
capturable -- the forward builds its hypotheses in-kernel, ``full_loss`` uses masked sums instead of boolean indexing, the NaN
assert on the projections is deferred to a device flag, gradient clipping and Adam (``capturable=True``) are tensor programs
-- so forward + loss + backward + gradient all-reduce + clip + optimizer step are recorded once into a
``torch.cuda.CUDAGraph`` with static input buffers and replayed per batch.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

from . import ddp
from .net import full_loss

Tensor = torch.Tensor


def _flatten(batch) -> Sequence[Tensor]:
    imgs, projs, dmin, dmax, gt, mask = batch
    return [imgs[k] for k in sorted(imgs)] + [projs[k] for k in sorted(projs)] + [dmin, dmax] + \
           [gt[k] for k in sorted(gt)] + [mask[k] for k in sorted(mask)]


class CapturedTrainStep:
    """``step(batch) -> (loss, abs depth error)`` tensors (device scalars, valid in stream order).

    The first ``warmup`` calls run the step eagerly on a side stream (real optimisation steps: MIOpen picks its kernels, the
    caching allocator and Adam's state settle); the next call captures the graph with that batch as the static inputs and
    replays it; later calls copy their batch into the static buffers and replay.  ``batch`` = (imgs, proj_matrices,
    depth_min, depth_max, depth_gt, mask) like ``train.synthetic_batch`` / the reference's collated sample, always of the
    same shapes.  The optimizer must be created with ``capturable=True`` (its step counter lives on the device); learning
    rate schedules keep working when the rate is a device tensor (``lr=torch.tensor(...)``), which torch's schedulers fill
    in place.  ``check()`` raises the deferred NaN assert of module.py:83,87 (one 4-byte read-back); the flagged step itself
    was a no-op on the weights (see ``_step``).  The flag is shared with every forward of the model: call ``check()`` after
    validation forwards too, so a bad validation sample is not attributed to the next training step."""

    def __init__(self, model, optimizer, regress: bool, clip: float = 2.0, warmup: int = 3):
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("CapturedTrainStep needs an optimizer created with capturable=True")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() != "nccl":
            raise ValueError("CapturedTrainStep: the gradient all-reduce is captured with the step and needs the RCCL (nccl) backend")
        self.model, self.opt, self.regress, self.clip = model, optimizer, regress, clip
        self.warmup = warmup
        self.calls = 0
        self.graph = None
        self.static = None
        self.out: Tuple[Tensor, Tensor] = None
        self.params = [p for p in model.parameters()]
        dev = self.params[0].device
        self.nan_flag = torch.zeros((1,), device=dev, dtype=torch.int32)
        model.train_nan_flag = self.nan_flag
        self.stream = torch.cuda.Stream(device=dev)

    def _step(self, batch):
        imgs, projs, dmin, dmax, gt, mask = batch
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        out = self.model(imgs, projs, dmin, dmax)
        loss = full_loss(out["depths"], out["depths_upsampled"], out["confidences"], gt, mask, dmin, dmax, self.regress)
        loss.backward()
        ddp.flat_allreduce_gradients(self.params)
        torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        # The reference asserts on a NaN projection BEFORE any update (module.py:83,87); here the assert is deferred to
        # ``check()``, so a flagged step must leave the weights alone: its (NaN) gradients are replaced by zeros and the
        # learning rate of this one step by 0 -- the parameters do not move, Adam's moments only decay by their betas
        # (they stay finite), and ``check()`` raises afterwards on a model that can still be saved or resumed.
        ok = self.nan_flag == 0
        zero = torch.zeros((), device=ok.device)
        for p in self.params:
            if p.grad is not None:
                p.grad = torch.where(ok, p.grad, zero)
        rates = [(g, g["lr"].clone()) for g in self.opt.param_groups if torch.is_tensor(g["lr"])]
        for g, _ in rates:
            g["lr"].mul_(ok.to(g["lr"].dtype).reshape(g["lr"].shape))
        self.opt.step()
        for g, saved in rates:
            g["lr"].copy_(saved)
        err = (out["depths_upsampled"][0].detach() - gt["level_0"]).abs().mean()
        return loss.detach(), err

    def step(self, batch) -> Tuple[Tensor, Tensor]:
        self.calls += 1
        cur = torch.cuda.current_stream()
        if self.graph is None and self.calls <= self.warmup:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self._step(batch)
            cur.wait_stream(self.stream)
            return out
        if self.graph is None:
            self.static = tuple({k: v.clone() for k, v in part.items()} if isinstance(part, dict) else part.clone() for part in batch)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):   # data-loader threads may pin memory meanwhile
                    self.out = self._step(self.static)
            cur.wait_stream(self.stream)
        else:
            for dst, src in zip(_flatten(self.static), _flatten(batch)):
                if dst.shape != src.shape:
                    raise RuntimeError(f"CapturedTrainStep: batch tensor of shape {tuple(src.shape)}, captured {tuple(dst.shape)}")
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.out

    def check(self) -> None:
        if int(self.nan_flag.item()):
            self.nan_flag.zero_()
            raise AssertionError("nan in proj (singular or non-finite camera matrix, module.py:83,87)")
