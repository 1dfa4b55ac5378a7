"""``Pipeline`` / ``full_loss`` -- the drop-in counterpart of the reference's ``models/net.py``.

Same constructor, ``forward`` signature, output dict keys and state-dict names as
``models.net.Pipeline`` (net.py:68-128), so ``eval.py`` / ``train.py`` style drivers and the
published checkpoints work unchanged.  What is inside is new:

* parameters live in a name-addressed tree built from ``schema.state_dict_schema()`` (no
  per-layer Python classes);
* test mode runs the MI355X engine of ``itermvs_amd.engine``: every kernel of a depth map is a
  hand-written HIP kernel behind the C ABI (fused warp + correlation, matrix-core convolutions with
  BatchNorm folded at load time and all views batched, CorrNet in one launch, ConvGRU gates in the conv
  epilogues, the depth head + probability regression in one launch, convex up-sampling), optionally
  replayed as one hipGraph per depth map;
* train mode (``itermvs_amd.train_graph``) is differentiable: each ``Evaluation`` call is one
  ``torch.autograd.Function`` on the fused correlation kernels (forward) and their scatter-add gradient
  kernel (backward); the dense layers run on PyTorch-ROCm autograd.

There is no CPU path: CPU tensors raise ``RuntimeError`` (the oracle under ``oracle/`` is test
infrastructure and is never imported from here).
"""
from __future__ import annotations

import math
from typing import Dict, Mapping

import torch
import torch.nn as nn

from .schema import state_dict_schema, strip_module_prefix


class _Node(nn.Module):
    """Anonymous container node of the parameter tree."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, is_buffer: bool) -> None:
    parts = dotted.split(".")
    node = root
    for part in parts[:-1]:
        if part not in node._modules:
            node.add_module(part, _Node())
        node = node._modules[part]
    if is_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], nn.Parameter(tensor))


def _default_init(name: str, shape) -> torch.Tensor:
    """PyTorch's default Conv2d / BatchNorm2d initialisation (what the reference trains from)."""
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    if name.endswith("running_mean") or name.endswith("bn.bias"):
        return torch.zeros(shape)
    if name.endswith("running_var") or name.endswith("bn.weight"):
        return torch.ones(shape)
    schema = state_dict_schema()
    wname = name if name.endswith("weight") else name[: -len("bias")] + "weight"
    wshape = schema[wname]
    # torch.nn.init._calculate_fan_in_and_fan_out takes size(1) * k * k for every weight, ConvTranspose2d's [in,out,k,k]
    # included (so CorrNet's conv3 / conv4 draw from 1/sqrt(16*9) and 1/sqrt(8*9))
    fan_in = wshape[1] * wshape[2] * wshape[3]
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape) * 2 - 1) * bound


class Pipeline(nn.Module):
    """net.py:68-128.  ``Pipeline(iteration=4, test=False)``;
    ``forward(imgs, proj_matrices, depth_min, depth_max)``."""

    def __init__(self, iteration: int = 4, test: bool = False):
        super().__init__()
        self.feature_dim = [8, 16, 32, 48]
        self.hidden_dim = 32
        self.iteration = iteration
        self.test = test
        for name, shape in state_dict_schema().items():
            is_buffer = "running_" in name or name.endswith("num_batches_tracked")
            _attach(self, name, _default_init(name, shape), is_buffer)
        self._engine = None
        self._runners = {}
        self.max_runners = 2        # input shapes whose captured hipGraphs are kept (least recently used dropped first)
        self._engine_version = None
        self._version_tensors = None
        self.use_graphs = False        # test mode: replay one hipGraph per depth map instead of launching kernel by kernel
        # module.py:83,87 assert on NaN projections inside every forward.  Eager test mode does the same (one 4-byte read
        # after the launches are enqueued); the graph mode never stalls the host: call check_projection_finite() when the
        # outputs are fetched (eval.py does, per batch).  None = follow that default.
        self.check_nan = None
        # training mode: a device int32[1] tensor here defers the same asserts (OR-ed with 1 by a forward that composed a NaN
        # projection) instead of reading the flag back inside every forward -- no host synchronisation, capturable
        # (train_step.CapturedTrainStep sets and checks it)
        self.train_nan_flag = None
        # storage type of the feature pyramids the correlation kernels gather from, in test AND train mode: "fp32" (reference
        # numerics), "bf16" / "fp16" (BASELINE cfg 4 / 5: half the gathered bytes, fp32 arithmetic, fp32 gradients); set before
        # the first forward (or call invalidate())
        self.feature_dtype = "fp32"
        # test mode: "bf16x3" (default; the 3x3 layers with more than 8 input channels on the bf16 matrix instructions with an
        # exact three-term split of both operands, fp32-rounding-class error) or "fp32" (exact fp32 MFMA); see InferenceEngine
        self.conv_arithmetic = "bf16x3"
        # test mode: the two launches that depend on FeatureNet only (reference features on the 1/4 grid, up-sampling weights) as a
        # parallel branch of the captured graph (InferenceEngine.side_branch; measured slower, off)
        self.side_branch = False
        # "device_fp64" (default) or "host_fp32": see InferenceEngine -- the second reproduces the reference's fp32
        # `src @ inverse(ref)` on the host (tap indices identical to a reference run on this host; eager mode only)
        self.projection = "device_fp64"
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    # -- weights ----------------------------------------------------------------------------
    def invalidate(self) -> None:
        """Drop the folded / re-laid-out inference weights and the captured hipGraphs.  Test mode folds BatchNorm, packs
        every weight into matrix-core operand order and captures graphs ONCE; they are rebuilt automatically after
        ``load_state_dict``, ``.to()`` / ``.cuda()``, a train/eval mode change, and whenever an autograd-visible in-place
        update of a parameter or buffer is detected at the next forward (optimizer steps, ``p.copy_()`` under ``no_grad``:
        the tensors' version counters).  Edits that bypass the version counter (``p.data.copy_(...)``, raw pointer writes,
        an EMA swap through ``.data``, or REPLACING a Parameter object of a sub-module by assignment -- the polled tensor
        list is collected once per packed engine; ``load_state_dict`` (also with ``assign=True``) and ``.to()`` do reset it)
        must be followed by an explicit ``invalidate()``."""
        self._engine = None
        self._runners = {}
        self._engine_version = None
        self._version_tensors = None

    def _weights_version(self) -> int:
        # polled on every test-mode forward (hipGraph replays included): the ~250 tensors are collected once per packed
        # engine instead of walking the nn.Module generators each time
        ts = self._version_tensors
        if ts is None:
            ts = self._version_tensors = tuple(self.parameters()) + tuple(self.buffers())
        v = 0
        for t in ts:
            v += t._version
        return v

    def load_checkpoint_state(self, state: Mapping[str, torch.Tensor], strict: bool = True):
        """Load a reference checkpoint's ``state_dict['model']`` (keys may carry ``module.``)."""
        return self.load_state_dict(strip_module_prefix(state), strict=strict)

    def train(self, mode: bool = True):
        # the folded inference weights do not depend on the flag: only a real mode change drops them, so a loop that calls
        # model.eval() per sample (train.py:test_sample) does not re-fold / re-capture every time
        if mode != self.training:
            self.invalidate()
        return super().train(mode)

    def check_projection_finite(self) -> None:
        """Raise ``AssertionError`` if any forward since the last check composed a NaN projection (module.py:83,87);
        synchronises with the device."""
        if self._engine is not None:
            self._engine.check_projection_finite()

    def projection_flag(self):
        """the engine's device-side NaN flag (int32[1], OR-ed with 1 by every forward that composed a NaN projection;
        module.py:83,87) or None before the first test-mode forward -- for drivers that download it with the results
        instead of calling check_projection_finite() (which synchronises)"""
        return None if self._engine is None else self._engine.nan_flag

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def weights(self) -> Dict[str, torch.Tensor]:
        return self.state_dict(keep_vars=True)

    # -- forward ----------------------------------------------------------------------------
    def forward(self, imgs, proj_matrices, depth_min, depth_max):
        x = imgs["level_0"]
        if not x.is_cuda:
            raise RuntimeError("itermvs_amd.Pipeline runs on an MI355X only (HIP kernels, no CPU fallback); "
                               "move the sample with tocuda() first")
        h, w = x.shape[-2:]
        if h % 32 or w % 32:
            raise RuntimeError(f"image height and width must be multiples of 32, got {h}x{w}")
        # cameras may stay on the host (tiny; a data loader has them there): projection="host_fp32" then composes them without
        # any synchronisation, also under use_graphs; the default mode simply uploads them
        cams_on_host = not proj_matrices["level_1"].is_cuda
        host_composed = self.test and self.projection == "host_fp32" and cams_on_host
        projs = {l: proj_matrices[f"level_{l}"].float() for l in (1, 2, 3)}
        if cams_on_host and not host_composed:
            projs = {l: t.to(x.device, non_blocking=True) for l, t in projs.items()}
        depth_min = depth_min.float().to(x.device)
        depth_max = depth_max.float().to(x.device)
        if self.test:
            from .engine import InferenceEngine
            if self._engine is not None and self._engine_version != self._weights_version():
                self.invalidate()                 # parameters were updated in place since the weights were packed
            if self._engine is None:
                self._engine = InferenceEngine(self.weights(), self.iteration, self.feature_dtype, self.projection, self.conv_arithmetic,
                                               self.side_branch)
                self._engine_version = self._weights_version()
            with torch.no_grad():
                composed = None
                if host_composed:
                    composed = self._engine.compose_host(torch.stack([projs[1], projs[2], projs[3]]))      # CPU [3,B,S,12]
                    # module.py:83,87 assert right here; the cameras are on the host, so this check is free
                    assert not bool(torch.isnan(composed).any()), "nan in proj (singular or non-finite camera matrix, module.py:83,87)"
                if self.use_graphs:
                    if self.projection != "device_fp64" and not host_composed:
                        raise RuntimeError("projection='host_fp32' with device-resident cameras reads them back on the host: not "
                                           "capturable -- pass proj_matrices as CPU tensors, or use eager mode")
                    from .engine import GraphedRunner
                    key = (tuple(x.shape), x.device.index, host_composed)
                    runner = self._runners.pop(key, None)
                    if runner is None:
                        # one captured graph + static inputs + private workspace per input shape (hundreds of MB at
                        # 1920x1280): keep the ``max_runners`` most recently used shapes, drop the others (their
                        # finalizers free the workspaces) -- a dataset with many image sizes must not grow without bound
                        while len(self._runners) >= max(1, int(self.max_runners)):
                            self._runners.pop(next(iter(self._runners)))
                        runner = GraphedRunner(self._engine, x.float(), projs, depth_min, depth_max, composed=composed)
                    self._runners[key] = runner            # (re-)inserted last: dict order = least recently used first
                    depth_up, conf_up = runner(x.float(), composed if host_composed else projs, depth_min, depth_max)
                elif host_composed:
                    depth_up, conf_up = self._engine.run(x.float(), None, depth_min, depth_max,
                                                         composed=composed.to(x.device, non_blocking=True))
                else:
                    depth_up, conf_up = self._engine.run(x.float(), projs, depth_min, depth_max)
            if self.check_nan if self.check_nan is not None else not self.use_graphs:
                self._engine.check_projection_finite()
            return {"depths_upsampled": depth_up, "confidence_upsampled": conf_up}
        from .train_graph import train_forward
        return train_forward(self.weights(), x.float(), projs, depth_min, depth_max, self.iteration,
                             bn_training=self.training, feature_dtype=self.feature_dtype, nan_flag=self.train_nan_flag)


def full_loss(depths, depths_upsampled, confidences, depths_gt, mask, depth_min, depth_max, regress=True):
    """net.py:131-190 (same signature); see ``itermvs_amd.train_graph.full_loss``."""
    from .train_graph import full_loss as _impl
    return _impl(depths, depths_upsampled, confidences, depths_gt, mask, depth_min, depth_max, regress)
