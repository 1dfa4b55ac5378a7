#!/usr/bin/env python3
"""Probe: the cfg-4 training step with every F.conv2d / F.conv_transpose2d of the training graph fed channels-last tensors
(MIOpen's NHWC implicit-GEMM kernels then need no layout transposes: 532 batched_transpose launches, 5 ms of a 59 ms step at
B = 4 in the eager profile).  Timing only; the HIP BatchNorm op needs dense NCHW planes, so its inputs are made contiguous again."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import benchmarks, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
_conv2d, _convT = F.conv2d, F.conv_transpose2d
cl = torch.channels_last


def conv2d(x, w, *a, **k):
    return _conv2d(x.contiguous(memory_format=cl), w.contiguous(memory_format=cl), *a, **k)


def convT(x, w, *a, **k):
    return _convT(x.contiguous(memory_format=cl), w.contiguous(memory_format=cl), *a, **k)


if mode != "baseline":
    F.conv2d, F.conv_transpose2d = conv2d, convT
    _bn = ops.bn_relu_train
    ops.bn_relu_train = lambda y, *a, **k: _bn(y.contiguous(), *a, **k)
res = benchmarks.train_step_leg(torch.device("cuda", 0), batch=4, feature_dtype="bf16", regress=True, warmup=3, steps=8)
res.pop("_step")
print(mode, {k: res[k] for k in ("ms_per_step", "samples_per_s", "loss")})
