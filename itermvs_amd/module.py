"""Drop-in counterparts of the reference's ``models/module.py`` primitives, HIP-backed.

Same names, argument meaning and return shapes as the reference functions they replace; all
tensors must live on the GPU (no CPU fallback).
"""
from __future__ import annotations

import torch

from . import ops


def differentiable_warping(src_fea, src_proj, ref_proj, depth_samples, return_mask=False):
    """module.py:68-125.  src_fea [B,C,H1,W1]; src_proj/ref_proj [B,4,4]; depth_samples
    [B,N,H,W] -> warped [B,C,N,H,W] (and the validity mask).  Differentiable w.r.t. ``src_fea``
    only, like the reference (its grid math is under no_grad, module.py:77)."""
    with torch.no_grad():
        mats = torch.stack([ref_proj.float(), src_proj.float()], dim=1)       # [B,2,4,4], view 0 = reference
        flag = torch.zeros(1, dtype=torch.int32, device=mats.device)
        proj12 = ops.compose_proj(mats, flag)[:, 0]                            # [B,12]
    return ops.warp(src_fea, proj12, depth_samples, return_mask)


def upsample(x, upsample_weight, scale=4):
    """module.py:127-140: convex combination up-sampling.  ``upsample_weight``
    [B,1,9,4,4,H,W] is already soft-maxed over dim 2 (itermvs.py:264); the HIP kernel applies the
    soft-max itself, so it is fed log-weights (softmax(log w) == w)."""
    if scale != 4:
        raise RuntimeError("upsample: the HIP kernel implements the reference's scale=4")
    b, _, h, w = x.shape
    logits = torch.log(upsample_weight.float()).reshape(b, 144, h, w)
    dummy = torch.ones(b, device=x.device)
    _, norm = ops.convex_upsample(logits, x.float().contiguous(), dummy, dummy, nd_channel=0, want_norm=True)
    return norm


def depth_normalization(depth, inverse_depth_min, inverse_depth_max):
    """module.py:142-146 (element-wise glue; stays a PyTorch expression)."""
    inverse_depth = 1.0 / (depth + 1e-5)
    return (inverse_depth - inverse_depth_max) / (inverse_depth_min - inverse_depth_max)


def depth_unnormalization(normalized_depth, inverse_depth_min, inverse_depth_max):
    """module.py:148-152."""
    return 1.0 / (inverse_depth_max + normalized_depth * (inverse_depth_min - inverse_depth_max))
