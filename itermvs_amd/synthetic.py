"""Seeded synthetic DTU-shaped inputs.

Produces the sample-dict schema of the reference's datasets
(datasets/dtu_yao_eval.py:154-158): ``imgs{'level_0'..'level_3'}`` [B,V,3,H/2^l,W/2^l],
``proj_matrices{'level_0'..'level_3'}`` [B,V,4,4] with rows 0-2 = K_l [R|t] and
row 3 = the extrinsic's last row (dtu_yao_eval.py:108-126), ``depth_min`` /
``depth_max`` [B].  Everything is generated on the CPU from fixed seeds so the
GPU box and the build container see bit-identical inputs.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

DTU_DEPTH_MIN = 425.0
DTU_DEPTH_MAX = 935.0


def _look_at_origin(center: np.ndarray) -> np.ndarray:
    """World->camera rotation whose +z axis points from ``center`` to the origin."""
    z = -center / np.linalg.norm(center)
    up = np.array([0.0, 1.0, 0.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])


def camera_parameters(num_views: int, height: int, width: int, radius: float = 680.0, ref_shift: int = 0):
    """(K0 [3,3] float64 at full resolution, [extrinsic [4,4] float32 per view]) of the rig :func:`make_cameras` projects with"""
    f = 1446.1 * (width / 640.0)
    cx = width / 2.0 + 11.6 * (width / 640.0)
    cy = height / 2.0 + 9.6 * (height / 512.0)
    k0 = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], dtype=np.float64)
    exts = []
    for v in range(num_views):
        step = (v + 1) // 2 * (1 if v % 2 else -1)
        az = math.radians(8.0 * step + 2.5 * ref_shift)
        el = math.radians(3.0 * ((v % 3) - 1))
        c = radius * np.array([math.sin(az) * math.cos(el), math.sin(el), -math.cos(az) * math.cos(el)])
        rot = _look_at_origin(c)
        ext = np.eye(4)
        ext[:3, :3] = rot
        ext[:3, 3] = -rot @ c
        exts.append(ext.astype(np.float32))
    return k0, exts


def make_cameras(num_views: int, height: int, width: int, radius: float = 680.0,
                 ref_shift: int = 0) -> Dict[str, np.ndarray]:
    """V pinhole cameras on an arc around the origin: azimuth steps of 8 deg
    alternating sign, elevation +-3 deg; DTU-like intrinsics scaled to ``width``.
    ``ref_shift`` rotates the whole rig so different reference views differ."""
    k0, exts = camera_parameters(num_views, height, width, radius, ref_shift)
    out = {f"level_{l}": [] for l in range(4)}
    for ext in exts:
        for l in range(4):
            k = k0.copy()
            k[:2] /= 2 ** l
            p = ext.copy()
            p[:3, :4] = (k.astype(np.float32) @ ext[:3, :4]).astype(np.float32)
            out[f"level_{l}"].append(p)
    return {k: np.stack(v) for k, v in out.items()}


def make_sample(batch: int = 1, num_views: int = 5, height: int = 512, width: int = 640,
                seed: int = 0) -> Dict[str, object]:
    """One batch of synthetic reference views (CPU tensors)."""
    assert height % 32 == 0 and width % 32 == 0, "H and W must be multiples of 32 (SURVEY 0.1)"
    gen = torch.Generator().manual_seed(seed)
    img0 = torch.rand((batch, num_views, 3, height, width), generator=gen) * 2 - 1
    imgs = {"level_0": img0}
    flat = img0.view(batch * num_views, 3, height, width)
    for l in range(1, 4):
        pooled = torch.nn.functional.avg_pool2d(flat, 2 ** l)
        imgs[f"level_{l}"] = pooled.view(batch, num_views, 3, height >> l, width >> l)
    projs = {f"level_{l}": [] for l in range(4)}
    for b in range(batch):
        cams = make_cameras(num_views, height, width, ref_shift=seed * batch + b)
        for k, v in cams.items():
            projs[k].append(v)
    proj_matrices = {k: torch.from_numpy(np.stack(v)) for k, v in projs.items()}
    return {
        "imgs": imgs,
        "proj_matrices": proj_matrices,
        "depth_min": torch.full((batch,), DTU_DEPTH_MIN, dtype=torch.float32),
        "depth_max": torch.full((batch,), DTU_DEPTH_MAX, dtype=torch.float32),
    }


def random_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the reference's state_dict schema
    (SURVEY.md section 9.4): Kaiming-uniform-like conv weights, non-trivial
    batch-norm statistics so BN folding is exercised.  Generated without
    importing any model class, so the GPU box can rebuild them."""
    from .schema import state_dict_schema

    gen = torch.Generator().manual_seed(1000 + seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in state_dict_schema().items():
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_var"):
            out[name] = torch.rand(shape, generator=gen) * 0.5 + 0.75
        elif name.endswith("running_mean"):
            out[name] = (torch.rand(shape, generator=gen) - 0.5) * 0.2
        elif name.endswith("bn.weight"):
            out[name] = torch.rand(shape, generator=gen) * 0.5 + 0.75
        elif name.endswith("bias"):
            out[name] = (torch.rand(shape, generator=gen) - 0.5) * 0.2
        else:  # conv / deconv weight
            fan_in = int(np.prod(shape[1:]))
            bound = math.sqrt(3.0 / fan_in) * 1.2
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
    return out


def _texture(u: np.ndarray, v: np.ndarray, seed: int) -> np.ndarray:
    """Procedural multi-octave texture in [-1,1]^3 as a function of surface coordinates."""
    rng = np.random.RandomState(seed)
    out = np.zeros(u.shape + (3,), dtype=np.float64)
    for octave in range(5):
        freq = 0.02 * (2.1 ** octave)
        for c in range(3):
            a, b, ph1, ph2 = rng.uniform(0.5, 1.5), rng.uniform(0.5, 1.5), rng.uniform(0, 6.28), rng.uniform(0, 6.28)
            th = rng.uniform(0, np.pi)
            uu = np.cos(th) * u + np.sin(th) * v
            vv = -np.sin(th) * u + np.cos(th) * v
            out[..., c] += (0.6 ** octave) * np.sin(a * freq * uu + ph1) * np.cos(b * freq * vv + ph2)
    out /= np.abs(out).max() + 1e-9
    return out


def make_scene_sample(num_views: int = 5, height: int = 512, width: int = 640, seed: int = 0,
                      tilt_deg: float = 20.0) -> Dict[str, object]:
    """Photo-consistent synthetic views: every camera of :func:`make_cameras` images the
    same textured, tilted plane through the world origin, so a trained network sees a
    real matching signal (stable arg-max, unlike noise images).  Also returns the exact
    reference-view depth map ``depth_gt`` [1,1,H,W] (camera-z of the plane)."""
    assert height % 32 == 0 and width % 32 == 0
    cams = make_cameras(num_views, height, width, ref_shift=seed)
    t = math.radians(tilt_deg)
    normal = np.array([math.sin(t), 0.3 * math.sin(t), -math.cos(t)])
    normal /= np.linalg.norm(normal)
    e1 = np.cross(normal, [0.0, 1.0, 0.0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(normal, e1)
    ys, xs = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    pix = np.stack([xs, ys, np.ones_like(xs)], axis=-1)                       # [H,W,3]
    images, depth_gt = [], None
    for vi in range(num_views):
        p = cams["level_0"][vi].astype(np.float64)                             # K[R|t]
        m, q = p[:3, :3], p[:3, 3]
        minv = np.linalg.inv(m)
        center = -minv @ q                                                     # camera centre (world)
        dirs = pix @ minv.T                                                    # ray directions, z_cam = 1 units
        s = -(center @ normal) / (dirs @ normal)                               # plane: X.n = 0
        pts = center + dirs * s[..., None]
        images.append(_texture(pts @ e1, pts @ e2, 77 + seed).transpose(2, 0, 1))
        if vi == 0:
            depth_gt = s                                                       # third row of K[R|t] is [r3|t3]
    img0 = torch.from_numpy(np.stack(images)[None].astype(np.float32))         # [1,V,3,H,W]
    imgs = {"level_0": img0}
    flat = img0.view(num_views, 3, height, width)
    for l in range(1, 4):
        imgs[f"level_{l}"] = torch.nn.functional.avg_pool2d(flat, 2 ** l).view(1, num_views, 3, height >> l, width >> l)
    return {
        "imgs": imgs,
        "proj_matrices": {k: torch.from_numpy(v[None]) for k, v in cams.items()},
        "depth_min": torch.full((1,), DTU_DEPTH_MIN, dtype=torch.float32),
        "depth_max": torch.full((1,), DTU_DEPTH_MAX, dtype=torch.float32),
        "depth_gt": torch.from_numpy(depth_gt[None, None].astype(np.float32)),
    }


def make_training_sample(num_views: int = 5, height: int = 512, width: int = 640, seed: int = 2, hole_fraction: float = 0.1):
    """One BASELINE cfg-4 shaped training sample (B = 1): the photo-consistent scene of :func:`make_scene_sample` with its
    exact depth as ground truth and a seeded validity mask with ``hole_fraction`` of the pixels masked out, in the
    reference's training schema (datasets/dtu_yao.py:227-232) -> (sample, depth_gt {'level_0','level_2'}, mask {...})."""
    sample = make_scene_sample(num_views=num_views, height=height, width=width, seed=seed)
    gt0 = sample["depth_gt"]
    gen = torch.Generator().manual_seed(31 + seed)
    m0 = (torch.rand(gt0.shape, generator=gen) > hole_fraction).float()
    return sample, {"level_0": gt0, "level_2": gt0[:, :, ::4, ::4].contiguous()}, \
        {"level_0": m0, "level_2": m0[:, :, ::4, ::4].contiguous()}


def make_training_batch(batch: int, num_views: int = 5, height: int = 512, width: int = 640, seed: int = 0,
                        hole_fraction: float = 0.0):
    """A training batch of ``batch`` DIFFERENT scenes / reference views in the reference's collated schema
    (train.py:89-90 with ``--batch_size``; datasets/dtu_yao.py:227-232): every tensor of :func:`make_training_sample`
    concatenated along dim 0; sample ``i`` is seeded ``seed + i``.
    -> (imgs, proj_matrices, depth_min [B], depth_max [B], depth_gt {'level_0','level_2'}, mask {...})."""
    if batch < 1:
        raise ValueError("batch must be >= 1")
    parts = [make_training_sample(num_views, height, width, seed=seed + i, hole_fraction=hole_fraction) for i in range(batch)]
    cat = lambda pick: torch.cat([pick(p) for p in parts], 0)  # noqa: E731
    s0 = parts[0][0]
    imgs = {k: cat(lambda p: p[0]["imgs"][k]) for k in s0["imgs"]}
    projs = {k: cat(lambda p: p[0]["proj_matrices"][k]) for k in s0["proj_matrices"]}
    gt = {k: cat(lambda p: p[1][k]) for k in parts[0][1]}
    mask = {k: cat(lambda p: p[2][k]) for k in parts[0][2]}
    return imgs, projs, cat(lambda p: p[0]["depth_min"]), cat(lambda p: p[0]["depth_max"]), gt, mask
