"""Depth-map filtering and fusion after inference (the reference's ``filter_depth``, eval.py:215-309) on the GPU:
one ``itermvs_fuse_depth`` launch per reference view instead of ~40 full-frame numpy passes per (reference, source)
pair.  Host side: the text formats of a scan folder (``pair.txt``, ``cams_1/*_cam.txt``, eval.py:56-65,90-100), PFM
depth / confidence maps (datasets/data_io.py) and the fused point cloud as a binary little-endian PLY with the vertex
layout of eval.py:298-308 (x, y, z float32; red, green, blue uint8)."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .data_io import read_pfm


def read_camera_parameters(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    """eval.py:56-65 -> (intrinsics [3,3], extrinsics [4,4]) float32"""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape((4, 4))
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape((3, 3))
    return intrinsics, extrinsics


def read_pair_file(filename: str) -> List[Tuple[int, List[int]]]:
    """eval.py:90-100"""
    data = []
    with open(filename) as f:
        num_viewpoint = int(f.readline())
        for _ in range(num_viewpoint):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if len(src_views) != 0:
                data.append((ref_view, src_views))
    return data


def pair_matrices(k_ref: np.ndarray, e_ref: np.ndarray, k_src: np.ndarray, e_src: np.ndarray) -> np.ndarray:
    """the six float32 matrices of one (reference, source) pair in itermvs_fuse_depth's [60] layout, inverted and
    composed with numpy in float32 exactly as eval.py:162-189 does"""
    t_rs = np.matmul(e_src, np.linalg.inv(e_ref))
    t_sr = np.matmul(e_ref, np.linalg.inv(e_src))
    parts = [np.linalg.inv(k_ref), t_rs[:3], k_src, np.linalg.inv(k_src), t_sr[:3], k_ref]
    return np.concatenate([np.asarray(p, np.float32).reshape(-1) for p in parts])


def fuse_reference_view(depth_ref, conf_ref, k_ref, e_ref, src_depths: Sequence, src_ks: Sequence, src_es: Sequence,
                        geo_pixel_thres=1.0, geo_depth_thres=0.01, photo_thres=0.3, geo_mask_thres=3, device="cuda"):
    """eval.py:238-269 for one reference view (numpy or torch inputs) -> torch tensors
    (depth_est_averaged float64, photo_mask, geo_mask, final_mask uint8, geo_mask_sum int32)"""
    dev = torch.device(device)
    as_t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, torch.float32)
    mats = np.stack([pair_matrices(np.asarray(k_ref), np.asarray(e_ref), np.asarray(k), np.asarray(e))
                     for k, e in zip(src_ks, src_es)])
    return ops.fuse_depth(as_t(depth_ref), as_t(conf_ref), [as_t(d) for d in src_depths], torch.from_numpy(mats).to(dev),
                          geo_pixel_thres, geo_depth_thres, photo_thres, geo_mask_thres)


def write_ply(filename: str, xyz: np.ndarray, rgb: np.ndarray) -> None:
    """binary little-endian PLY, vertex = (x, y, z float32, red, green, blue uint8) -- the element eval.py:298-308 builds"""
    n = xyz.shape[0]
    v = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n)
    with open(filename, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def read_scan_image(filename: str, img_wh: Tuple[int, int], want_pixels: bool = True):
    """eval.py:68-74 ``read_img``: -> (RGB float32 [H,W,3] in 0..1 resized to ``img_wh`` or None, original_h, original_w).
    The reference resizes with ``cv2.resize(INTER_LINEAR)``; cv2 is not a dependency here, PIL's bilinear filter takes its
    place (vertex colours only -- the geometry never reads the pixels)."""
    from PIL import Image
    with Image.open(filename) as im:
        original_w, original_h = im.size
        px = None
        if want_pixels:
            im = im.convert("RGB")
            if (original_w, original_h) != tuple(img_wh):
                im = im.resize(tuple(img_wh), Image.BILINEAR)
            px = np.asarray(im, dtype=np.float32) / 255.0
    return px, original_h, original_w


def filter_depth(scan_folder: str, out_folder: str, plyfilename: str, geo_pixel_thres: float, geo_depth_thres: float,
                 photo_thres: float, images: Dict[int, np.ndarray] = None, intrinsics_scale: Tuple[float, float] = None,
                 geo_mask_thres: int = 3, device: str = "cuda", img_wh: Tuple[int, int] = None) -> Dict[str, float]:
    """eval.py:215-309: for every reference view of ``pair.txt`` fuse its depth map with its source views' and append the
    surviving pixels to one point cloud.

    Like the reference (eval.py:231-232, 251-252) the ``cams_1`` intrinsics of EVERY view are rescaled by
    ``img_wh / original image size`` and the points are coloured from the resized reference image.  Either
      * ``img_wh`` = (width, height) of the depth maps: ``<scan_folder>/images/{view:08d}.jpg`` is opened for its original
        size (and, for reference views, its pixels); a missing image raises -- the filter refuses to run with a guessed K; or
      * ``images[view]`` = [H,W,3] float RGB in 0..1 at the depth maps' resolution plus ``intrinsics_scale`` =
        (img_w / original_w, img_h / original_h) for callers that hold the images already (default (1, 1))."""
    pairs = read_pair_file(os.path.join(scan_folder, "pair.txt"))
    if img_wh is None and images is None and intrinsics_scale is None:
        raise ValueError("filter_depth: pass img_wh (images/ folder is read for the original sizes and colours) or "
                         "images + intrinsics_scale; fusing with unscaled intrinsics would be silently wrong")
    cams, depths, pixels = {}, {}, {}

    def image_path(v):
        base = os.path.join(scan_folder, "images/{:0>8}".format(v))
        for ext in (".jpg", ".png", ".jpeg"):
            if os.path.isfile(base + ext):
                return base + ext
        raise FileNotFoundError(f"{base}.jpg: the filter needs every view's image for the intrinsics scale (eval.py:231-232)")

    def scale_of(v, want_pixels):
        if img_wh is None:
            return intrinsics_scale or (1.0, 1.0)
        px, oh, ow = read_scan_image(image_path(v), img_wh, want_pixels)
        if px is not None:
            pixels[v] = px
        return img_wh[0] / ow, img_wh[1] / oh

    def cam(v, want_pixels=False):
        if v not in cams or (want_pixels and img_wh is not None and v not in pixels):
            k, e = read_camera_parameters(os.path.join(scan_folder, "cams_1/{:0>8}_cam.txt".format(v)))
            sx, sy = scale_of(v, want_pixels)
            k = k.copy()
            k[0] *= sx           # python float * float32 row -> float32, like eval.py:231-232
            k[1] *= sy
            cams[v] = (k, e)
        return cams[v]

    def depth(v):
        if v not in depths:
            d = read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0]
            depths[v] = torch.from_numpy(np.ascontiguousarray(np.squeeze(d))).to(device)
        return depths[v]

    vertexs, colors, stats = [], [], {}
    for ref_view, src_views in pairs:
        k_ref, e_ref = cam(ref_view, want_pixels=True)
        conf = np.squeeze(read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(ref_view)))[0])
        avg, photo, geo, final, _ = fuse_reference_view(depth(ref_view), conf, k_ref, e_ref, [depth(v) for v in src_views],
                                                        [cam(v)[0] for v in src_views], [cam(v)[1] for v in src_views],
                                                        geo_pixel_thres, geo_depth_thres, photo_thres, geo_mask_thres, device)
        final_np = final.cpu().numpy().astype(bool)
        stats[ref_view] = (float(geo.float().mean()), float(photo.float().mean()), float(final.float().mean()))
        h, w = final_np.shape
        x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
        x, y, d = x[final_np], y[final_np], avg.cpu().numpy()[final_np]
        xyz_ref = np.matmul(np.linalg.inv(k_ref), np.vstack((x, y, np.ones_like(x))) * d)          # eval.py:291-292
        xyz_world = np.matmul(np.linalg.inv(e_ref), np.vstack((xyz_ref, np.ones_like(x))))[:3]    # eval.py:293-294
        vertexs.append(xyz_world.transpose((1, 0)))
        if images is not None:
            img = images[ref_view]
        elif ref_view in pixels:
            img = pixels.pop(ref_view)
        else:
            img = np.full((h, w, 3), 0.5, np.float32)
        if img.shape[:2] != (h, w):
            raise ValueError(f"view {ref_view}: image is {img.shape[:2]}, depth map is {(h, w)}")
        colors.append((img[final_np] * 255).astype(np.uint8))
    write_ply(plyfilename, np.concatenate(vertexs, 0).astype(np.float32), np.concatenate(colors, 0))
    return stats
