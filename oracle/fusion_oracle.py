"""CPU restatement (numpy) of the geometric / photometric filter that follows the depth-inference path:
``reproject_with_depth`` (eval.py:154-194), ``check_geometric_consistency`` (eval.py:197-212) and the per-reference
fusion arithmetic of ``filter_depth`` (eval.py:215-309).  TEST INFRASTRUCTURE ONLY: imported by tests/ and never by
the product path (itermvs_amd/fusion.py calls the HIP kernel).

PARITY UNPINNED.  eval.py cannot be imported in this container (cv2, plyfile and torchvision are absent, SURVEY.md
section 8(c)) and the reference holds no fixtures for this stage, so the restatement follows the source text and, for
``cv2.remap(..., INTER_LINEAR)`` (OpenCV 4.x, not vendored), the published algorithm:
  * coordinates are rounded to 1/32 pixel: s = cvRound(coord * 32) (round-half-even); integer part s >> 5, fraction s & 31;
  * the four weights are products of the float32 1-D coefficients (1 - f/32, f/32);
  * dst = v00*w00 + v01*w01 + v10*w10 + v11*w11 in float32, taps outside the image read the border value 0.
dtype promotion follows numpy's: pixel grids are int64, depths float32, camera matrices float32 (np.linalg.inv and
float32 @ float32 stay float32), every product with the float64 point array is float64.  Matrix products are evaluated
k-ascending without fused multiply-add; BLAS may differ from that in the last float64 bit."""
import numpy as np

INTER_BITS = 5
INTER_TAB = 1 << INTER_BITS


def remap_bilinear(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(src, map_x, map_y, interpolation=cv2.INTER_LINEAR) for float32 single-channel ``src`` (eval.py:176)."""
    h, w = src.shape
    with np.errstate(invalid="ignore", over="ignore"):
        sx = np.rint(map_x.astype(np.float64) * INTER_TAB)
        sy = np.rint(map_y.astype(np.float64) * INTER_TAB)
    big = np.float64(2 ** 31 - 1)
    sx = np.where(np.isfinite(sx), np.clip(sx, -big - 1, big), -big - 1).astype(np.int64)     # saturate_cast<int>
    sy = np.where(np.isfinite(sy), np.clip(sy, -big - 1, big), -big - 1).astype(np.int64)
    ix, iy = sx >> INTER_BITS, sy >> INTER_BITS
    fx = (sx & (INTER_TAB - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB)
    fy = (sy & (INTER_TAB - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB)
    wx0, wx1 = np.float32(1.0) - fx, fx
    wy0, wy1 = np.float32(1.0) - fy, fy

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok, v, np.float32(0.0)).astype(np.float32)

    out = tap(iy, ix) * (wy0 * wx0)
    out = out + tap(iy, ix + 1) * (wy0 * wx1)
    out = out + tap(iy + 1, ix) * (wy1 * wx0)
    out = out + tap(iy + 1, ix + 1) * (wy1 * wx1)
    return out.astype(np.float32)


def _mat3(m, p):
    """(3x3 float32 upcast) @ (3xN float64), k ascending, no FMA"""
    m = m.astype(np.float64)
    return np.stack([(m[i, 0] * p[0] + m[i, 1] * p[1]) + m[i, 2] * p[2] for i in range(3)])


def _mat4(m, p):
    """rows 0..2 of (4x4 float32 upcast) @ [p; 1]"""
    m = m.astype(np.float64)
    return np.stack([((m[i, 0] * p[0] + m[i, 1] * p[1]) + m[i, 2] * p[2]) + m[i, 3] * 1.0 for i in range(3)])


def pair_matrices(k_ref, e_ref, k_src, e_src):
    """the six float32 matrices of one (reference, source) pair, computed like eval.py:162-189 does on the host"""
    return dict(a_ref=np.linalg.inv(k_ref), t_rs=np.matmul(e_src, np.linalg.inv(e_ref)), k_src=k_src,
                a_src=np.linalg.inv(k_src), t_sr=np.matmul(e_ref, np.linalg.inv(e_src)), k_ref=k_ref)


def reproject_with_depth(depth_ref, k_ref, e_ref, depth_src, k_src, e_src):
    """eval.py:154-194 -> depth_reprojected, x_reprojected, y_reprojected, x_src, y_src (float32 [H,W])"""
    h, w = depth_ref.shape
    m = pair_matrices(k_ref, e_ref, k_src, e_src)
    x_ref, y_ref = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x_ref, y_ref = x_ref.reshape(-1), y_ref.reshape(-1)
    d = depth_ref.reshape(-1).astype(np.float64)
    xyz_ref = _mat3(m["a_ref"], np.stack([x_ref * d, y_ref * d, d]))
    xyz_src = _mat4(m["t_rs"], xyz_ref)
    kx = _mat3(m["k_src"], xyz_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_src = kx[:2] / kx[2:3]
    x_src = xy_src[0].reshape(h, w).astype(np.float32)
    y_src = xy_src[1].reshape(h, w).astype(np.float32)
    sampled = remap_bilinear(depth_src, x_src, y_src).reshape(-1).astype(np.float64)
    xyz_src2 = _mat3(m["a_src"], np.stack([xy_src[0] * sampled, xy_src[1] * sampled, sampled]))
    xyz_rep = _mat4(m["t_sr"], xyz_src2)
    depth_rep = xyz_rep[2].reshape(h, w).astype(np.float32)
    kr = _mat3(m["k_ref"], xyz_rep)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_rep = kr[:2] / (kr[2:3] + 1e-6)
    return (depth_rep, xy_rep[0].reshape(h, w).astype(np.float32), xy_rep[1].reshape(h, w).astype(np.float32), x_src, y_src)


def check_geometric_consistency(depth_ref, k_ref, e_ref, depth_src, k_src, e_src, geo_pixel_thres, geo_depth_thres):
    """eval.py:197-212 -> mask (bool), depth_reprojected (zeroed outside the mask), x_src, y_src"""
    h, w = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, w), np.arange(0, h))
    depth_rep, x_rep, y_rep, x_src, y_src = reproject_with_depth(depth_ref, k_ref, e_ref, depth_src, k_src, e_src)
    with np.errstate(invalid="ignore", divide="ignore"):
        dist = np.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
        rel = np.abs(depth_rep - depth_ref) / depth_ref
        mask = np.logical_and(dist < geo_pixel_thres, rel < geo_depth_thres)
    depth_rep = depth_rep.copy()
    depth_rep[~mask] = 0
    return mask, depth_rep, x_src, y_src


def fuse_reference_view(depth_ref, conf_ref, k_ref, e_ref, src_depths, src_ks, src_es, geo_pixel_thres=1.0,
                        geo_depth_thres=0.01, photo_thres=0.3, geo_mask_thres=3):
    """the per-reference part of filter_depth (eval.py:238-269):
    -> depth_est_averaged (float64), photo_mask, geo_mask, final_mask (bool), geo_mask_sum (int32)"""
    photo_mask = conf_ref > photo_thres
    geo_sum = np.zeros(depth_ref.shape, np.int32)
    acc = 0
    for d_src, k_src, e_src in zip(src_depths, src_ks, src_es):
        m, d_rep, _, _ = check_geometric_consistency(depth_ref, k_ref, e_ref, d_src, k_src, e_src, geo_pixel_thres, geo_depth_thres)
        geo_sum += m.astype(np.int32)
        acc = acc + d_rep                                   # sum(all_srcview_depth_ests): left to right, float32
    averaged = (acc + depth_ref) / (geo_sum + 1)            # float32 / int32 -> float64
    geo_mask = geo_sum >= geo_mask_thres
    return averaged, photo_mask, geo_mask, np.logical_and(photo_mask, geo_mask), geo_sum


def unproject_points(depth_avg, mask, k_ref, e_ref):
    """eval.py:287-296: world coordinates [n,3] float64 of the pixels in ``mask``"""
    h, w = depth_avg.shape
    x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x, y, d = x[mask], y[mask], depth_avg[mask]
    xyz_ref = _mat3(np.linalg.inv(k_ref), np.stack([x * d, y * d, d]))
    return _mat4(np.linalg.inv(e_ref), xyz_ref).transpose(1, 0)
