"""Depth-map filter / fusion (eval.py:154-309): the numpy oracle's own properties on CPU, the HIP kernel against the
oracle bit for bit on the GPU, and the scan-folder driver end to end.  The oracle is UNPINNED (eval.py needs cv2)."""
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as FO

DEV = "cuda"


def _scene(h=96, w=128, n_views=5, seed=0, noise=0.0):
    """a tilted plane seen by cameras on an arc: per view (K, E, depth map, confidence)"""
    rng = np.random.default_rng(seed)
    k = np.array([[1.2 * w, 0, w / 2 + 1.3], [0, 1.2 * w, h / 2 - 0.7], [0, 0, 1]], np.float32)
    normal, offset = np.array([0.1, -0.05, 1.0]), 700.0          # plane n.X = offset in world coordinates
    views = []
    for v in range(n_views):
        ang = np.deg2rad(6.0 * (v - n_views // 2))
        r = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        c = np.array([120.0 * np.sin(ang), 4.0 * v, 0.0])
        e = np.eye(4)
        e[:3, :3], e[:3, 3] = r, -r @ c
        e = e.astype(np.float32)
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        rays = np.linalg.inv(k.astype(np.float64)) @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)])
        rw = r.T @ rays                                          # ray directions in world space
        t = (offset - normal @ c) / (normal @ rw)                # X = c + t * rw on the plane; depth = t (rays have z = 1)
        d = t.reshape(h, w).astype(np.float32)
        if noise:
            d = d * (1 + noise * rng.standard_normal(d.shape)).astype(np.float32)
        conf = rng.uniform(0, 1, (h, w)).astype(np.float32)
        views.append((k.copy(), e, d, conf))
    return views


def test_oracle_identity_and_plane_consistency():
    views = _scene()
    k, e, d, _ = views[0]
    m, d_rep, xs, ys = FO.check_geometric_consistency(d, k, e, d, k, e, 1.0, 0.01)
    assert m.all() and np.abs(d_rep - d).max() < 1e-3
    # an exact plane is consistent wherever the reprojection stays inside the source image
    k2, e2, d2, _ = views[3]
    m, d_rep, xs, ys = FO.check_geometric_consistency(d, k, e, d2, k2, e2, 1.0, 0.01)
    inside = (xs > 1) & (xs < d.shape[1] - 2) & (ys > 1) & (ys < d.shape[0] - 2)
    assert m[inside].mean() > 0.999 and np.abs(d_rep[inside & m] / d[inside & m] - 1).max() < 1e-3
    assert not m[~((xs > -1) & (xs < d.shape[1]) & (ys > -1) & (ys < d.shape[0]))].any()


def test_oracle_remap_matches_plain_bilinear_on_the_32nd_grid():
    rng = np.random.default_rng(1)
    src = rng.standard_normal((20, 30)).astype(np.float32)
    gx = (rng.integers(0, 29 * 32, (20, 30)) / 32.0).astype(np.float32)     # exactly representable 1/32 positions
    gy = (rng.integers(0, 19 * 32, (20, 30)) / 32.0).astype(np.float32)
    x0, y0 = np.floor(gx).astype(int), np.floor(gy).astype(int)
    fx, fy = gx - x0, gy - y0
    want = (src[y0, x0] * (1 - fy) * (1 - fx) + src[y0, x0 + 1] * (1 - fy) * fx + src[y0 + 1, x0] * fy * (1 - fx)
            + src[y0 + 1, x0 + 1] * fy * fx)
    assert np.abs(FO.remap_bilinear(src, gx, gy) - want).max() < 1e-5
    # outside the image: zero border, NaN coordinates: zero
    out = FO.remap_bilinear(src, np.array([[-5.0, 40.0, np.nan]], np.float32), np.array([[3.0, 3.0, 3.0]], np.float32))
    assert (out == 0).all()


def test_oracle_fusion_counts_and_average():
    views = _scene(noise=0.002, seed=3)
    k, e, d, conf = views[2]
    others = [views[i] for i in (0, 1, 3, 4)]
    avg, photo, geo, final, cnt = FO.fuse_reference_view(d, conf, k, e, [o[2] for o in others], [o[0] for o in others],
                                                         [o[1] for o in others])
    assert avg.dtype == np.float64 and cnt.dtype == np.int32 and cnt.max() <= 4 and cnt.min() >= 0
    assert (photo == (conf > np.float32(0.3))).all() and (geo == (cnt >= 3)).all() and (final == (photo & geo)).all()
    assert (avg[cnt == 0] == d[cnt == 0]).all()                 # no consistent source view: the reference depth itself
    assert np.abs(avg[cnt == 4] / d[cnt == 4] - 1).max() < 0.01


def test_driver_formats_roundtrip(tmp_path):
    from itermvs_amd import fusion
    cam = tmp_path / "00000003_cam.txt"
    cam.write_text("extrinsic\n1 0 0 1.5\n0 1 0 -2\n0 0 1 3\n0 0 0 1\n\nintrinsic\n100 0 32\n0 101 24\n0 0 1\n\n425 2.5\n")
    k, e = fusion.read_camera_parameters(str(cam))
    assert k.dtype == np.float32 and k[1, 1] == 101 and e[0, 3] == 1.5 and e.shape == (4, 4)
    (tmp_path / "pair.txt").write_text("3\n0\n2 1 0.5 2 0.4\n1\n0\n2\n1 0 0.9\n")
    assert fusion.read_pair_file(str(tmp_path / "pair.txt")) == [(0, [1, 2]), (2, [0])]
    xyz = np.arange(12, dtype=np.float32).reshape(4, 3)
    rgb = (np.arange(12) * 5).astype(np.uint8).reshape(4, 3)
    fusion.write_ply(str(tmp_path / "c.ply"), xyz, rgb)
    raw = (tmp_path / "c.ply").read_bytes()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex 4" in head and b"format binary_little_endian 1.0" in head and len(body) == 4 * 15
    assert np.frombuffer(body[:12], "<f4").tolist() == [0.0, 1.0, 2.0] and body[12:15] == bytes([0, 5, 10])
    m = fusion.pair_matrices(k, e, k, e)
    assert m.shape == (60,) and np.allclose(m[9:21].reshape(3, 4), np.eye(4)[:3], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(96, 128, 0.0, 0), (96, 128, 0.004, 1), (70, 90, 0.02, 2)])
def test_fuse_kernel_matches_oracle_bit_for_bit(case):
    from itermvs_amd import fusion
    h, w, noise, seed = case
    views = _scene(h, w, 5, seed, noise)
    if seed == 2:                                    # holes and invalid depths in the maps
        views[1][2][10:20, 30:50] = 0.0
        views[2][2][5, 5] = np.nan
    k, e, d, conf = views[2]
    others = [views[i] for i in (0, 1, 3, 4)]
    want = FO.fuse_reference_view(d, conf, k, e, [o[2] for o in others], [o[0] for o in others], [o[1] for o in others])
    got = fusion.fuse_reference_view(d, conf, k, e, [o[2] for o in others], [o[0] for o in others], [o[1] for o in others],
                                     device=DEV)
    avg, photo, geo, final, cnt = [t.cpu().numpy() for t in got]
    assert (cnt == want[4]).all()                                           # integer work: exact
    assert (photo.astype(bool) == want[1]).all() and (geo.astype(bool) == want[2]).all() and (final.astype(bool) == want[3]).all()
    both = np.isfinite(want[0])
    assert (np.isfinite(avg) == both).all() and (avg[both] == want[0][both]).all()    # float64 average: bit for bit
    if noise < 0.01:
        assert final.mean() > 0.2


def test_read_scan_image_reports_original_size_and_resizes(tmp_path):
    """eval.py:68-74: pixels in 0..1 at img_wh, plus the ORIGINAL height / width the intrinsics scale is built from"""
    from PIL import Image
    from itermvs_amd import fusion
    rgb = np.zeros((40, 60, 3), np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 255, 128, 0
    Image.fromarray(rgb).save(str(tmp_path / "a.png"))
    px, oh, ow = fusion.read_scan_image(str(tmp_path / "a.png"), (30, 20))
    assert (oh, ow) == (40, 60) and px.shape == (20, 30, 3) and px.dtype == np.float32
    assert np.allclose(px[..., 0], 1.0) and np.allclose(px[..., 1], 128 / 255.0) and np.allclose(px[..., 2], 0.0)
    none, oh, ow = fusion.read_scan_image(str(tmp_path / "a.png"), (30, 20), want_pixels=False)
    assert none is None and (oh, ow) == (40, 60)
    (tmp_path / "pair.txt").write_text("0\n")
    with pytest.raises(ValueError, match="img_wh"):      # neither img_wh nor an explicit scale: refuse instead of guessing K
        fusion.filter_depth(str(tmp_path), str(tmp_path), str(tmp_path / "x.ply"), 1.0, 0.01, 0.3)


@pytest.mark.gpu
@pytest.mark.parametrize("orig_scale", [1.0, 2.0])
def test_filter_depth_scene_folder(tmp_path, orig_scale):
    """eval.py:215-309 end to end on a synthetic scan folder: PFMs + cams + images + pair.txt -> PLY on the plane.
    ``orig_scale`` = 2: the images on disk (and the intrinsics in cams_1, which describe them) are twice the size of the
    depth maps, so the fused cloud only lands on the plane if every view's K is rescaled like eval.py:231-232,251-252."""
    from PIL import Image
    from itermvs_amd import fusion
    from itermvs_amd.data_io import save_pfm
    h, w = 64, 96
    views = _scene(h, w, 5, 5, 0.001)
    scan, out = tmp_path / "scan1", tmp_path / "out"
    (scan / "cams_1").mkdir(parents=True)
    (scan / "images").mkdir(parents=True)
    (out / "depth_est").mkdir(parents=True)
    (out / "confidence").mkdir(parents=True)
    lines = ["5"]
    for v, (k, e, d, conf) in enumerate(views):
        rows = lambda m: "\n".join(" ".join(repr(float(x)) for x in r) for r in m)
        k_disk = k.copy()
        k_disk[:2] *= orig_scale                    # intrinsics of the full-size image on disk
        (scan / "cams_1" / "{:0>8}_cam.txt".format(v)).write_text(f"extrinsic\n{rows(e)}\n\nintrinsic\n{rows(k_disk)}\n\n425 2.5\n")
        img = np.zeros((int(h * orig_scale), int(w * orig_scale), 3), np.uint8)
        img[..., 0], img[..., 1], img[..., 2] = 10 + 40 * v, 200, 7
        Image.fromarray(img).save(str(scan / "images" / "{:0>8}.jpg".format(v)), quality=100)
        save_pfm(str(out / "depth_est" / "{:0>8}.pfm".format(v)), d)
        save_pfm(str(out / "confidence" / "{:0>8}.pfm".format(v)), np.full_like(conf, 0.9))
        srcs = [u for u in range(5) if u != v]
        lines += [str(v), f"{len(srcs)} " + " ".join(f"{u} 1.0" for u in srcs)]
    (scan / "pair.txt").write_text("\n".join(lines) + "\n")
    stats = fusion.filter_depth(str(scan), str(out), str(tmp_path / "fused.ply"), 1.0, 0.01, 0.3, device=DEV, img_wh=(w, h))
    assert len(stats) == 5 and all(s[2] > 0.3 for s in stats.values())
    head, body = (tmp_path / "fused.ply").read_bytes().split(b"end_header\n")
    n = int(head.split(b"element vertex ")[1].split(b"\n")[0])
    pts = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    assert len(pts) == n > 1000
    plane = 0.1 * pts["x"] - 0.05 * pts["y"] + 1.0 * pts["z"]
    assert np.abs(plane - 700.0).max() < 5.0                                 # every fused point lies on the plane
    assert np.abs(pts["g"].astype(int) - 200).max() <= 3 and set(np.round((pts["r"].astype(int) - 10) / 40.0).astype(int)) <= set(range(5))
    if orig_scale != 1.0:       # and with the camera files taken at face value it does not
        with pytest.raises(AssertionError):
            st = fusion.filter_depth(str(scan), str(out), str(tmp_path / "bad.ply"), 1.0, 0.01, 0.3, device=DEV,
                                     intrinsics_scale=(1.0, 1.0))
            assert all(s[2] > 0.3 for s in st.values())
