"""Input side of the path: scan-folder formats, projection-matrix construction
(dtu_yao_eval.py:105-126), and the GPU image pyramid against its (unpinned: cv2 absent) CPU restatement."""
import os

import time

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO

DEV = "cuda"


def write_scan(root, n_views=5, hw=(96, 160), orig_scale=1.0, seed=0):
    """a scan folder in the reference's layout from the photo-consistent synthetic scene; returns (sample, scan name)"""
    from PIL import Image
    from itermvs_amd import synthetic
    h, w = hw
    s = synthetic.make_scene_sample(num_views=n_views, height=h, width=w, seed=seed)
    scan = os.path.join(root, "scan1")
    os.makedirs(os.path.join(scan, "cams_1"))
    os.makedirs(os.path.join(scan, "images"))
    k0, exts = synthetic.camera_parameters(n_views, h, w, ref_shift=seed)
    lines = [str(n_views)]
    for v in range(n_views):
        img = ((s["imgs"]["level_0"][0, v].permute(1, 2, 0).numpy() + 1) * 127.5).round().clip(0, 255).astype(np.uint8)
        if orig_scale != 1.0:
            img = np.asarray(Image.fromarray(img).resize((int(w * orig_scale), int(h * orig_scale)), Image.BILINEAR))
        Image.fromarray(img).save(os.path.join(scan, "images", "{:0>8}.png".format(v)))
        e = exts[v]
        k = np.array(k0, dtype=np.float64)
        k[:2] *= orig_scale
        rows = lambda m: "\n".join(" ".join(repr(float(x)) for x in r) for r in m)
        with open(os.path.join(scan, "cams_1", "{:0>8}_cam.txt".format(v)), "w") as f:
            f.write(f"extrinsic\n{rows(e)}\n\nintrinsic\n{rows(k)}\n\n425.0 2.5 192 935.0\n")
        srcs = [u for u in range(n_views) if u != v]
        lines += [str(v), f"{len(srcs)} " + " ".join(f"{u} 1.0" for u in srcs)]
    with open(os.path.join(scan, "pair.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return s, "scan1"


def test_proj_matrices_follow_the_reference_operation_order():
    """dtu_yao_eval.py:105-126: K scaled to the inference size, x0.125, then doubled per level; P = [K E[:3]; E[3]]"""
    from itermvs_amd.scan_dataset import build_proj_matrices
    rng = np.random.default_rng(0)
    k = np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]], np.float32)
    e = np.eye(4, dtype=np.float32)
    e[:3, :3] = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
    e[:3, 3] = rng.standard_normal(3).astype(np.float32) * 100
    pm = build_proj_matrices(k, e, (1600, 1152), (1600, 1200))
    # restated independently (sequential in-place scaling like the reference, float32)
    kk = k.copy()
    kk[0] *= 1600 / 1600
    kk[1] *= 1152 / 1200
    kk[:2, :] *= 0.125
    for lvl in (3, 2, 1, 0):
        want = e.copy()
        want[:3, :4] = np.matmul(kk, want[:3, :4])
        assert pm[f"level_{lvl}"].dtype == np.float32 and np.array_equal(pm[f"level_{lvl}"], want)
        assert np.array_equal(pm[f"level_{lvl}"][3], e[3])
        kk[:2, :] *= 2
    assert np.array_equal(k, np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]], np.float32))   # input untouched


def test_scan_folder_dataset_reads_the_reference_layout(tmp_path):
    from itermvs_amd.scan_dataset import ScanFolderDataset, read_cam_file
    s, scan = write_scan(str(tmp_path), 5, (96, 160), orig_scale=1.25)
    ds = ScanFolderDataset(str(tmp_path), [scan], nviews=3, img_wh=(160, 96))
    assert len(ds) == 5 and ds.metas[2] == (scan, 2, [0, 1, 3, 4])
    item = ds[2]
    assert item["raw"].shape == (3, 120, 200, 3) and item["raw"].dtype == torch.uint8
    assert item["filename"].format("depth_est", ".pfm") == "scan1/depth_est/00000002.pfm"          # eval.py:141-151 template
    assert float(item["depth_min"]) == 425.0 and float(item["depth_max"]) == 935.0
    # projection matrices = the synthetic sample's (views 2, 0, 1), built from the 1.25x camera files and rescaled
    for l in (1, 2, 3):
        want = s["proj_matrices"][f"level_{l}"][0, [2, 0, 1]]
        got = item["proj_matrices"][f"level_{l}"]
        assert got.shape == (3, 4, 4) and float((got - want).abs().max() / want.abs().max()) < 1e-5
    k, e, dmin, dmax = read_cam_file(os.path.join(str(tmp_path), scan, "cams_1", "00000000_cam.txt"))
    assert k.dtype == np.float32 and e.shape == (4, 4) and (dmin, dmax) == (425.0, 935.0)
    lst = tmp_path / "list.txt"
    lst.write_text(scan + "\n")
    assert len(ScanFolderDataset(str(tmp_path), str(lst), 5, (160, 96))) == 5


def test_image_oracle_is_half_pixel_bilinear():
    """the restated cv2.resize(INTER_LINEAR) equals torch's bilinear interpolation with half-pixel centres (the same
    published formula), and its power-of-two levels are the central 2 x 2 means"""
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    raw = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    p = IO.read_img_pyramid(raw, (160, 96))
    t = torch.from_numpy(2 * raw.astype(np.float32) / 255. - 1).permute(2, 0, 1)[None]
    ti = F.interpolate(t, size=(96, 160), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(ti - p["level_0"]).max() < 1e-6
    l0 = p["level_0"]
    assert np.array_equal(p["level_1"], (l0[0::2, 0::2] * 0.5 + l0[0::2, 1::2] * 0.5) * 0.5 + (l0[1::2, 0::2] * 0.5 + l0[1::2, 1::2] * 0.5) * 0.5)
    assert np.array_equal(p["level_3"], (l0[3::8, 3::8] * 0.5 + l0[3::8, 4::8] * 0.5) * 0.5 + (l0[4::8, 3::8] * 0.5 + l0[4::8, 4::8] * 0.5) * 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("src_hw,dst_hw", [((120, 200), (96, 160)), ((96, 160), (96, 160)), ((75, 131), (96, 160))])
def test_image_pyramid_kernel_matches_the_oracle(src_hw, dst_hw):
    from itermvs_amd import ops
    rng = np.random.default_rng(2)
    raw = rng.integers(0, 256, (3,) + src_hw + (3,), dtype=np.uint8)
    got = ops.image_pyramid(torch.from_numpy(raw).to(DEV), dst_hw[0], dst_hw[1])
    for v in range(3):
        want = IO.read_img_pyramid(raw[v], (dst_hw[1], dst_hw[0]))
        for l in range(4):
            g = got[f"level_{l}"][v].permute(1, 2, 0).cpu().numpy()
            assert g.shape == want[f"level_{l}"].shape
            assert np.abs(g - want[f"level_{l}"]).max() <= 1e-6, (v, l)


@pytest.mark.gpu
def test_eval_folder_mode_end_to_end(tmp_path):
    """eval.py --dataset folder: images / cams / pair.txt on disk -> PFMs equal to the in-memory pipeline on the same
    (8-bit quantised) images, through the prefetcher, the uint8 upload and the pyramid kernel"""
    import eval as E
    from itermvs_amd.data_io import read_pfm
    from itermvs_amd.scan_dataset import ScanFolderDataset, to_device
    s, scan = write_scan(str(tmp_path / "data"), 4, (96, 160))
    out = tmp_path / "out"
    args = E.build_parser().parse_args(["--dataset", "folder", "--testpath", str(tmp_path / "data"), "--n_views", "4",
                                        "--img_wh", "160", "96", "--iteration", "2", "--outdir", str(out)])
    assert E.save_depth(args) == 4
    model = E.load_model(args, torch.device(DEV))
    ds = ScanFolderDataset(str(tmp_path / "data"), [scan], 4, (160, 96))
    for i in (0, 3):
        imgs, projs, dmin, dmax = to_device(ds[i], torch.device(DEV), all_levels=True)
        assert imgs["level_0"].shape == (1, 4, 3, 96, 160) and imgs["level_3"].shape == (1, 4, 3, 12, 20)
        with torch.no_grad():
            want = model(imgs, projs, dmin, dmax)["depths_upsampled"][0, 0].cpu().numpy()
        got = np.squeeze(read_pfm(str(out / scan / "depth_est" / "{:0>8}.pfm".format(i)))[0])
        assert np.array_equal(got, want)
    # the reference view of item 0 is view 0: its level-0 image is the scene image up to the 8-bit quantisation
    imgs, _, _, _ = to_device(ds[0], torch.device(DEV))
    assert float((imgs["level_0"][0, 0].cpu() - s["imgs"]["level_0"][0, 0]).abs().max()) <= 1.0 / 255 + 1e-6


@pytest.mark.gpu
def test_prefetcher_stages_one_sample_ahead(tmp_path):
    """the prefetcher's contract: item n+1's upload + pyramid kernel are enqueued on the side stream BEFORE item n is
    handed out (ordering: its ``ready`` event is already recorded while the consumer works on n), the tensors equal a
    direct ``to_device``, and a consumer that leaves early does not strand the decoder thread"""
    from itermvs_amd.scan_dataset import Prefetcher, ScanFolderDataset, to_device
    _, scan = write_scan(str(tmp_path / "data"), 4, (96, 160))
    ds = ScanFolderDataset(str(tmp_path / "data"), [scan], 4, (160, 96))
    dev = torch.device(DEV)
    pf = Prefetcher(ds, [0, 1, 2, 3], dev, depth=1)
    time.sleep(0.5)                              # the decoder gets ahead of the consumer (item 0 queued, item 1 decoded and waiting)
    seen, ahead = [], []
    for sample, (imgs, projs, dmin, dmax) in pf:
        ahead.append(pf.staged_ahead)
        time.sleep(0.15)                         # a consumer slower than the decoder: the next item is always decoded in time
        torch.cuda.current_stream().synchronize()
        want = to_device(ds[len(seen)], dev)
        assert torch.equal(imgs["level_0"], want[0]["level_0"]) and torch.equal(projs["level_2"], want[1]["level_2"])
        assert torch.equal(dmin, want[2]) and torch.equal(dmax, want[3])
        seen.append(sample["filename"])
    assert len(seen) == 4 and len(set(seen)) == 4
    assert ahead == [1, 2, 3, 3]                 # while item n was consumed, item n+1 was already staged (none after the last)
    assert not pf.thread.is_alive()
    # a decoder SLOWER than the consumer: item n is handed out as soon as it is decoded, not after item n+1's decode
    class Slow:
        def __init__(self, inner): self.inner = inner
        def __len__(self): return len(self.inner)
        def __getitem__(self, i):
            time.sleep(0.3)
            return self.inner[i]
    t0, stamps = time.perf_counter(), []
    for _ in Prefetcher(Slow(ds), [0, 1, 2], dev, depth=1):
        stamps.append(time.perf_counter() - t0)
    assert len(stamps) == 3 and stamps[0] < 0.55 and stamps[1] < 0.9, stamps     # 0.3 s per decode, none waited for twice
                                                                               # (the blocking form hands item 0 out at >= 0.6 s)
    # leaving early: close() drains the queue and the worker exits instead of blocking on a full queue
    pf2 = Prefetcher(ds, [0, 1, 2, 3], dev, depth=1)
    for _ in pf2:
        break
    pf2.thread.join(timeout=10)
    assert not pf2.thread.is_alive()
