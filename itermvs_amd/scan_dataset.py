"""Scan-folder input side: the on-disk formats in front of the hot path.

``ScanFolderDataset`` reads what the reference's ``datasets/dtu_yao_eval.py`` reads -- ``<scan>/pair.txt``,
``<scan>/cams_1/{view:08d}_cam.txt``, ``<scan>/images/{view:08d}.jpg`` -- and yields the reference's sample dict
(dtu_yao_eval.py:154-158) with one difference made for the MI355X: the images stay the DECODED uint8 RGB arrays
(``raw`` [V,Hs,Ws,3]) and normalisation + resize + pyramid run on the GPU (``ops.image_pyramid`` /
``itermvs_image_pyramid``) after a 5x smaller host-to-device copy.  ``to_device`` turns such a sample into exactly the
``imgs`` / ``proj_matrices`` / ``depth_min`` / ``depth_max`` tensors ``Pipeline.forward`` takes; ``Prefetcher`` decodes the
next samples on a host thread and uploads them on a side stream while the current depth map computes.

Projection matrices follow dtu_yao_eval.py:105-126 operation by operation in float32: intrinsics scaled to the inference
size, then x0.125 and three doublings, ``P_l[:3,:4] = K_l @ E[:3,:4]``, last row = the extrinsic's.  (The reference
hard-codes the DTU image size 1600 x 1200 in that scaling; here the size of the image file is used, which is the same
number for DTU and the right one for any other folder -- eval.py:231-232 does the same in its filter stage.)
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .fusion import read_pair_file


def read_cam_file(filename: str) -> Tuple[np.ndarray, np.ndarray, float, float]:
    """dtu_yao_eval.py:39-52 -> (intrinsics [3,3] f32, extrinsics [4,4] f32, depth_min, depth_max)"""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape((4, 4))
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape((3, 3))
    return intrinsics, extrinsics, float(lines[11].split()[0]), float(lines[11].split()[-1])


def build_proj_matrices(intrinsics: np.ndarray, extrinsics: np.ndarray, img_wh: Sequence[int],
                        orig_wh: Sequence[int]) -> Dict[str, np.ndarray]:
    """dtu_yao_eval.py:105-126 for one view -> {'level_0'..'level_3': [4,4] float32}; same operation order (the
    intrinsics are scaled in place: x img/orig, x0.125, then x2 three times)"""
    k = np.array(intrinsics, dtype=np.float32, copy=True)
    k[0] *= img_wh[0] / orig_wh[0]
    k[1] *= img_wh[1] / orig_wh[1]
    out = {}
    k[:2, :] *= 0.125
    for lvl in (3, 2, 1, 0):
        p = np.array(extrinsics, dtype=np.float32, copy=True)
        p[:3, :4] = np.matmul(k, p[:3, :4])
        out[f"level_{lvl}"] = p
        k[:2, :] *= 2
    return out


class ScanFolderDataset(torch.utils.data.Dataset):
    """metas like dtu_yao_eval.py:19-34: one item per (scan, reference view) of every ``pair.txt``"""

    def __init__(self, datapath: str, scans, nviews: int = 5, img_wh: Sequence[int] = (1600, 1152)):
        self.datapath, self.nviews, self.img_wh = datapath, nviews, tuple(img_wh)
        if isinstance(scans, str):                       # a list file like the reference's --testlist
            with open(scans) as f:
                scans = [line.rstrip() for line in f if line.strip()]
        self.metas: List[Tuple[str, int, List[int]]] = []
        for scan in scans:
            with open(os.path.join(datapath, scan, "pair.txt")) as f:
                n = int(f.readline())
                for _ in range(n):                       # (views without source views are kept, like the reference)
                    ref = int(f.readline().rstrip())
                    srcs = [int(x) for x in f.readline().rstrip().split()[1::2]]
                    self.metas.append((scan, ref, srcs))

    def __len__(self) -> int:
        return len(self.metas)

    def image_path(self, scan: str, vid: int) -> str:
        base = os.path.join(self.datapath, scan, "images", "{:0>8}".format(vid))
        for ext in (".jpg", ".png", ".jpeg"):
            if os.path.isfile(base + ext):
                return base + ext
        raise FileNotFoundError(base + ".jpg")

    def __getitem__(self, idx: int) -> dict:
        from PIL import Image
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        raws, projs = [], {f"level_{l}": [] for l in range(4)}
        depth_min = depth_max = None
        for i, vid in enumerate(view_ids):
            with Image.open(self.image_path(scan, vid)) as im:
                raw = np.asarray(im.convert("RGB"), dtype=np.uint8)
            k, e, dmin, dmax = read_cam_file(os.path.join(self.datapath, scan, "cams_1", "{:0>8}_cam.txt".format(vid)))
            pm = build_proj_matrices(k, e, self.img_wh, (raw.shape[1], raw.shape[0]))
            for l in projs:
                projs[l].append(pm[l])
            raws.append(raw)
            if i == 0:
                depth_min, depth_max = dmin, dmax
        if any(r.shape != raws[0].shape for r in raws):
            raise ValueError(f"{scan}: the views of one sample must share the image size")
        return {"raw": torch.from_numpy(np.stack(raws)), "img_wh": self.img_wh,
                "proj_matrices": {l: torch.from_numpy(np.stack(v)) for l, v in projs.items()},
                "depth_min": torch.tensor(depth_min, dtype=torch.float32), "depth_max": torch.tensor(depth_max, dtype=torch.float32),
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}


def to_device(sample: dict, dev, all_levels: bool = False) -> Tuple[dict, dict, torch.Tensor, torch.Tensor]:
    """one ScanFolderDataset item -> (imgs, proj_matrices, depth_min, depth_max) on ``dev`` with a leading batch dimension
    of 1, the images normalised / resized / pyramided by the HIP kernel on the current stream"""
    from . import ops
    raw = sample["raw"].to(dev, non_blocking=True)
    w, h = sample["img_wh"]
    imgs = {k: v.unsqueeze(0) for k, v in ops.image_pyramid(raw, h, w, all_levels).items()}
    projs = {k: v.to(dev, non_blocking=True).unsqueeze(0) for k, v in sample["proj_matrices"].items()}
    return imgs, projs, sample["depth_min"].to(dev, non_blocking=True).view(1), sample["depth_max"].to(dev, non_blocking=True).view(1)


class Prefetcher:
    """Iterates a ScanFolderDataset shard with ONE sample always staged ahead on the device: a host thread decodes the
    images of the next ``depth`` samples into pinned memory; before item n is handed to the consumer, the upload and the
    pyramid kernel of item n+1 are already enqueued on a side HIP stream (their ``ready`` event recorded), so decode, H2D
    and pyramid of the next reference view overlap the current depth map's kernels and its D2H.  The host waits on the
    item's (long completed) event before the item is yielded; no stream waits on another stream.  ``close()`` (or leaving the iteration early) stops the
    worker thread."""

    def __init__(self, dataset, indices: Sequence[int], dev, depth: int = 2):
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dataset, self.indices, self.dev = dataset, list(indices), dev
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self.stream = torch.cuda.Stream(device=self.dev)
        self._stop = threading.Event()
        self.staged_ahead = 0          # items whose upload was enqueued before the previous item was yielded (tests read it)
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _work(self) -> None:
        try:
            torch.cuda.set_device(self.dev)   # pin through THIS rank's device context, not device 0's
            for i in self.indices:
                s = self.dataset[i]
                s["raw"] = s["raw"].pin_memory()
                if not self._put(s):
                    return
        except Exception as e:  # noqa: BLE001  (surfaced to the consumer)
            self._put(e)
            return
        self._put(None)

    def close(self) -> None:
        """stop the decoder thread (a consumer that leaves early must not strand it on a full queue)"""
        self._stop.set()
        while True:
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        self.thread.join(timeout=5)

    _PENDING = object()      # _stage(block=False): nothing decoded yet

    def _stage(self, block: bool = True):
        """next decoded sample -> (sample, device tensors, ready event) with the upload + pyramid enqueued on the side
        stream; None at the end of the shard; ``_PENDING`` when ``block`` is False and the decoder has nothing ready"""
        while True:
            try:
                # not blocking = a 5 ms grace period: the decoder blocked on a full queue needs a moment to hand over the
                # item it already holds after the consumer took the previous one
                s = self.q.get(timeout=1.0 if block else 0.005)
                break
            except queue.Empty:
                if not block:
                    return self._PENDING
                if not self.thread.is_alive() and self.q.empty():
                    raise RuntimeError("Prefetcher: the decoder thread ended without delivering the end-of-shard marker")
        if s is None:
            return None
        if isinstance(s, Exception):
            raise s
        with torch.cuda.stream(self.stream):
            tensors = to_device(s, self.dev)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return s, tensors, ready

    def __iter__(self):
        try:
            nxt = self._stage()
            while nxt is not None:
                s, tensors, ready = nxt
                # item n+1 goes on its way to the device before item n is consumed -- when it is already decoded.  When
                # decoding is the slower side the consumer gets item n NOW (the GPU must not idle for one decode per item)
                # and n+1 is staged after the yield
                nxt = self._stage(block=False)
                if nxt is not None and nxt is not self._PENDING:
                    self.staged_ahead += 1
                cur = torch.cuda.current_stream(self.dev)
                # the HOST waits for the (long finished: staged one item ahead) upload instead of making the compute stream
                # wait on the side stream's event: a cross-stream wait in front of a depth map's launches measured +60..+150 us
                # per map on MI355X (tools/transfer_lab.py), a host wait on a completed event costs nothing
                ready.synchronize()
                for t in list(tensors[0].values()) + list(tensors[1].values()) + [tensors[2], tensors[3]]:
                    t.record_stream(cur)           # allocated on the side stream, used on this one
                yield s, tensors
                if nxt is self._PENDING:
                    nxt = self._stage()
        finally:
            self.close()
