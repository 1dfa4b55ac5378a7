"""CPU oracle of the input side -- TEST INFRASTRUCTURE ONLY (see oracle/itermvs_oracle.py for the rules).

Restates ``read_img`` of the reference's datasets/dtu_yao_eval.py:61-74: normalisation to -1..1, ``cv2.resize`` to the
inference size and the three lower pyramid levels.  **Parity unpinned**: cv2 is not installed in this image, so
``cv2.resize(INTER_LINEAR)`` on float32 is restated from its published algorithm (modules/imgproc/src/resize.cpp,
resizeGeneric_ with HResizeLinear / VResizeLinear: half-pixel centres, coefficient = float(fx - floor(fx)), borders
clamped with weight 0, horizontal pass then vertical pass in float32) and cannot be checked against the library here.
"""
import numpy as np


def _axis(n_dst: int, n_src: int):
    scale = float(n_src) / float(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0.0, n_src - 1
    s1 = np.minimum(s + 1, n_src - 1)
    return s, s1, (np.float32(1.0) - f).astype(np.float32), f.astype(np.float32)


def resize_linear(img: np.ndarray, wh) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for float32 H x W x C"""
    w, h = wh
    x0, x1, ax0, ax1 = _axis(w, img.shape[1])
    y0, y1, ay0, ay1 = _axis(h, img.shape[0])
    rows = img[:, x0] * ax0[None, :, None] + img[:, x1] * ax1[None, :, None]          # horizontal pass
    return (rows[y0] * ay0[:, None, None] + rows[y1] * ay1[:, None, None]).astype(np.float32)


def read_img_pyramid(raw_u8: np.ndarray, img_wh) -> dict:
    """dtu_yao_eval.py:61-74 on a decoded uint8 H x W x 3 image -> {'level_0'..'level_3'} float32 H x W x 3"""
    np_img = 2 * np.array(raw_u8, dtype=np.float32) / 255. - 1
    np_img = resize_linear(np_img, img_wh)
    h, w, _ = np_img.shape
    return {"level_3": resize_linear(np_img, (w // 8, h // 8)), "level_2": resize_linear(np_img, (w // 4, h // 4)),
            "level_1": resize_linear(np_img, (w // 2, h // 2)), "level_0": np_img}
