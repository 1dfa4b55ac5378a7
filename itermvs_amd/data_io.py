"""PFM reader / writer for depth and confidence maps -- the on-disk format either side of the hot
path (reference: datasets/data_io.py:6-73; written by eval.py:141-151).

Format: ASCII header ``Pf\\n`` (1 channel) or ``PF\\n`` (3 channels), ``<width> <height>\\n``,
``<scale>\\n`` (negative = little-endian), then rows bottom-to-top as raw float32.
"""
from __future__ import annotations

import os
import re
import sys
from typing import Tuple

import numpy as np


def save_pfm(filename: str, image: np.ndarray, scale: float = 1.0) -> None:
    """data_io.py:45-73: float32 H x W (or H x W x 1 / H x W x 3), stored bottom-up."""
    if image.dtype != np.float32:
        raise ValueError("PFM images must be float32")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = b"PF\n"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = b"Pf\n"
    else:
        raise ValueError("PFM image must be H x W, H x W x 1 or H x W x 3")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    header = magic + f"{image.shape[1]} {image.shape[0]}\n".encode() + ("%f\n" % (-scale if little else scale)).encode()
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    with open(filename, "wb") as f:
        f.write(header)
        np.ascontiguousarray(image[::-1]).tofile(f)


def read_pfm(filename: str) -> Tuple[np.ndarray, float]:
    """data_io.py:6-42 -> (H x W x C float32 array, scale)."""
    with open(filename, "rb") as f:
        magic = f.readline().decode("utf-8").rstrip()
        if magic not in ("PF", "Pf"):
            raise ValueError("not a PFM file")
        channels = 3 if magic == "PF" else 1
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError("malformed PFM header")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f4")
    return np.flipud(data.reshape(height, width, channels)).copy(), abs(scale)
