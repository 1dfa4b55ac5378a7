#!/usr/bin/env python3
"""Can the reference's fp32 ``torch.inverse(ref_proj)`` (models/module.py:81,86) be restated as a FIXED fp32 operation
sequence -- the precondition for a graph-capturable device kernel that reproduces the reference's projection bits?

The reference composes ``src_proj @ inverse(ref_proj)`` in fp32 with whatever LAPACK back-end its PyTorch build carries (MKL
on this host's CPU build, MAGMA / cuSOLVER on a CUDA device): there is no single reference bit pattern.  This probe takes
the 90 camera matrices of the bench's synthetic scenes (6 seeds x 3 levels x 5 views) and compares ``torch.inverse`` on THIS
host against 16 textbook orderings of LU with partial pivoting + two triangular solves against the identity (row-major and
LAPACK's column-major view of the same memory; a*b+c fused or not; division or reciprocal-multiply for the pivots and for
the diagonal of U).  Result on this host (MKL 2024.2): 0 of 90 matrices reproduced bit for bit by any ordering; the closest
family (column-major view, fused multiply-add, reciprocal pivots) deviates by 1e-7 of the inverse's scale -- the size of the
difference between any two LAPACK builds.  MKL's blocked / vectorised kernels do not follow a documented scalar order.
Consequence: `projection="host_fp32"` (read the cameras back, call the host's torch.inverse) stays the only
bit-faithful mode and it is faithful to ONE host; the capturable default composes in fp64 on the device and rounds once,
which moves no sampling position by more than 2.1e-4 px (tests/test_kernels_gpu.py::test_compose_proj_tap_indices...).

    python tools/inverse_order_probe.py > profiles/r04/r04_inverse_order_probe.txt
"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itermvs_amd import synthetic  # noqa: E402

f32 = np.float32


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def getrf(a, use_fma, recip):
    a = a.copy()
    piv = list(range(4))
    for j in range(4):
        p = j + int(np.argmax(np.abs(a[j:, j])))
        if p != j:
            a[[j, p]] = a[[p, j]]
            piv[j], piv[p] = piv[p], piv[j]
        r = f32(1) / a[j, j]
        for i in range(j + 1, 4):
            a[i, j] = f32(a[i, j] * r) if recip else f32(a[i, j] / a[j, j])
        for i in range(j + 1, 4):
            for k in range(j + 1, 4):
                a[i, k] = fma(-a[i, j], a[j, k], a[i, k]) if use_fma else f32(a[i, k] - f32(a[i, j] * a[j, k]))
    return a, piv


def solve_identity(lu, piv, use_fma, recip_u):
    x = np.eye(4, dtype=f32)[piv]
    for c in range(4):
        for i in range(4):
            s = x[i, c]
            for k in range(i):
                s = fma(-lu[i, k], x[k, c], s) if use_fma else f32(s - f32(lu[i, k] * x[k, c]))
            x[i, c] = s
        for i in reversed(range(4)):
            s = x[i, c]
            for k in range(i + 1, 4):
                s = fma(-lu[i, k], x[k, c], s) if use_fma else f32(s - f32(lu[i, k] * x[k, c]))
            x[i, c] = f32(s * (f32(1) / lu[i, i])) if recip_u else f32(s / lu[i, i])
    return x


def main():
    torch.set_num_threads(1)
    mats = []
    for seed in range(6):
        s = synthetic.make_sample(1, 5, 512, 640, seed=seed)
        for l in (1, 2, 3):
            mats += [s["proj_matrices"][f"level_{l}"][0, v].numpy().astype(f32) for v in range(5)]
    ref = [torch.inverse(torch.from_numpy(m)).numpy() for m in mats]
    print(torch.__config__.show().split("\n")[3].strip())
    for colmajor, use_fma, recip, recip_u in itertools.product([0, 1], [0, 1], [0, 1], [0, 1]):
        exact, worst = 0, 0.0
        for m, r in zip(mats, ref):
            a = np.ascontiguousarray(m.T) if colmajor else m
            lu, piv = getrf(a, use_fma, recip)
            x = solve_identity(lu, piv, use_fma, recip_u)
            x = x.T if colmajor else x
            exact += int(np.array_equal(x, r))
            worst = max(worst, float(np.max(np.abs(x - r)) / np.abs(r).max()))
        print(f"column-major view {colmajor}  fma {use_fma}  reciprocal pivots {recip}  reciprocal diagonal {recip_u}:  "
              f"bit-exact {exact} / {len(mats)},  worst |difference| / scale {worst:.2e}")


if __name__ == "__main__":
    main()
