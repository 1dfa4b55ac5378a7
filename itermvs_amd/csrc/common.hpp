// Shared device helpers for libitermvs_hip.so (gfx950 / CDNA4, wave64).
// Built with -ffp-contract=off: every a*b+c below is two roundings unless written as fmaf(),
// so the coordinate math follows the reference's op-by-op fp32 evaluation.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "itermvs_hip.h"

#define ITERMVS_RETURN_IF(cond, code) \
    do {                              \
        if (cond) return (code);      \
    } while (0)

static inline int itermvs_launch_status() {
    return hipGetLastError() == hipSuccess ? ITERMVS_OK : ITERMVS_ERR_LAUNCH;
}

// Results-corrupting knock-out switches (timing-only forms of the kernels: csrc/experiments/*.patch, *.inc) must never reach a
// product build: they exist only in experiment sources, and any of their macros on the command line of a build without
// -DITERMVS_TUNING stops the compilation here.  (The Makefile additionally rejects every -DITERMVS_* it does not know.)
#if !defined(ITERMVS_TUNING) && (defined(ITERMVS_TILE3_KO) || defined(ITERMVS_TILE_KO_BF16) || defined(ITERMVS_SWEEP_KO) || \
                                 defined(ITERMVS_CORRNET_KO) || defined(ITERMVS_HEAD_KO))
#error "knock-out switches (ITERMVS_*_KO*) compute WRONG results: experiment builds only (make TUNING=1 ...)"
#endif

// Tile shapes, persistence and kernel forms are compile-time choices of the shipped library (each one measured, DESIGN.md
// section 4).  Only a library built with `make TUNING=1` (-DITERMVS_TUNING; the sweeps of tools/conv_bench.py) reads the
// ITERMVS_* environment overrides; in the product build this is a constant nullptr and the overrides do not exist.
static inline const char* itermvs_tuning_env(const char* name) {
#ifdef ITERMVS_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- the exact three-term bf16 split of fp32 operands (conv_tile3.hip, corrnet.hip) ----
namespace itermvs {
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
// two fp32 values -> their (h, m, l) bf16 terms, packed [value 0 | value 1 << 16]
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& H, uint32_t& M, uint32_t& L) {
    const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);   // exact
    const uint32_t c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float t0 = r0 - __uint_as_float(c0 & 0xffff0000u), t1 = r1 - __uint_as_float(c1 & 0xffff0000u);   // exact, <= 8 bits
    H = __builtin_amdgcn_perm(b1, b0, 0x07060302u);      // (b0 >> 16) | (b1 & 0xffff0000)
    M = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    L = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
}


// Reductions over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15 of a wave) on the VALU's data-parallel-primitive path:
// quad_perm (lane ^ 1, lane ^ 2), row_half_mirror (quad 0 <-> 1, 2 <-> 3 once the quads are uniform), row_mirror (the two halves).
// Every step combines a lane with a partner that holds the other half of the step's group, so a commutative operation gives all 16
// lanes the bits an xor butterfly (__shfl_xor 1, 2, 4, 8 = four LDS-crossbar round trips) gives them.
template <int CTRL>
__device__ __forceinline__ float row_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int row_dpp(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
constexpr int kDppXor1 = 0xb1, kDppXor2 = 0x4e, kDppHalfMirror = 0x141, kDppMirror = 0x140;   // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int kDppRowShl = 0x100;                                                             // + n: lane l reads lane l + n of its row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, row_dpp<kDppXor1>(v));
    v = fmaxf(v, row_dpp<kDppXor2>(v));
    v = fmaxf(v, row_dpp<kDppHalfMirror>(v));
    return fmaxf(v, row_dpp<kDppMirror>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += row_dpp<kDppXor1>(v);
    v += row_dpp<kDppXor2>(v);
    v += row_dpp<kDppHalfMirror>(v);
    return v + row_dpp<kDppMirror>(v);
}
__device__ __forceinline__ int row16_sum(int v) {
    v += row_dpp<kDppXor1>(v);
    v += row_dpp<kDppXor2>(v);
    v += row_dpp<kDppHalfMirror>(v);
    return v + row_dpp<kDppMirror>(v);
}

}  // namespace itermvs

// compute units of the current device (grid size of persistent kernels)
static inline int itermvs_num_cus() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

// timing hooks (profile.cpp)
void itermvs_profile_begin(int kind, hipStream_t stream);
void itermvs_profile_end(int kind, hipStream_t stream);
void itermvs_profile_cancel();

namespace itermvs {

// Source-map sampling position of one (pixel, hypothesis, view): models/module.py:99-115
// followed by grid_sample's align_corners=True un-normalisation (GridSampler.h:31).
struct WarpGeom {
    float xr, yr;       // W1/W, H1/H                        (module.py:95-96)
    float half_w, half_h;  // (W1-1)/2, (H1-1)/2            (module.py:112-113)
    float w1m1, h1m1;   // W1-1, H1-1
    float gw, gh;       // sample-grid W, H as float         (module.py:106-107)
};

__device__ __forceinline__ WarpGeom make_geom(int W, int H, int W1, int H1) {
    WarpGeom g;
    g.xr = (float)((double)W1 / (double)W);
    g.yr = (float)((double)H1 / (double)H);
    g.half_w = (float)((double)(W1 - 1) / 2.0);
    g.half_h = (float)((double)(H1 - 1) / 2.0);
    g.w1m1 = (float)(W1 - 1);
    g.h1m1 = (float)(H1 - 1);
    g.gw = (float)W;
    g.gh = (float)H;
    return g;
}

// rot @ (xs, ys, 1): k-ordered fma chain like a BLAS dot (module.py:99)
__device__ __forceinline__ void ray_dir(const float* __restrict__ m, float xs, float ys, float& rx,
                                        float& ry, float& rz) {
    rx = fmaf(m[1], ys, m[0] * xs) + m[2];
    ry = fmaf(m[5], ys, m[4] * xs) + m[6];
    rz = fmaf(m[9], ys, m[8] * xs) + m[10];
}

// x / y with one v_rcp_f32 and a one-step residual correction: q = x*r; q += (x - q*y)*r.
// Correctly rounded except for rare half-ulp ties (the IEEE expansion is ~10 instructions and the
// gather kernels are VALU-issue bound); -DITERMVS_EXACT_DIV restores plain IEEE division.
__device__ __forceinline__ float div_rcp(float x, float y, float r) {
#ifdef ITERMVS_EXACT_DIV
    (void)r;
    return x / y;
#else
    const float q = x * r;
    return fmaf(fmaf(-q, y, x), r, q);
#endif
}
__device__ __forceinline__ float rcp_approx(float y) { return __builtin_amdgcn_rcpf(y); }

// same arithmetic as project() below with the three divisors' reciprocals shared / precomputed
struct WarpRcp {
    float r_half_w, r_half_h;
};
__device__ __forceinline__ WarpRcp make_rcp(const WarpGeom& g) {
    WarpRcp r;
    r.r_half_w = (float)(1.0 / (double)g.half_w);
    r.r_half_h = (float)(1.0 / (double)g.half_h);
    return r;
}
__device__ __forceinline__ void project_fast(const WarpGeom& g, const WarpRcp& rc, const float* __restrict__ m, float rx,
                                             float ry, float rz, float d, float& ix, float& iy) {
    float X = rx * d + m[3];
    float Y = ry * d + m[7];
    float Z = rz * d + m[11];
    if (!(Z > 1e-2f)) {  // module.py:105-108
        X = g.gw;
        Y = g.gh;
        Z = 1.0f;
    }
    const float rz1 = rcp_approx(Z);
    const float px = div_rcp(X, Z, rz1);
    const float py = div_rcp(Y, Z, rz1);
    const float gx = div_rcp(px, g.half_w, rc.r_half_w) - 1.0f;   // module.py:112-113
    const float gy = div_rcp(py, g.half_h, rc.r_half_h) - 1.0f;
    ix = ((gx + 1.0f) * 0.5f) * g.w1m1;                           // GridSampler.h:31
    iy = ((gy + 1.0f) * 0.5f) * g.h1m1;
}

// returns the un-normalised source coordinates (ix, iy); `valid` follows module.py:105,110-111
__device__ __forceinline__ void project(const WarpGeom& g, const float* __restrict__ m, float rx, float ry,
                                        float rz, float d, float& ix, float& iy, bool* valid) {
    float X = rx * d + m[3];
    float Y = ry * d + m[7];
    float Z = rz * d + m[11];
    const bool front = Z > 1e-2f;
    if (!front) {  // module.py:105-108: sample-grid W/H, not the source map's
        X = g.gw;
        Y = g.gh;
        Z = 1.0f;
    }
    const float px = X / Z;
    const float py = Y / Z;
    if (valid) *valid = front && px >= 0.0f && px < g.gw && py >= 0.0f && py < g.gh;
    const float gx = px / g.half_w - 1.0f;
    const float gy = py / g.half_h - 1.0f;
    ix = ((gx + 1.0f) * 0.5f) * g.w1m1;
    iy = ((gy + 1.0f) * 0.5f) * g.h1m1;
}

// The four bilinear taps of grid_sample(bilinear, zeros, align_corners=True): clamped integer
// coordinates plus weights already zeroed for out-of-range taps (NaN/inf -> all zero).
struct Taps {
    int x0, x1, y0, y1;
    float nw, ne, sw, se;
};
// what make_taps decided, for itermvs_tap_indices (the diagnostic entry that exposes the fused kernels' tap indices):
// floor(ix), floor(iy) as floats and the validity of the two columns / two rows (bit 0: x0, 1: x1, 2: y0, 3: y1)
struct TapDiag {
    float fx0, fy0;
    int bits;
};

__device__ __forceinline__ Taps make_taps(float ix, float iy, int W1, int H1, TapDiag* diag = nullptr) {
    Taps t;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    const float wmax = (float)(W1 - 1), hmax = (float)(H1 - 1);
    const bool vx0 = fx0 >= 0.0f && fx0 <= wmax;
    const bool vx1 = fx1 >= 0.0f && fx1 <= wmax;
    const bool vy0 = fy0 >= 0.0f && fy0 <= hmax;
    const bool vy1 = fy1 >= 0.0f && fy1 <= hmax;
    const float ax = fx1 - ix, bx = ix - fx0;
    const float ay = fy1 - iy, by = iy - fy0;
    t.nw = (vx0 && vy0) ? ax * ay : 0.0f;
    t.ne = (vx1 && vy0) ? bx * ay : 0.0f;
    t.sw = (vx0 && vy1) ? ax * by : 0.0f;
    t.se = (vx1 && vy1) ? bx * by : 0.0f;
    t.x0 = vx0 ? (int)fx0 : 0;
    t.x1 = vx1 ? (int)fx1 : 0;
    t.y0 = vy0 ? (int)fy0 : 0;
    t.y1 = vy1 ? (int)fy1 : 0;
    if (diag) {
        diag->fx0 = fx0;
        diag->fy0 = fy0;
        diag->bits = (vx0 ? 1 : 0) | (vx1 ? 2 : 0) | (vy0 ? 4 : 0) | (vy1 ? 8 : 0);
    }
    return t;
}

// models/module.py:148-152
__device__ __forceinline__ float unnormalize_depth(float nd, float inv_min, float inv_max) {
    return 1.0f / (inv_max + nd * (inv_min - inv_max));
}

// hypothesis n of the iteration branch around the normalised depth `nd` (itermvs.py:291-293) and plane n of the N initial
// hypotheses (itermvs.py:13-17): ONE definition for the fused forward kernels, their gradients and itermvs_tap_indices
__device__ __forceinline__ float iter_hypothesis(float nd, float off, float inv_min, float inv_max) {
    float ns = nd + off;
    ns = fminf(fmaxf(ns, 0.0f), 1.0f);
    return unnormalize_depth(ns, inv_min, inv_max);
}
__device__ __forceinline__ float init_hypothesis(int n, int N, float inv_min, float inv_max) {
    const float frac = (float)n / (float)(N - 1);
    return 1.0f / (inv_max + frac * (inv_min - inv_max));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// F.interpolate(x, scale_factor=s, 'bilinear') for integer s, optional tanh: output element t of [M, H*s, W*s]
// (bilinear_up_kernel, and the launches that carry it beside another piece of work)
// `interleave` > 0: M = B * interleave maps, stored pixel-major [B, H*s, W*s, interleave] instead of planar
__device__ __forceinline__ void bilinear_up_body(const float* __restrict__ x, int M, int H, int W, int scale, int act,
                                                 float* __restrict__ out, int64_t t, int interleave = 0) {
    const int OH = H * scale, OW = W * scale;
    if (t >= (int64_t)M * OH * OW) return;
    const int ox = (int)(t % OW);
    const int oy = (int)((t / OW) % OH);
    const int m = (int)(t / ((int64_t)OW * OH));
    const float rs = 1.0f / (float)scale;
    float sy = ((float)oy + 0.5f) * rs - 0.5f;
    float sx = ((float)ox + 0.5f) * rs - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > H - 1 ? H - 1 : y0;
    x0 = x0 > W - 1 ? W - 1 : x0;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* xm = x + (size_t)m * H * W;
    const float top = xm[(size_t)y0 * W + x0] * lx0 + xm[(size_t)y0 * W + x1] * lx1;
    const float bot = xm[(size_t)y1 * W + x0] * lx0 + xm[(size_t)y1 * W + x1] * lx1;
    float v = top * ly0 + bot * ly1;
    if (act == 1) v = tanhf(v);
    if (interleave > 0) {
        const int bb = m / interleave, ss = m - bb * interleave;
        out[(((int64_t)bb * OH + oy) * OW + ox) * interleave + ss] = v;
    } else {
        out[t] = v;
    }
}

}  // namespace itermvs
