// itermvs_fuse_depth: geometric + photometric filter of one reference view against its source views --
// reproject_with_depth (eval.py:154-194), check_geometric_consistency (eval.py:197-212) and the per-reference
// arithmetic of filter_depth (eval.py:238-269) in ONE pass: a thread owns a reference pixel, walks the S source
// views and keeps the consistency count and the depth sum in registers; no per-view reprojection maps or masks
// ever reach memory (the reference materialises six [H,W] float64/float32 arrays per pair).
// Arithmetic mirrors numpy's promotion rules in the reference: pixel grid int64, depths float32, camera matrices
// float32 (inverted / composed on the host exactly like eval.py does), every product with a point in float64;
// cv2.remap(INTER_LINEAR) as published (1/32-pixel coordinates, float32 weights, zero border).  oracle/fusion_oracle.py
// is the CPU restatement the tests compare against, bit for bit.
#include <math.h>

#include "common.hpp"

namespace itermvs {

struct FuseArgs {
    const float* depth_ref;
    const float* conf_ref;
    const float* depth_src[ITERMVS_MAX_SRC];
    const float* mats;      // [S][60]: a_ref(9) t_rs(12, rows 0..2) k_src(9) a_src(9) t_sr(12) k_ref(9), float32
    double* depth_avg;
    uint8_t* photo_mask;
    uint8_t* geo_mask;
    uint8_t* final_mask;
    int32_t* geo_sum;
    int S, H, W, geo_mask_thres;
    double geo_pixel_thres;
    float geo_depth_thres, photo_thres;
};

__device__ __forceinline__ void mat3(const float* __restrict__ m, double x, double y, double z, double& ox, double& oy,
                                     double& oz) {
    ox = ((double)m[0] * x + (double)m[1] * y) + (double)m[2] * z;
    oy = ((double)m[3] * x + (double)m[4] * y) + (double)m[5] * z;
    oz = ((double)m[6] * x + (double)m[7] * y) + (double)m[8] * z;
}
__device__ __forceinline__ void mat34(const float* __restrict__ m, double x, double y, double z, double& ox, double& oy,
                                      double& oz) {
    ox = (((double)m[0] * x + (double)m[1] * y) + (double)m[2] * z) + (double)m[3] * 1.0;
    oy = (((double)m[4] * x + (double)m[5] * y) + (double)m[6] * z) + (double)m[7] * 1.0;
    oz = (((double)m[8] * x + (double)m[9] * y) + (double)m[10] * z) + (double)m[11] * 1.0;
}

__device__ __forceinline__ long long remap_fixed(float c) {   // cvRound(c * 32) with saturate_cast<int>
    const double v = (double)c * 32.0;
    if (!isfinite(v)) return -2147483648LL;
    const double r = rint(v);
    return r > 2147483647.0 ? 2147483647LL : (r < -2147483648.0 ? -2147483648LL : (long long)r);
}

__device__ __forceinline__ float remap_bilinear(const float* __restrict__ src, int H, int W, float mx, float my) {
    const long long sx = remap_fixed(mx), sy = remap_fixed(my);
    const long long ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    auto tap = [&](long long yy, long long xx) {
        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? src[yy * W + xx] : 0.0f;
    };
    float out = tap(iy, ix) * (wy0 * wx0);
    out = out + tap(iy, ix + 1) * (wy0 * fx);
    out = out + tap(iy + 1, ix) * (fy * wx0);
    out = out + tap(iy + 1, ix + 1) * (fy * fx);
    return out;
}

__global__ void __launch_bounds__(256) fuse_depth_kernel(const FuseArgs a) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const int p = y * a.W + x;
    const float dref = a.depth_ref[p];
    const double d = (double)dref;
    const double px = (double)x * d, py = (double)y * d;
    int geo = 0;
    float acc = 0.0f;
    for (int s = 0; s < a.S; ++s) {
        const float* __restrict__ m = a.mats + s * 60;
        double rx, ry, rz, sx, sy, sz, kx, ky, kz;
        mat3(m, px, py, d, rx, ry, rz);                    // reference 3-D space      (eval.py:162-163)
        mat34(m + 9, rx, ry, rz, sx, sy, sz);              // source 3-D space         (eval.py:165-166)
        mat3(m + 21, sx, sy, sz, kx, ky, kz);              // source pixel             (eval.py:168-169)
        const double xs = kx / kz, ys = ky / kz;
        const float xs32 = (float)xs, ys32 = (float)ys;
        const double smp = (double)remap_bilinear(a.depth_src[s], a.H, a.W, xs32, ys32);   // eval.py:176
        double qx, qy, qz, wx, wy, wz, jx, jy, jz;
        mat3(m + 30, xs * smp, ys * smp, smp, qx, qy, qz);  // back to source 3-D space (eval.py:181-182)
        mat34(m + 39, qx, qy, qz, wx, wy, wz);             // reference 3-D space      (eval.py:184-185)
        const float drep = (float)wz;
        mat3(m + 51, wx, wy, wz, jx, jy, jz);
        const float xr = (float)(jx / (jz + 1e-6)), yr = (float)(jy / (jz + 1e-6));        // eval.py:189
        const double ddx = (double)xr - (double)x, ddy = (double)yr - (double)y;
        const double dist = sqrt(ddx * ddx + ddy * ddy);                                    // eval.py:203
        const float rel = fabsf(drep - dref) / dref;                                        // eval.py:205-206
        const bool ok = dist < a.geo_pixel_thres && rel < a.geo_depth_thres;
        geo += ok ? 1 : 0;
        acc = acc + (ok ? drep : 0.0f);                                                     // eval.py:209,262
    }
    const float total = acc + dref;
    a.depth_avg[p] = (double)total / (double)(geo + 1);                                    // eval.py:264
    const bool photo = a.conf_ref[p] > a.photo_thres, g = geo >= a.geo_mask_thres;
    if (a.photo_mask) a.photo_mask[p] = photo;
    if (a.geo_mask) a.geo_mask[p] = g;
    if (a.final_mask) a.final_mask[p] = photo && g;
    if (a.geo_sum) a.geo_sum[p] = geo;
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_fuse_depth(const float* depth_ref, const float* conf_ref, const float* const* depth_src,
                                  const float* mats, int32_t S, int32_t H, int32_t W, double geo_pixel_thres,
                                  float geo_depth_thres, float photo_thres, int32_t geo_mask_thres, double* depth_avg,
                                  uint8_t* photo_mask, uint8_t* geo_mask, uint8_t* final_mask, int32_t* geo_sum,
                                  void* stream) {
    ITERMVS_RETURN_IF(!depth_ref || !conf_ref || !depth_src || !mats || !depth_avg, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(H < 1 || W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(S < 1 || S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    FuseArgs a;
    a.depth_ref = depth_ref; a.conf_ref = conf_ref; a.mats = mats;
    for (int s = 0; s < ITERMVS_MAX_SRC; ++s) {
        a.depth_src[s] = s < S ? depth_src[s] : nullptr;
        ITERMVS_RETURN_IF(s < S && !depth_src[s], ITERMVS_ERR_NULL);
    }
    a.depth_avg = depth_avg; a.photo_mask = photo_mask; a.geo_mask = geo_mask; a.final_mask = final_mask;
    a.geo_sum = geo_sum;
    a.S = S; a.H = H; a.W = W; a.geo_mask_thres = geo_mask_thres;
    a.geo_pixel_thres = geo_pixel_thres; a.geo_depth_thres = geo_depth_thres; a.photo_thres = photo_thres;
    hipLaunchKernelGGL(fuse_depth_kernel, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    return itermvs_launch_status();
}
