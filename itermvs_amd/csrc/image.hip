// Input side of the path: image normalisation, resize to the inference size and the
// four-level image pyramid of datasets/dtu_yao_eval.py:61-74 (read_img), on the GPU, from the decoded uint8 RGB image.
//
//   np_img   = 2 * uint8 / 255. - 1                                   (float32, op by op)
//   level_0  = cv2.resize(np_img, (W, H), INTER_LINEAR)               half-pixel centres, no anti-aliasing:
//                fx = (dx + 0.5) * (Ws / W) - 0.5; sx = floor(fx); fx -= sx; borders clamp with weight 0;
//                horizontal pass first (S[sx] * (1 - fx) + S[sx + 1] * fx), then vertical, float32
//   level_l  = cv2.resize(level_0, (W >> l, H >> l), INTER_LINEAR)    for exact powers of two this is the mean of the
//                CENTRAL 2 x 2 pixels of every 2^l x 2^l block (fx = fy = 0.5)
// cv2 is not in this image: the resize follows the published algorithm (parity unpinned for this stage, like the filter).
// Uploading the uint8 image (5.8 MB for a 1600 x 1200 DTU view) instead of the float32 pyramid (29.5 MB) also cuts the
// host-to-device traffic of a depth map five-fold.
#include "common.hpp"

namespace itermvs {

__device__ __forceinline__ void resize_axis(int d, double scale, int n_src, int& s0, int& s1, float& a0, float& a1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= n_src - 1) { f = 0.0f; s = n_src - 1; }
    s0 = s;
    s1 = s + 1 < n_src ? s + 1 : n_src - 1;
    a0 = 1.0f - f;
    a1 = f;
}

__device__ __forceinline__ float normalise_u8(uint8_t v) {   // 2 * x / 255. - 1, float32 op by op
    return (2.0f * (float)v) / 255.0f - 1.0f;
}

// thread per output pixel (v, y, x); src [V,Hs,Ws,3] uint8 interleaved RGB; out [V,3,H,W] float32 planes
__global__ void image_level0_kernel(const uint8_t* __restrict__ src, int V, int Hs, int Ws, int H, int W, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)V * H * W) return;
    const int x = (int)(t % W), y = (int)((t / W) % H), v = (int)(t / ((int64_t)W * H));
    int x0, x1, y0, y1;
    float ax0, ax1, ay0, ay1;
    resize_axis(x, (double)Ws / (double)W, Ws, x0, x1, ax0, ax1);
    resize_axis(y, (double)Hs / (double)H, Hs, y0, y1, ay0, ay1);
    const uint8_t* s = src + (int64_t)v * Hs * Ws * 3;
    const uint8_t* r0 = s + (int64_t)y0 * Ws * 3;
    const uint8_t* r1 = s + (int64_t)y1 * Ws * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = normalise_u8(r0[x0 * 3 + c]) * ax0 + normalise_u8(r0[x1 * 3 + c]) * ax1;   // horizontal pass
        const float bot = normalise_u8(r1[x0 * 3 + c]) * ax0 + normalise_u8(r1[x1 * 3 + c]) * ax1;
        out[(((int64_t)v * 3 + c) * H + y) * W + x] = top * ay0 + bot * ay1;                            // vertical pass
    }
}

// level l = mean of the central 2 x 2 pixels of each 2^l x 2^l block of level 0 (cv2.resize INTER_LINEAR at an exact
// power-of-two ratio): horizontal pass a * 0.5 + b * 0.5, then vertical
__global__ void image_down_kernel(const float* __restrict__ l0, int M, int H, int W, int lvl, float* __restrict__ out) {
    const int h = H >> lvl, w = W >> lvl;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * h * w) return;
    const int x = (int)(t % w), y = (int)((t / w) % h), m = (int)(t / ((int64_t)w * h));
    const int sx = (x << lvl) + (1 << (lvl - 1)) - 1, sy = (y << lvl) + (1 << (lvl - 1)) - 1;
    const float* p = l0 + ((int64_t)m * H + sy) * W + sx;
    const float top = p[0] * 0.5f + p[1] * 0.5f;
    const float bot = p[W] * 0.5f + p[W + 1] * 0.5f;
    out[t] = top * 0.5f + bot * 0.5f;
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_image_pyramid(const uint8_t* src, int32_t V, int32_t Hs, int32_t Ws, int32_t H, int32_t W,
                                     float* level0, float* level1, float* level2, float* level3, void* stream) {
    ITERMVS_RETURN_IF(!src || !level0, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(V < 1 || Hs < 1 || Ws < 1 || H < 1 || W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((level1 || level2 || level3) && ((H | W) & 7), ITERMVS_ERR_DIMS);
    const int64_t n0 = (int64_t)V * H * W;
    hipLaunchKernelGGL(image_level0_kernel, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, V, Hs,
                       Ws, H, W, level0);
    float* lv[3] = {level1, level2, level3};
    for (int l = 1; l <= 3; ++l) {
        if (!lv[l - 1]) continue;
        const int64_t n = (int64_t)V * 3 * (H >> l) * (W >> l);
        hipLaunchKernelGGL(image_down_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, level0,
                           V * 3, H, W, l, lv[l - 1]);
    }
    return itermvs_launch_status();
}
