// itermvs_conv3x3_conv1x1: a 3x3 convolution 32 -> 64 with ReLU followed by a 1x1 convolution 64 -> NO (+ bias) in ONE launch,
// the 64-channel tensor never stored.  The two small heads of IterMVS that have exactly this shape:
//   * the convex up-sampling weights   models/itermvs.py:243-247, 262-263   upsample: conv3x3 32 -> 64, ReLU, conv1x1 64 -> 144
//   * the hidden-state initialisation  models/itermvs.py:153-157, 159-160   hidden_init_head: conv3x3 32 -> 64, ReLU, conv1x1 64 -> 32 + bias
// As two launches (conv_tile3 + conv_mfma / split-k) they cost 13.2 + 10.6 us and 7.5 + 5.2 us at cfg 1 for 1.1 / 0.2 GFLOP:
// launch ramp, weight staging and a 5 MB (1.3 MB) round trip of the hidden tensor on maps of 20 480 (5 120) pixels.
//
// Structure = head_coop_kernel's (head.hip): a persistent workgroup walks 16-pixel row segments and splits each layer over its
// four waves, one wave per SIMD, every wave keeping ITS weight slices in registers for the whole launch:
//   3x3 layer   wave w: output block w (channels 16w .. 16w+15) over both 16-channel input chunks: 9 taps x 2 chunks x 4 steps
//               = 72 v_mfma_f32_16x16x4_f32 (exact fp32), B operands read as ds_read_b128 from the staged tile
//               T[chunk][q][row 3][col 18][s]; ReLU, result to LDS Y[mb][q][l16][r] -- the D layout of the MFMA IS the B layout of
//               the next layer
//   1x1 layer   wave w: output blocks w, w+4, w+8 (< NOB): 16 k-steps each from four ds_read_b128 of Y; bias; stores of 16
//               consecutive pixels per channel (64-byte runs) to the NCHW planes
// The next tile's halo is fetched into registers while the current tile computes (two tile buffers in LDS).
#include <stdlib.h>

#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kS2T = 2 * 4 * 3 * 18 * 4;      // staged tile: [chunk][q][row][col][s] = 1728 floats
constexpr int kS2Y = 4 * 4 * 16 * 4;          // hidden layer: [mb][q][l16][r]
constexpr int kS2MaxOB = 3;                   // output blocks per wave: NO <= 192

struct Stack2Args {
    const float* x;          // [B,32,H,W] planes
    int64_t x_sb;
    const float* w0t;        // 3x3 weights 32 -> 64, tile format [9][2][4][64][4]
    const float* w1p;        // 1x1 weights, [NOB][4][4][16][4]: element (ob, m, q, i, r) = W1[ob*16 + i][m*16 + q*4 + r]
    const float* bias;       // [NOB*16] or nullptr
    float* out;              // [B,NO,H,W] planes
    int64_t out_sb;
    int H, W, tiles_x, NO, NOB;
};

__global__ void __launch_bounds__(256, 2) stack2_coop_kernel(const Stack2Args a, const int tiles_total) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kS2T + kS2Y];
    float* __restrict__ Y = smem + 2 * kS2T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const uint32_t plane = (uint32_t)(a.H * a.W);

    // this wave's weight slices -> registers, once
    f32x4 wc[9][2], w1r[kS2MaxOB][4], bias[kS2MaxOB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
            wc[tap][ch] = *reinterpret_cast<const f32x4*>(a.w0t + ((((tap * 2 + ch) * 4 + q) * 64) + wave * 16 + l16) * 4);
#pragma unroll
    for (int k = 0; k < kS2MaxOB; ++k) {
        const int ob = wave + 4 * k;
        const bool on = ob < a.NOB;          // wave-uniform
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1)
            w1r[k][m1] = on ? *reinterpret_cast<const f32x4*>(a.w1p + ((((ob * 4 + m1) * 4 + q) * 16) + l16) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bias[k] = (on && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + ob * 16 + q * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    // staging: 1728 floats per tile = [32 channels][3 rows][18 columns]; thread t moves items t, t+256, ...; which (channel, row,
    // column) an item is does not depend on the tile
    constexpr int ITEMS = (32 * 54 + 255) / 256;
    float st[ITEMS];
    uint32_t it_plane[ITEMS];
    int it_dy[ITEMS], it_dx[ITEMS], it_lds[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int item = tid + i * 256;
        const int c = item / 54, r = item - c * 54;
        const int row = r / 18, col = r - row * 18;
        it_plane[i] = (uint32_t)c * plane;
        it_dy[i] = item < 32 * 54 ? row - 1 : -(1 << 20);        // surplus items: always out of range
        it_dx[i] = col - 1;
        // channel c = chunk*16 + qq*4 + s  ->  T[chunk][qq][row][col][s]
        it_lds[i] = item < 32 * 54 ? ((((c >> 4) * 4 + ((c >> 2) & 3)) * 3 + row) * 18 + col) * 4 + (c & 3) : -1;
    }
    const int rows_per_b = a.H * a.tiles_x;
    auto fetch = [&](int tile) {
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;
        const float* __restrict__ base = a.x + (int64_t)b * a.x_sb;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int gy = y + it_dy[i], gx = x0 + it_dx[i];
            const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            st[i] = ok ? base[it_plane[i] + (uint32_t)(gy * a.W + gx)] : 0.0f;
        }
    };
    auto stash = [&](float* __restrict__ T) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (it_lds[i] >= 0) T[it_lds[i]] = st[i];
    };

    int tile = blockIdx.x, buf = 0;
    if (tile < tiles_total) {
        fetch(tile);
        stash(smem);
    }
    if (tile + (int)gridDim.x < tiles_total) fetch(tile + gridDim.x);
    for (; tile < tiles_total; tile += gridDim.x, buf ^= 1) {
        __syncthreads();            // this tile's staging is visible; the previous tile's readers of Y are done
        const float* __restrict__ T = smem + buf * kS2T;
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;

        // ---- 3x3 layer: output block `wave` ----
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(T + (((ch * 4 + q) * 3 + ky) * 18 + l16 + kx) * 4);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[tap][ch][s2], bv[s2], acc0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0[r] = fmaxf(acc0[r], 0.0f);
        *reinterpret_cast<f32x4*>(Y + ((wave * 4 + q) * 16 + l16) * 4) = acc0;
        // the next tile's halo (fetched one iteration ago) goes to the other buffer, the one after it into registers
        if (tile + (int)gridDim.x < tiles_total) {
            stash(smem + (buf ^ 1) * kS2T);
            if (tile + 2 * (int)gridDim.x < tiles_total) fetch(tile + 2 * gridDim.x);
        }
        __syncthreads();

        // ---- 1x1 layer: output blocks wave, wave + 4, wave + 8 ----
        f32x4 yv[4];
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1) yv[m1] = *reinterpret_cast<const f32x4*>(Y + ((m1 * 4 + q) * 16 + l16) * 4);
        const int px = x0 + l16;
        float* __restrict__ ob_base = a.out + (int64_t)b * a.out_sb + (size_t)y * a.W + px;
#pragma unroll
        for (int k = 0; k < kS2MaxOB; ++k) {
            const int ob = wave + 4 * k;
            if (ob < a.NOB) {            // wave-uniform
                f32x4 acc1 = bias[k];
#pragma unroll
                for (int m1 = 0; m1 < 4; ++m1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[k][m1][r], yv[m1][r], acc1, 0, 0, 0);
                if (px < a.W) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = ob * 16 + q * 4 + r;
                        if (co < a.NO) ob_base[(size_t)co * plane] = acc1[r];
                    }
                }
            }
        }
    }
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_conv3x3_conv1x1(const float* x, int64_t x_sb, int32_t B, int32_t H, int32_t W, const float* w0_tile,
                                       const float* w1_packed, const float* bias1, int32_t NO, float* out, int64_t out_sb,
                                       void* stream) {
    ITERMVS_RETURN_IF(!x || !w0_tile || !w1_packed || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || H < 1 || W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(NO < 1 || NO > 16 * 4 * kS2MaxOB, ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(((((uintptr_t)w0_tile) | ((uintptr_t)w1_packed) | ((uintptr_t)bias1)) & 15) != 0, ITERMVS_ERR_ALIGN);
    ITERMVS_RETURN_IF((int64_t)32 * H * W >= ((int64_t)1 << 31), ITERMVS_ERR_DIMS);
    Stack2Args a;
    a.x = x; a.x_sb = x_sb; a.w0t = w0_tile; a.w1p = w1_packed; a.bias = bias1; a.out = out; a.out_sb = out_sb;
    a.H = H; a.W = W; a.tiles_x = (W + 15) / 16; a.NO = NO; a.NOB = (NO + 15) / 16;
    const int64_t tiles = (int64_t)a.tiles_x * H * B;
    ITERMVS_RETURN_IF(tiles > 0x7fffffff, ITERMVS_ERR_DIMS);
    const int resident = 2 * itermvs_num_cus();          // one tile shared by the four waves of a persistent workgroup, two per CU
    const int grid = (int)(tiles < resident ? tiles : resident);
    hipLaunchKernelGGL(stack2_coop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, (int)tiles);
    return itermvs_launch_status();
}
