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
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int kS2T = 2 * 4 * 3 * 18 * 4;      // staged tile: [chunk][q][row][col][s] = 1728 floats
constexpr int kS2Y = 4 * 4 * 16 * 4;          // hidden layer: [mb][q][l16][r]
constexpr int kS2MaxOB = 3;                   // output blocks per wave: NO <= 192

struct Stack2Args {
    const float* x;          // [B,32,H,W] planes
    int64_t x_sb;
    const void* w0t;         // 3x3 weights 32 -> 64: fp32 tile format [9][2][4][64][4]; W3: bf16 [wave 4][tap 9][term 3][lane 64][8]
    const void* w1p;         // 1x1 weights: fp32 [NOB][4][4][16][4], element (ob, m, q, i, r) = W1[ob*16 + i][m*16 + q*4 + r];
                             // W3: bf16 [NOB][k group 2][term 3][lane 64][8]
    const float* bias;       // [NOB*16] or nullptr
    float* out;              // [B,NO,H,W] planes
    int64_t out_sb;
    int H, W, tiles_x, NO, NOB;
};

// ---------------------------------------------------------------------------------------------
// W3 (weight_format 3): both layers in the bf16x3 arithmetic of conv_tile3.hip (operands split exactly into three bf16 terms, the
// six largest cross products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).  K = 32 = a lane's eight channels (j / 4) * 16 + 4 q
// + j % 4 of a 32-channel group -- for the 1x1 layer exactly the channels the lane produced as D of the 3x3 layer (like head.hip's
// W2B), for the 3x3 layer the order the tile is staged in: a staging item = (q, row, column) = eight plane loads, split, three
// 16-byte LDS stores into T3[term][q][row][col].  Per tile and wave 54 + <= 36 MFMAs of 16 cycles instead of 72 + <= 48 of 40.
// ---------------------------------------------------------------------------------------------
constexpr int kS3TileB = 3 * 4 * 3 * 18 * 16;         // staged tile as bf16 triples: [term][q][row][col][16 B] = 10 368 B
constexpr int kS3YB = 3 * 2 * 64 * 16;                // hidden layer: [term][k group][lane][16 B]

template <bool W3>
__global__ void __launch_bounds__(256, 2) stack2_coop_kernel(const Stack2Args a, const int tiles_total) {
  if constexpr (W3) {
    // (the l terms of the 3x3 weights -- 36 registers per lane -- and the biases sit in LDS: with them in registers the kernel spills
    //  at the 256 VGPRs two workgroups per CU leave a wave)
    __shared__ __attribute__((aligned(16))) char smem3[2 * kS3TileB + kS3YB + 4 * 9 * 64 * 16 + kS2MaxOB * 4 * 16 * 4];
    char* __restrict__ Y3 = smem3 + 2 * kS3TileB;
    char* __restrict__ WL3 = Y3 + kS3YB;                                   // [wave][tap][lane][16 B]
    float* __restrict__ BS3 = reinterpret_cast<float*>(WL3 + 4 * 9 * 64 * 16);        // [output block NOB <= 12][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const uint32_t plane = (uint32_t)(a.H * a.W);

    // staging: item = (q, row, column) of the 3 x 18 halo tile = the lane-q channels {4q .. 4q+3, 16+4q .. 16+4q+3} of one pixel
    const int it_q = tid / 54, it_r = tid - it_q * 54;
    const int it_row = it_r / 18, it_col = it_r - it_row * 18;
    const bool it_live = tid < 4 * 54;
    float st[8];
    const int rows_per_b = a.H * a.tiles_x;
    auto fetch = [&](int tile) {
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;
        const float* __restrict__ base = a.x + (int64_t)b * a.x_sb;
        const int gy = y + it_row - 1, gx = x0 + it_col - 1;
        const bool ok = it_live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
        for (int j = 0; j < 8; ++j) st[j] = ok ? base[(uint32_t)((j >> 2) * 16 + 4 * it_q + (j & 3)) * plane + (uint32_t)(gy * a.W + gx)] : 0.0f;
    };
    auto stash = [&](char* __restrict__ T) {
        if (it_live) {
            u32x4 Hh, Mm, Ll;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t h, m, l;
                split_pair(st[2 * k], st[2 * k + 1], h, m, l);
                Hh[k] = h; Mm[k] = m; Ll[k] = l;
            }
            char* __restrict__ d = T + ((it_q * 3 + it_row) * 18 + it_col) * 16;
            *reinterpret_cast<u32x4*>(d) = Hh;
            *reinterpret_cast<u32x4*>(d + kS3TileB / 3) = Mm;
            *reinterpret_cast<u32x4*>(d + 2 * (kS3TileB / 3)) = Ll;
        }
    };

    int tile = blockIdx.x, buf = 0;
    if (tile < tiles_total) fetch(tile);            // the first tile's loads go out in front of the weights' (loads return in order)

    // this wave's split weight slices -> registers, once
    bf8 wc[9][2], w1r[kS2MaxOB][2][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wc[tap][pl] = reinterpret_cast<const bf8*>(a.w0t)[((wave * 9 + tap) * 3 + pl) * 64 + lane];
        reinterpret_cast<bf8*>(WL3)[(wave * 9 + tap) * 64 + lane] = reinterpret_cast<const bf8*>(a.w0t)[((wave * 9 + tap) * 3 + 2) * 64 + lane];
    }
    if (tid < kS2MaxOB * 4 * 16) BS3[tid] = (a.bias && tid < a.NOB * 16) ? a.bias[tid] : 0.0f;
#pragma unroll
    for (int k = 0; k < kS2MaxOB; ++k) {
        const int ob = wave + 4 * k;
        const bool on = ob < a.NOB;          // wave-uniform
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                w1r[k][g][pl] = reinterpret_cast<const bf8*>(a.w1p)[(((on ? ob : 0) * 2 + g) * 3 + pl) * 64 + lane];
    }

    if (tile < tiles_total) stash(smem3);
    if (tile + (int)gridDim.x < tiles_total) fetch(tile + gridDim.x);
    for (; tile < tiles_total; tile += gridDim.x, buf ^= 1) {
        __syncthreads();            // this tile's staging is visible; the previous tile's readers of Y are done
        const char* __restrict__ T = smem3 + buf * kS3TileB;
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;

        // ---- 3x3 layer: output block `wave`, all 32 input channels per MFMA; six cross products per tap, small terms first ----
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const char* __restrict__ bp = T + ((q * 3 + ky) * 18 + l16 + kx) * 16;
            const bf8 xh = *reinterpret_cast<const bf8*>(bp), xm = *reinterpret_cast<const bf8*>(bp + kS3TileB / 3);
            const bf8 xl = *reinterpret_cast<const bf8*>(bp + 2 * (kS3TileB / 3));
            const bf8 wl = reinterpret_cast<const bf8*>(WL3)[(wave * 9 + tap) * 64 + lane];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[tap][0], xl, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[tap][1], xm, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[tap][1], xh, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[tap][0], xm, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[tap][0], xh, acc0, 0, 0, 0);
        }
        {   // ReLU, split; channels wave*16 + 4q + r = slots (wave & 1) * 4 + r of k group wave >> 1: 8 bytes per term
            uint32_t h0, m0, l0, h1, m1, l1;
            split_pair(fmaxf(acc0[0], 0.0f), fmaxf(acc0[1], 0.0f), h0, m0, l0);
            split_pair(fmaxf(acc0[2], 0.0f), fmaxf(acc0[3], 0.0f), h1, m1, l1);
            char* __restrict__ d = Y3 + ((wave >> 1) * 64 + lane) * 16 + (wave & 1) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + kS3YB / 3) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(d + 2 * (kS3YB / 3)) = u32x2{l0, l1};
        }
        // the next tile's halo (fetched one iteration ago) goes to the other buffer, the one after it into registers
        if (tile + (int)gridDim.x < tiles_total) {
            stash(smem3 + (buf ^ 1) * kS3TileB);
            if (tile + 2 * (int)gridDim.x < tiles_total) fetch(tile + 2 * gridDim.x);
        }
        __syncthreads();

        // ---- 1x1 layer: output blocks wave, wave + 4, wave + 8 ----
        bf8 yh[2], ym[2], yl[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const char* __restrict__ sy = Y3 + (g * 64 + lane) * 16;
            yh[g] = *reinterpret_cast<const bf8*>(sy);
            ym[g] = *reinterpret_cast<const bf8*>(sy + kS3YB / 3);
            yl[g] = *reinterpret_cast<const bf8*>(sy + 2 * (kS3YB / 3));
        }
        const int px = x0 + l16;
        float* __restrict__ ob_base = a.out + (int64_t)b * a.out_sb + (size_t)y * a.W + px;
#pragma unroll
        for (int k = 0; k < kS2MaxOB; ++k) {
            const int ob = wave + 4 * k;
            if (ob < a.NOB) {            // wave-uniform
                f32x4 acc1 = *reinterpret_cast<const f32x4*>(BS3 + ob * 16 + q * 4);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[k][g][2], yh[g], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[k][g][0], yl[g], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[k][g][1], ym[g], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[k][g][1], yh[g], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[k][g][0], ym[g], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[k][g][0], yh[g], acc1, 0, 0, 0);
                }
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
  } else {
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
            wc[tap][ch] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.w0t) + ((((tap * 2 + ch) * 4 + q) * 64) + wave * 16 + l16) * 4);
#pragma unroll
    for (int k = 0; k < kS2MaxOB; ++k) {
        const int ob = wave + 4 * k;
        const bool on = ob < a.NOB;          // wave-uniform
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1)
            w1r[k][m1] = on ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.w1p) + ((((ob * 4 + m1) * 4 + q) * 16) + l16) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
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
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_conv3x3_conv1x1(const float* x, int64_t x_sb, int32_t B, int32_t H, int32_t W, const void* w0_tile,
                                       const void* w1_packed, int32_t weight_format, const float* bias1, int32_t NO, float* out,
                                       int64_t out_sb, void* stream) {
    ITERMVS_RETURN_IF(!x || !w0_tile || !w1_packed || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(weight_format != 0 && weight_format != 3, ITERMVS_ERR_LAYOUT);
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
    if (weight_format == 3) hipLaunchKernelGGL(stack2_coop_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, (int)tiles);
    else hipLaunchKernelGGL(stack2_coop_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, (int)tiles);
    return itermvs_launch_status();
}
