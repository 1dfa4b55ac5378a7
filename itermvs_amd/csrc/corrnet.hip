// itermvs_corrnet: the whole CorrNet (models/itermvs.py:352-381) in ONE launch.
//
//   c0 = relu(conv3x3(x, 8 -> 8))                    c1 = relu(conv3x3 s2 (c0, 8 -> 16))       c2 = relu(conv3x3 s2 (c1, 16 -> 32))
//   u1 = c1 + deconv3x3 s2 (c2, 32 -> 16)            u0 = c0 + deconv3x3 s2 (u1, 16 -> 8)      y  = conv3x3(u0, 8 -> 1) + bias
//
// As six launches (itermvs_conv2d per layer) one CorrNet costs ~55 us at cfg 1 for ~4 us of arithmetic: five launch
// boundaries and five ramp-ups on maps of 20k pixels.  Here a workgroup owns a 32 x 32 output tile of one map and walks the
// U-Net with every intermediate in LDS; the halo each layer needs is recomputed (x 45 x 45 -> c0 43 x 43 -> c1 21 x 21 ->
// c2 10 x 10 -> u1 18 x 18 -> u0 34 x 34 -> y 32 x 32; 1.5x the multiply-adds of the layer-by-layer form), positions
// outside the image are stored as zeros so every layer sees the zero padding the reference's layer sees.
//
// Arithmetic: fp32 on the vector ALUs as v_pk_fma_f32 -- a thread owns 1 or 2 pixels and 2 .. 8 output channels in
// (channel, channel+1) register pairs; the weights of a (input channel, tap) are wave-uniform, reach the SIMD through the
// scalar cache (s_load) and enter the packed FMA as an SGPR pair, the input value is splat with op_sel.  (The layers with 8
// output channels would waste half of a 16-wide matrix-core tile, and fp32 MFMA has the vector rate on gfx950 anyway.)
#include "common.hpp"

namespace itermvs {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kCnTile = 32;
constexpr int kCnThreads = 512;
constexpr int XS = 45, XP = 46;      // region size / LDS row pitch of x
constexpr int C0S = 43, C0P = 44;    // c0 (later u0 in place, region rows / columns 6 .. 39)
constexpr int C1S = 21, C1P = 22;    // c1 (later u1 in place, 2 .. 19), half resolution
constexpr int C2S = 10, C2P = 11;    // c2, quarter resolution
constexpr int kSzX = 8 * XS * XP, kSzC0 = 8 * C0S * C0P, kSzC1 = 16 * C1S * C1P, kSzC2 = 32 * C2S * C2P;
constexpr int kOffC0 = kSzX, kOffC1 = 0, kOffC2 = kSzC1;      // c1 / c2 reuse the x region once c0 is complete
static_assert(kSzC1 + kSzC2 <= kSzX, "c1 + c2 must fit the x region");
constexpr int kCnLdsFloats = kSzX + kSzC0;
// packed weight set of one CorrNet (floats): see ops.pack_corrnet_weights
constexpr int kW0 = 0;                         // [ci 8][tap 9][co 8]
constexpr int kW1 = kW0 + 8 * 9 * 8;           // [ci 8][tap 9][co 16]
constexpr int kW2 = kW1 + 8 * 9 * 16;          // [ci 16][tap 9][co 32]
constexpr int kW3 = kW2 + 16 * 9 * 32;         // deconv [ci 32][ky 3][kx 3][co 16]
constexpr int kW4 = kW3 + 32 * 9 * 16;         // deconv [ci 16][ky 3][kx 3][co 8]
constexpr int kW5 = kW4 + 16 * 9 * 8;          // [ci 8][tap 9] (co = 1)
constexpr int kWBias = kW5 + 72;
constexpr int kCnWeightFloats = kWBias + 4;

struct CorrNetArgs {
    const float* x;
    const float* w[3];
    int seg_end[3];
    float* out;
    float* out2;
    int64_t x_sn, out_sn, out2_sn;
    int M, H, W, tiles_x;
};

__device__ __forceinline__ v2f splat(float v) { return v2f{v, v}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__global__ void __launch_bounds__(kCnThreads) corrnet_kernel(const CorrNetArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* __restrict__ X = lds;
    float* __restrict__ C0 = lds + kOffC0;
    float* __restrict__ C1 = lds + kOffC1;
    float* __restrict__ C2 = lds + kOffC2;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int m = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int X0 = tx * kCnTile, Y0 = ty * kCnTile;
    const float* __restrict__ wt = a.w[m < a.seg_end[0] ? 0 : (m < a.seg_end[1] ? 1 : 2)];
    const int H = a.H, W = a.W, H2 = H >> 1, W2 = W >> 1, H4 = H >> 2, W4 = W >> 2;

    // ---- x tile (+8 / +12 halo) -> LDS, zeros outside the image and in the pad column ----
    {
        const float* __restrict__ xm = a.x + (int64_t)m * a.x_sn;
        for (int i = tid; i < kSzX; i += kCnThreads) {
            const int ci = i / (XS * XP), r = i - ci * (XS * XP);
            const int ry = r / XP, rx = r - ry * XP;
            const int gy = Y0 - 8 + ry, gx = X0 - 8 + rx;
            const bool ok = rx < XS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            X[i] = ok ? xm[(int64_t)ci * H * W + gy * W + gx] : 0.0f;
        }
    }
    __syncthreads();

    // ---- c0 = relu(conv(x)): 43 x 43, a thread owns 2 neighbouring pixels x 8 channels ----
    for (int item = tid; item < C0S * 22; item += kCnThreads) {
        const int oy = item / 22, ox = (item - oy * 22) * 2;
        v2f acc[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = acc[1][k] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* __restrict__ row = X + (ci * XS + oy + ky) * XP + ox;
                const float in[4] = {row[0], row[1], row[2], row[3]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const v2f* __restrict__ w = reinterpret_cast<const v2f*>(wt + kW0 + (ci * 9 + ky * 3 + kx) * 8);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[0][k] = pk_fma(w[k], splat(in[kx]), acc[0][k]);
                        acc[1][k] = pk_fma(w[k], splat(in[kx + 1]), acc[1][k]);
                    }
                }
            }
        const int gy = Y0 - 7 + oy, gx = X0 - 7 + ox;
        const bool iny = gy >= 0 && gy < H;
        const bool in0 = iny && gx >= 0 && gx < W, in1 = iny && gx + 1 >= 0 && gx + 1 < W && ox + 1 < C0S;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* __restrict__ o = C0 + ((2 * k + h) * C0S + oy) * C0P + ox;
                o[0] = in0 ? fmaxf(acc[0][k][h], 0.0f) : 0.0f;
                o[1] = in1 ? fmaxf(acc[1][k][h], 0.0f) : 0.0f;      // (ox + 1 == 43 lands in the pad column)
            }
    }
    __syncthreads();

    // ---- c1 = relu(conv s2 (c0)): 21 x 21 at half resolution, a thread owns 1 pixel x 16 channels ----
    if (tid < C1S * C1S) {
        const int oy = tid / C1S, ox = tid - oy * C1S;
        v2f acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* __restrict__ row = C0 + (ci * C0S + 2 * oy + ky) * C0P + 2 * ox;
                const float in[3] = {row[0], row[1], row[2]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const v2f* __restrict__ w = reinterpret_cast<const v2f*>(wt + kW1 + (ci * 9 + ky * 3 + kx) * 16);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = pk_fma(w[k], splat(in[kx]), acc[k]);
                }
            }
        const int gy = (Y0 >> 1) - 3 + oy, gx = (X0 >> 1) - 3 + ox;
        const bool ok = gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) C1[((2 * k + h) * C1S + oy) * C1P + ox] = ok ? fmaxf(acc[k][h], 0.0f) : 0.0f;
    }
    __syncthreads();

    // ---- c2 = relu(conv s2 (c1)): 10 x 10 at quarter resolution; waves 2c, 2c+1 own channels 8c .. 8c+7 ----
    {
        const int chunk = wave >> 1, idx = (wave & 1) * 64 + lane;
        if (idx < C2S * C2S) {
            const int oy = idx / C2S, ox = idx - oy * C2S;
            v2f acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = v2f{0.0f, 0.0f};
#pragma unroll 4
            for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* __restrict__ row = C1 + (ci * C1S + 2 * oy + ky) * C1P + 2 * ox;
                    const float in[3] = {row[0], row[1], row[2]};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const v2f* __restrict__ w = reinterpret_cast<const v2f*>(wt + kW2 + (ci * 9 + ky * 3 + kx) * 32 + chunk * 8);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = pk_fma(w[k], splat(in[kx]), acc[k]);
                    }
                }
            const int gy = (Y0 >> 2) - 1 + oy, gx = (X0 >> 2) - 1 + ox;
            const bool ok = gy >= 0 && gy < H4 && gx >= 0 && gx < W4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) C2[((chunk * 8 + 2 * k + h) * C2S + oy) * C2P + ox] = ok ? fmaxf(acc[k][h], 0.0f) : 0.0f;
        }
    }
    __syncthreads();

    // ---- u1 = c1 + deconv(c2): 18 x 18 (c1 rows / columns 2 .. 19), in place.  ConvTranspose2d(3, stride 2, pad 1, out_pad 1):
    //      out[o] = sum_i in[i] w[o - 2i + 1].  A thread owns the 2 x 2 output block (2a, 2b) .. (2a+1, 2b+1) of the region (whose
    //      origin is an odd coordinate) and 4 channels; it needs in[a .. a+1][b .. b+1]:
    //        (2a  , 2b  ) = in11 w00 + in10 w02 + in01 w20 + in00 w22        (2a  , 2b+1) = in11 w01 + in01 w21
    //        (2a+1, 2b  ) = in11 w10 + in10 w12                              (2a+1, 2b+1) = in11 w11
    //      waves 2c, 2c+1 own channels 4c .. 4c+3 ----
    {
        const int chunk = wave >> 1, idx = (wave & 1) * 64 + lane;
        if (idx < 81) {
            const int ba = idx / 9, bb = idx - ba * 9;
            v2f ee[2], eo[2], oe[2], oo[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) ee[k] = eo[k] = oe[k] = oo[k] = v2f{0.0f, 0.0f};
#pragma unroll 4
            for (int ci = 0; ci < 32; ++ci) {
                const float* __restrict__ p = C2 + (ci * C2S + ba) * C2P + bb;
                const float in00 = p[0], in01 = p[1], in10 = p[C2P], in11 = p[C2P + 1];
                const v2f* __restrict__ w = reinterpret_cast<const v2f*>(wt + kW3 + ci * 9 * 16 + chunk * 4);   // [tap][16 co]: tap stride 8 v2f
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    ee[k] = pk_fma(w[0 * 8 + k], splat(in11), ee[k]);
                    ee[k] = pk_fma(w[2 * 8 + k], splat(in10), ee[k]);
                    ee[k] = pk_fma(w[6 * 8 + k], splat(in01), ee[k]);
                    ee[k] = pk_fma(w[8 * 8 + k], splat(in00), ee[k]);
                    eo[k] = pk_fma(w[1 * 8 + k], splat(in11), eo[k]);
                    eo[k] = pk_fma(w[7 * 8 + k], splat(in01), eo[k]);
                    oe[k] = pk_fma(w[3 * 8 + k], splat(in11), oe[k]);
                    oe[k] = pk_fma(w[5 * 8 + k], splat(in10), oe[k]);
                    oo[k] = pk_fma(w[4 * 8 + k], splat(in11), oo[k]);
                }
            }
            const int gy = (Y0 >> 1) - 1 + 2 * ba, gx = (X0 >> 1) - 1 + 2 * bb;
            const bool y0 = gy >= 0 && gy < H2, y1 = gy + 1 >= 0 && gy + 1 < H2;
            const bool x0 = gx >= 0 && gx < W2, x1 = gx + 1 >= 0 && gx + 1 < W2;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float* __restrict__ o = C1 + ((chunk * 4 + 2 * k + h) * C1S + 2 * ba + 2) * C1P + 2 * bb + 2;
                    o[0] = (y0 && x0) ? o[0] + ee[k][h] : 0.0f;
                    o[1] = (y0 && x1) ? o[1] + eo[k][h] : 0.0f;
                    o[C1P] = (y1 && x0) ? o[C1P] + oe[k][h] : 0.0f;
                    o[C1P + 1] = (y1 && x1) ? o[C1P + 1] + oo[k][h] : 0.0f;
                }
        }
    }
    __syncthreads();

    // ---- u0 = c0 + deconv(u1): 34 x 34 (c0 rows / columns 6 .. 39), in place; a thread owns one 2 x 2 block x 8 channels ----
    if (tid < 17 * 17) {
        const int ba = tid / 17, bb = tid - ba * 17;
        v2f ee[4], eo[4], oe[4], oo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ee[k] = eo[k] = oe[k] = oo[k] = v2f{0.0f, 0.0f};
#pragma unroll 2
        for (int ci = 0; ci < 16; ++ci) {
            const float* __restrict__ p = C1 + (ci * C1S + ba + 2) * C1P + bb + 2;
            const float in00 = p[0], in01 = p[1], in10 = p[C1P], in11 = p[C1P + 1];
            const v2f* __restrict__ w = reinterpret_cast<const v2f*>(wt + kW4 + ci * 9 * 8);   // [tap][8 co]: tap stride 4 v2f
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ee[k] = pk_fma(w[0 * 4 + k], splat(in11), ee[k]);
                ee[k] = pk_fma(w[2 * 4 + k], splat(in10), ee[k]);
                ee[k] = pk_fma(w[6 * 4 + k], splat(in01), ee[k]);
                ee[k] = pk_fma(w[8 * 4 + k], splat(in00), ee[k]);
                eo[k] = pk_fma(w[1 * 4 + k], splat(in11), eo[k]);
                eo[k] = pk_fma(w[7 * 4 + k], splat(in01), eo[k]);
                oe[k] = pk_fma(w[3 * 4 + k], splat(in11), oe[k]);
                oe[k] = pk_fma(w[5 * 4 + k], splat(in10), oe[k]);
                oo[k] = pk_fma(w[4 * 4 + k], splat(in11), oo[k]);
            }
        }
        const int gy = Y0 - 1 + 2 * ba, gx = X0 - 1 + 2 * bb;
        const bool y0 = gy >= 0 && gy < H, y1 = gy + 1 >= 0 && gy + 1 < H;
        const bool x0 = gx >= 0 && gx < W, x1 = gx + 1 >= 0 && gx + 1 < W;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* __restrict__ o = C0 + ((2 * k + h) * C0S + 2 * ba + 6) * C0P + 2 * bb + 6;
                o[0] = (y0 && x0) ? o[0] + ee[k][h] : 0.0f;
                o[1] = (y0 && x1) ? o[1] + eo[k][h] : 0.0f;
                o[C0P] = (y1 && x0) ? o[C0P] + oe[k][h] : 0.0f;
                o[C0P + 1] = (y1 && x1) ? o[C0P + 1] + oo[k][h] : 0.0f;
            }
    }
    __syncthreads();

    // ---- y = conv(u0, 8 -> 1) + bias: 32 x 32, a thread owns 2 neighbouring pixels ----
    {
        const int oy = tid >> 4, ox = (tid & 15) * 2;
        v2f acc = splat(wt[kWBias]);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* __restrict__ row = C0 + (ci * C0S + oy + ky + 6) * C0P + ox + 6;
                const float in[4] = {row[0], row[1], row[2], row[3]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = pk_fma(splat(wt[kW5 + ci * 9 + ky * 3 + kx]), v2f{in[kx], in[kx + 1]}, acc);
            }
        const int gy = Y0 + oy, gx = X0 + ox;
        if (gy < H) {
            float* __restrict__ o = a.out + (int64_t)m * a.out_sn + gy * W + gx;
            if (gx < W) o[0] = acc[0];
            if (gx + 1 < W) o[1] = acc[1];
            if (a.out2) {
                float* __restrict__ o2 = a.out2 + (int64_t)m * a.out2_sn + gy * W + gx;
                if (gx < W) o2[0] = acc[0];
                if (gx + 1 < W) o2[1] = acc[1];
            }
        }
    }
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_corrnet(const float* x, int64_t x_sn, const float* const* weights, const int32_t* seg_end, int32_t n_seg,
                               int32_t M, int32_t H, int32_t W, float* out, int64_t out_sn, float* out2, int64_t out2_sn,
                               void* stream) {
    ITERMVS_RETURN_IF(!x || !weights || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(M < 1 || H < 4 || W < 4 || (H & 3) || (W & 3) || n_seg < 1 || n_seg > 3, ITERMVS_ERR_DIMS);
    CorrNetArgs a;
    a.x = x; a.x_sn = x_sn;
    for (int i = 0; i < 3; ++i) {
        const int k = i < n_seg ? i : n_seg - 1;
        ITERMVS_RETURN_IF(!weights[k], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(((uintptr_t)weights[k]) % 16, ITERMVS_ERR_ALIGN);
        a.w[i] = weights[k];
        a.seg_end[i] = (i < n_seg - 1 && seg_end) ? seg_end[i] : M;
    }
    a.out = out; a.out2 = out2; a.out_sn = out_sn; a.out2_sn = out2_sn;
    a.M = M; a.H = H; a.W = W; a.tiles_x = (W + kCnTile - 1) / kCnTile;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(corrnet_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kCnLdsFloats * 4) == hipSuccess;
    ITERMVS_RETURN_IF(!attr_ok, ITERMVS_ERR_LAUNCH);
    const dim3 grid(a.tiles_x * ((H + kCnTile - 1) / kCnTile), M);
    hipLaunchKernelGGL(corrnet_kernel, grid, dim3(kCnThreads), kCnLdsFloats * 4, (hipStream_t)stream, a);
    return itermvs_launch_status();
}
