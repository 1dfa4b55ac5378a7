// itermvs_head_regress: the tail of the depth head fused with the softmax regression.
//   x [B,32,P] = relu(conv3x3(hidden))              models/itermvs.py:121-126 (depth_head[0:2], run by itermvs_conv2d)
//   y = relu(W1 x)            W1 [64,32]  (1x1)     depth_head[2:4]
//   logits = W2 y + b2        W2 [256,64] (1x1)     depth_head[4]
//   nd = window regression of softmax(logits)       models/itermvs.py:171-190 / 201-219 (== itermvs_prob_regress)
// Unfused this is three launches and a 21 MB logits tensor written and read back per GRU iteration.
//
// One wave owns 16 pixels and both GEMMs on v_mfma_f32_16x16x4_f32 (D: lane l holds channels
// mb*16 + (l>>4)*4 + r of pixel l&15):
//   * GEMM 1 (K = 32): input channel of (step group u, k-slot q, step s) is u*16 + q*4 + s, so a lane's four
//     operands of a group are one ds_read_b128 of the packed W1 and four plane loads of x;
//   * GEMM 2 (K = 64) is chained IN REGISTERS: its k-step (mb1, r) takes hidden channel mb1*16 + q*4 + r --
//     exactly accumulator acc1[mb1][r] of the lane that needs it as B operand -- so y never leaves the VGPRs;
//   * each lane ends with 64 of its pixel's 256 logits; max / sum / first-arg-max are combined over the four
//     lanes of a pixel with two xor-shuffles; the nine window bins go through LDS and lane q = 0 evaluates
//     the regression with the arithmetic of prob_regress_kernel (update.hip).
// W1 (8 KB) and W2 (64 KB) sit in LDS, shared by the four waves of a workgroup (64 pixels).
#include <stdlib.h>

#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int kHeadBins = ITERMVS_PROB_BINS;
constexpr int kHeadWin = 2 * ITERMVS_WINDOW_RADIUS + 1;
constexpr int kW1Floats = 4 * 2 * 4 * 16 * 4;      // [mb1][u][q][i][s]
constexpr int kW2Floats = 16 * 4 * 4 * 16 * 4;     // [mb2][mb1][q][i][r]
constexpr int kHeadLds = (kW1Floats + kW2Floats + 4 * kHeadWin * 16) * 4;

struct HeadArgs {
    const float* x;
    int64_t x_sb;
    const float* w1p;
    const float* w2p;
    const float* bias2;
    float* nd0;
    float* nd1;
    int64_t nd_sb0, nd_sb1;
    int64_t* best;
    int P;
};

// GEMM 2 + softmax statistics + first arg-max + window regression for one wave's 16 pixels; `y` = the lane's
// accumulators of the hidden layer after ReLU (channel mb*16 + q*4 + r), `w2` = packed W2 in LDS or global memory.
struct HeadOut {
    float* nd0;
    float* nd1;
    int64_t nd_sb0, nd_sb1;
    int64_t* best;
    int P;
};

__device__ __forceinline__ void head_tail(const f32x4 (&acc1)[4], const float* __restrict__ w2,
                                          const float* __restrict__ bias2, float* __restrict__ win, int wave, int q,
                                          int l16, bool live, int b, int p, const HeadOut& a) {
    // GEMM 2: logits[256] = W2 y + b2; accumulators start from the bias
    f32x4 acc2[16];
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
        const f32x4 bs = *reinterpret_cast<const f32x4*>(bias2 + mb * 16 + q * 4);
        acc2[mb] = bs;
    }
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(w2 + (((mb * 4 + m1) * 4 + q) * 16 + l16) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], acc1[m1][r], acc2[mb], 0, 0, 0);
        }

    // softmax statistics over the pixel's 256 bins: own 64 (bin = mb*16 + q*4 + r), then the four q-lanes
    float m = acc2[0][0];
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc2[mb][r]);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float s = 0.0f, bv = -1.0f;
    int bi = 0;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = expf(acc2[mb][r] - m);
            acc2[mb][r] = e;
            s += e;
            if (e > bv) {          // strict: the lowest own bin wins ties (bins are visited in increasing order)
                bv = e;
                bi = mb * 16 + q * 4 + r;
            }
        }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    // First arg-max of p = e / s (torch.argmax first-max rule on the probabilities, itermvs.py:176).
    // Dividing by the common s is monotonic, so the arg-max of e decides -- unless another bin lies within
    // 2^-22 of the maximum, where the quotients may round to the same float: then (rare, wave-uniform
    // branch) the quotients themselves are compared, exactly like prob_regress_kernel.
    float gm = fmaxf(bv, __shfl_xor(bv, 16));
    gm = fmaxf(gm, __shfl_xor(gm, 32));
    const float thresh = gm * 0.99999976f;
    int close = 0;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) close += acc2[mb][r] >= thresh ? 1 : 0;
    close += __shfl_xor(close, 16);
    close += __shfl_xor(close, 32);
    float bp = bv;
    if (__any(close > 1)) {
        bp = -1.0f;
#pragma unroll
        for (int mb = 0; mb < 16; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pk = acc2[mb][r] / s;
                if (pk > bp) {
                    bp = pk;
                    bi = mb * 16 + q * 4 + r;
                }
            }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float op = __shfl_xor(bp, sh);
        const int oi = __shfl_xor(bi, sh);
        if (op > bp || (op == bp && oi < bi)) {
            bp = op;
            bi = oi;
        }
    }
    // window k*-4 .. k*+4 (unclamped positions) -> LDS, as e; lane q == 0 divides and regresses
    const int lo = bi - ITERMVS_WINDOW_RADIUS;
    float* __restrict__ wv = win + wave * (kHeadWin * 16);
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = mb * 16 + q * 4 + r - lo;
            if (off >= 0 && off < kHeadWin) wv[off * 16 + l16] = acc2[mb][r];
        }
    __syncthreads();
    if (q == 0 && live) {
        float num = 0.0f, den = 1e-6f;   // itermvs.py:212
        for (int i = 0; i < kHeadWin; ++i) {
            int k = lo + i;
            k = k < 0 ? 0 : (k > kHeadBins - 1 ? kHeadBins - 1 : k);   // clamp; duplicates double-counted
            const float pk = wv[(k - lo) * 16 + l16] / s;
            num = num + (float)k * pk;
            den = den + pk;
        }
        const float nd = (num / den) / (float)(kHeadBins - 1);
        if (a.nd0) a.nd0[b * a.nd_sb0 + p] = nd;
        if (a.nd1) a.nd1[b * a.nd_sb1 + p] = nd;
        if (a.best) a.best[(size_t)b * a.P + p] = bi;
    }
}

__global__ void __launch_bounds__(256) head_regress_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ w1 = smem;
    float* __restrict__ w2 = smem + kW1Floats;
    float* __restrict__ win = smem + kW1Floats + kW2Floats;      // [wave][9][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l16 = lane & 15;
    const int b = blockIdx.y;
    const int p = (blockIdx.x * 4 + wave) * 16 + l16;
    const bool live = p < a.P;

    // x operands first (plane loads, out of range -> 0), then the weight copy: both are in flight together
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (int64_t)b * a.x_sb), 0, (int)(32u * (uint32_t)a.P * 4u), 0x00020000);
    const uint32_t xoff = live ? ((uint32_t)(q * 4) * (uint32_t)a.P + (uint32_t)p) * 4u : 0x7fffffffu;
    float xv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            xv[u][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xoff, (uint32_t)(u * 16 + s) * (uint32_t)a.P * 4u, 0));
    {
        const f32x4* __restrict__ s1 = reinterpret_cast<const f32x4*>(a.w1p);
        const f32x4* __restrict__ s2 = reinterpret_cast<const f32x4*>(a.w2p);
        f32x4 t[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) t[i] = s1[tid + i * 256];
#pragma unroll
        for (int i = 0; i < 2; ++i) reinterpret_cast<f32x4*>(w1)[tid + i * 256] = t[i];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = s2[tid + (h * 8 + i) * 256];
#pragma unroll
            for (int i = 0; i < 8; ++i) reinterpret_cast<f32x4*>(w2)[tid + (h * 8 + i) * 256] = t[i];
        }
    }
    __syncthreads();

    // GEMM 1: y[64] = relu(W1 x)
    f32x4 acc1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc1[mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(w1 + (((mb * 2 + u) * 4 + q) * 16 + l16) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], xv[u][s], acc1[mb], 0, 0, 0);
        }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[mb][r] = fmaxf(acc1[mb][r], 0.0f);

    HeadOut o;
    o.nd0 = a.nd0; o.nd1 = a.nd1; o.nd_sb0 = a.nd_sb0; o.nd_sb1 = a.nd_sb1; o.best = a.best; o.P = a.P;
    head_tail(acc1, w2, a.bias2, win, wave, q, l16, live, b, p, o);
}

// ---------------------------------------------------------------------------------------------
// itermvs_head_fused: the WHOLE depth head in one launch -- its dilated 3x3 layer (32 -> 32, ReLU), the two 1x1 layers and
// the regression above (head_coop_kernel below; a one-tile-per-wave form measured 27.8 vs 20 us and was removed).
// ---------------------------------------------------------------------------------------------
struct FusedArgs {
    const float* hidden;     // [B,32,H,W] planes
    int64_t h_sb;
    const void* w0t;         // 3x3 weights: fp32 tile format [9][2][4][32][4]; C3: bf16 [block 2][tap 9][term 3][lane 64][8]
    const float* w1p;
    const void* w2p;         // W2B: bf16 [16][2][3][64][8] (ops.pack_head_w2_split3), else fp32 packed like itermvs_head_regress
    const float* bias2;
    HeadOut out;
    int H, W, tiles_x;
    // confidence head riding in the same launch (head_coop_kernel<true>, itermvs.py:147-151,197-199): its dilated 3x3 layer reads
    // the same staged tile of `hidden`
    const float* wct;        // 3x3 weights 32 -> 32, tile format like w0t
    const float* cdot;       // 1x1 layer: 32 weights + bias
    float* conf;             // [B,1,H,W]
    int64_t conf_sb;
};

// ---------------------------------------------------------------------------------------------
// itermvs_head_fused, cooperative form: ONE 16-pixel tile is shared by the four waves of a workgroup.
//
// A form with one whole tile per wave (432 MFMAs) ran 1280 tiles on 1024 SIMDs at cfg 1, so a quarter of the SIMDs ran
// two tiles back to back while the others waited -- 27 us for 6.6 us of matrix work per tile.
// Here a persistent workgroup walks tiles (row segments of 16 pixels) and splits each layer over its four waves, one
// wave per SIMD, so every SIMD of a CU carries the same 108 MFMAs per tile:
//   3x3 dilated conv 32 -> 32   wave (mb0 = w & 1, ch = w >> 1): output block mb0 over input chunk ch, 9 taps x 4 steps;
//                               the two chunk partials meet in LDS (P)
//   1x1 32 -> 64 (+ReLU)        wave w: output block w (8 MFMAs), result to LDS (Y)
//   1x1 64 -> 256               wave w: bins 64w .. 64w+63 (64 MFMAs)
//   softmax / first arg-max / window regression: statistics combined over the 4 q-lanes (xor shuffles) and the 4 waves
//                               (LDS), same arithmetic and tie rule as head_tail above
// Every wave keeps ITS weight slices in registers for the whole launch (36 + 8 + 64 values per lane, loaded once): no
// per-tile weight traffic at all.  The next tile's hidden-state halo (3 rows x 20 columns x 32 channels) is fetched into
// registers while the current tile computes.
// LDS exchange layouts are [..][q][l16][4]: a lane writes / reads one ds_*_b128 at its own (q, l16) position, because the
// D layout of v_mfma_f32_16x16x4_f32 (channel q*4 + r, pixel l16) is exactly the B layout the next layer needs.
// ---------------------------------------------------------------------------------------------
constexpr int kCoT = 2 * 4 * 3 * 20 * 4;   // staged tile: [chunk][q][row][col][s]
constexpr int kCoP = 2 * 2 * 4 * 16 * 4;   // conv partials: [chunk][mb0][q][l16][r]
constexpr int kCoY = 3 * 2 * 64 * 4;       // hidden layer: fp32 [mb1][q][l16][r] (1024 floats), or -- W2B -- its three bf16 terms as B
                                           // operands of the 64 -> 256 layer: [plane h,m,l][k group][lane][8 bf16]
constexpr int kCoLgStride = 256 + 4;       // logits [16 pixels][256 bins], rows padded against bank conflicts
constexpr int kCoLds = 2 * kCoT + kCoP + kCoY + 16 * kCoLgStride + kHeadBins;      // (+ the 256 biases of the last layer)
constexpr int kCoWc = 9 * 2 * 4 * 32 * 4;  // the confidence head's 3x3 weights (CONF form): [tap][chunk][q][co 32][s]
static_assert(kCoP <= 16 * kCoLgStride, "the confidence partials alias the logits buffer");

// CONF: the confidence head (dilated 3x3 32 -> 32, ReLU, 1x1 -> 1, sigmoid: itermvs.py:147-151,198) evaluated on the same staged
// tile -- the last GRU iteration's launch carries it instead of a launch of its own (11.5 us at cfg 1).  Its 3x3 weights sit in
// LDS (37 KB; the depth head's fill the wave's registers), its two chunk partials use the logits buffer, which is idle until the
// 64 -> 256 layer is done; wave 0 folds ReLU, the 1x1 layer and the sigmoid into the phase of the depth head's first 1x1 layer.
// W2B: the 64 -> 256 layer (64 of a wave's 108 fp32 matrix instructions per tile, 32 cycles each) in the bf16x3 form of conv_tile3.hip:
// the hidden layer's ReLU results are split exactly into three bf16 terms by the wave that produces them (8 bytes per lane and
// plane), the wave's W2 slice -- split on the host -- sits in registers as A operands of v_mfma_f32_16x16x32_bf16 (K = 32 = the
// lane's own eight channels (2g + j/4) * 16 + 4q + j%4 of k group g: the D layout of the 32 -> 64 layer IS this B layout), and
// the six largest cross products per k group replace 32 fp32 instructions: 48 x 16 instead of 64 x 32 matrix-pipe cycles.
// C3 (w0_format 3, with W2B, without CONF): the dilated 3x3 layer in bf16x3 as well.  K = 32 = ALL 32 input channels of a tap (a
// lane's eight slots j = channels (j / 4) * 16 + 4 q + j % 4), so the four waves split as (output block w & 1, taps 0..4 | 5..8)
// instead of (output block, input chunk): 30 / 24 MFMAs of 16 cycles instead of 36 of 40; the tile is staged as bf16 triples by
// (q, row, column) items (eight plane loads, split, three 16-byte stores), the h and m terms of a wave's weights sit in registers,
// the l terms in LDS (the kernel is at its register limit).  The two tap-half partials meet in LDS like the two chunk partials.
constexpr int kC3T = 3 * 4 * 3 * 20 * 4;       // staged tile as bf16 triples, in floats: [term][q][row][col][16 B] = 11 520 B
constexpr int kC3PlaneB = 4 * 3 * 20 * 16;
constexpr int kC3WlB = 4 * 5 * 64 * 16;        // l terms of the 3x3 weights: [wave][tap of the wave <= 5][lane][16 B]

template <bool CONF, bool W2B, bool C3 = false>
__global__ void __launch_bounds__(256, 2) head_coop_kernel(const FusedArgs a, const int tiles_total) {
    static_assert(!C3 || (W2B && !CONF), "bf16x3 3x3 layer: with the bf16x3 last layer, without the confidence rider");
    constexpr int kTT = C3 ? kC3T : kCoT;           // floats per tile buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ Pp = smem + 2 * kTT;        // (two tile buffers in front: the next tile is stashed while this one computes)
    float* __restrict__ Y = Pp + kCoP;
    float* __restrict__ LG = Y + kCoY;
    float* __restrict__ PC = LG;                    // CONF: conv partials of the confidence head
    float* __restrict__ BS = LG + 16 * kCoLgStride; // the last layer's biases (accumulator start values, re-read per tile: 16 registers less)
    float* __restrict__ WC = smem + kCoLds;         // CONF: its 3x3 weights
    char* __restrict__ WL3 = reinterpret_cast<char*>(BS + kHeadBins);      // C3: l terms of this layer's weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int mb0 = wave & 1, ch = wave >> 1;
    const uint32_t plane = (uint32_t)(a.H * a.W);

    // this wave's weight slices -> registers, once
    f32x4 wc[C3 ? 1 : 9], w1r[2], w2r[W2B ? 1 : 4][W2B ? 1 : 4];
    bf8 w2b[W2B ? 4 : 1][W2B ? 2 : 1][W2B ? 3 : 1];
    bf8 w3h[C3 ? 5 : 1], w3m[C3 ? 5 : 1];           // C3: terms h, m of taps ch * 5 + k (the second half has four)
    if constexpr (C3) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int tap = min(ch * 5 + k, 8);
            const bf8* __restrict__ src = reinterpret_cast<const bf8*>(a.w0t) + ((mb0 * 9 + tap) * 3) * 64 + lane;
            w3h[k] = src[0];
            w3m[k] = src[64];
            reinterpret_cast<bf8*>(WL3)[(wave * 5 + k) * 64 + lane] = src[128];
        }
    } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            wc[tap] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.w0t) + ((((tap * 2 + ch) * 4 + q) * 32) + mb0 * 16 + l16) * 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) w1r[u] = *reinterpret_cast<const f32x4*>(a.w1p + ((((wave * 2 + u) * 4 + q) * 16) + l16) * 4);
    if constexpr (W2B) {
#pragma unroll
        for (int mbl = 0; mbl < 4; ++mbl)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    w2b[mbl][g][pl] = reinterpret_cast<const bf8*>(a.w2p)[(((wave * 4 + mbl) * 2 + g) * 3 + pl) * 64 + lane];
    } else {
#pragma unroll
        for (int mbl = 0; mbl < 4; ++mbl)
#pragma unroll
            for (int m1 = 0; m1 < 4; ++m1)
                w2r[mbl][m1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.w2p) + (((((wave * 4 + mbl) * 4 + m1) * 4 + q) * 16) + l16) * 4);
    }
    BS[tid] = a.bias2[tid];                         // (256 threads, 256 bins; visible after the loop's first barrier)
    if constexpr (CONF) {
        f32x4 t[kCoWc / 4 / 256];
#pragma unroll
        for (int i = 0; i < kCoWc / 4 / 256; ++i) t[i] = reinterpret_cast<const f32x4*>(a.wct)[tid + i * 256];
#pragma unroll
        for (int i = 0; i < kCoWc / 4 / 256; ++i) reinterpret_cast<f32x4*>(WC)[tid + i * 256] = t[i];
        if (tid < 33) WC[kCoWc + tid] = a.cdot[tid];      // the 1x1 layer {w[32], bias} behind the 3x3 weights
    }

    const int rows_per_b = a.H * a.tiles_x;
    // C3 staging: item = (q, row, column) of the 3 x 20 halo tile = the lane-q channels {4q .. 4q+3, 16+4q .. 16+4q+3} of one pixel
    const int c3_q = tid / 60, c3_r = tid - c3_q * 60;
    const int c3_row = c3_r / 20, c3_col = c3_r - c3_row * 20;
    float st3[C3 ? 8 : 1];
    auto fetch3 = [&](int tile) {
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;
        const float* __restrict__ base = a.hidden + (int64_t)b * a.h_sb;
        const int gy = y + 2 * c3_row - 2, gx = x0 + c3_col - 2;
        const bool ok = tid < 240 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
        for (int j = 0; j < (C3 ? 8 : 1); ++j)
            st3[j] = ok ? base[(uint32_t)((j >> 2) * 16 + 4 * c3_q + (j & 3)) * plane + (uint32_t)(gy * a.W + gx)] : 0.0f;
    };
    auto stash3 = [&](float* __restrict__ T) {
        if constexpr (C3) {
            if (tid < 240) {
                u32x4 Hh, Mm, Ll;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t h, m, l;
                    split_pair(st3[2 * k], st3[2 * k + 1], h, m, l);
                    Hh[k] = h; Mm[k] = m; Ll[k] = l;
                }
                char* __restrict__ d = reinterpret_cast<char*>(T) + ((c3_q * 3 + c3_row) * 20 + c3_col) * 16;
                *reinterpret_cast<u32x4*>(d) = Hh;
                *reinterpret_cast<u32x4*>(d + kC3PlaneB) = Mm;
                *reinterpret_cast<u32x4*>(d + 2 * kC3PlaneB) = Ll;
            }
        }
    };
    // staging: 1920 floats per tile = [32 channels][3 rows][20 columns]; thread t moves items t, t+256, ...  Which (channel,
    // row, column) an item is does not depend on the tile: its plane offset, row / column displacement and LDS slot are
    // computed once.
    constexpr int ITEMS = (32 * 60 + 255) / 256;
    float st[ITEMS];
    // (two registers per item: the plane offset, and LDS slot | column << 12 | row << 17 | live << 19 packed)
    uint32_t it_plane[ITEMS], it_meta[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int item = tid + i * 256;
        const int c = item / 60, r = item - c * 60;
        const int row = r / 20, col = r - row * 20;
        it_plane[i] = (uint32_t)c * plane;
        // channel c = chunk*16 + qq*4 + s  ->  T[chunk][qq][row][col][s]
        const int lds = ((((c >> 4) * 4 + ((c >> 2) & 3)) * 3 + row) * 20 + col) * 4 + (c & 3);
        it_meta[i] = item < 32 * 60 ? (uint32_t)lds | (uint32_t)col << 12 | (uint32_t)row << 17 | 1u << 19 : 0u;      // surplus items: dead
    }
    auto fetch = [&](int tile) {
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;
        const float* __restrict__ base = a.hidden + (int64_t)b * a.h_sb;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int gy = y + 2 * (int)((it_meta[i] >> 17) & 3u) - 2, gx = x0 + (int)((it_meta[i] >> 12) & 31u) - 2;
            const bool ok = (it_meta[i] >> 19) != 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            st[i] = ok ? base[it_plane[i] + (uint32_t)(gy * a.W + gx)] : 0.0f;
        }
    };
    auto stash = [&](float* __restrict__ T) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (it_meta[i] >> 19) T[it_meta[i] & 0xfffu] = st[i];
    };

    auto fetch_tile = [&](int t) { if constexpr (C3) fetch3(t); else fetch(t); };
    auto stash_tile = [&](float* __restrict__ T) { if constexpr (C3) stash3(T); else stash(T); };
    int tile = blockIdx.x, buf = 0;
    if (tile < tiles_total) {
        fetch_tile(tile);
        stash_tile(smem);
    }
    if (tile + (int)gridDim.x < tiles_total) fetch_tile(tile + gridDim.x);
    for (; tile < tiles_total; tile += gridDim.x, buf ^= 1) {
        __syncthreads();            // this tile's staging is visible; the previous tile's readers of P / Y / LG are done
        const float* __restrict__ T = smem + buf * kTT;
        const int b = tile / rows_per_b, rem = tile - b * rows_per_b;
        const int y = rem / a.tiles_x, x0 = (rem - y * a.tiles_x) * 16;

        // ---- dilated 3x3 layer: block mb0, input chunk ch (C3: block mb0, tap half ch, all 32 input channels) ----
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* __restrict__ tb = T + ((ch * 4 + q) * 3) * 80 + l16 * 4;
        if constexpr (C3) {
            const char* __restrict__ t3 = reinterpret_cast<const char*>(T) + (q * 3 * 20 + l16) * 16;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (k < 4 || ch == 0) {          // wave-uniform: taps 0..4 | 5..8
                    const int tap = ch * 5 + k;      // (ch is wave-uniform; ky / kx below are folded per branch of it)
                    const int ky = tap / 3, kx = tap - ky * 3;
                    const char* __restrict__ bp = t3 + (ky * 20 + 2 * kx) * 16;
                    const bf8 xh = *reinterpret_cast<const bf8*>(bp), xm = *reinterpret_cast<const bf8*>(bp + kC3PlaneB);
                    const bf8 xl = *reinterpret_cast<const bf8*>(bp + 2 * kC3PlaneB);
                    const bf8 wl = reinterpret_cast<const bf8*>(WL3)[(wave * 5 + k) * 64 + lane];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3h[k], xl, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3m[k], xm, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3m[k], xh, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3h[k], xm, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3h[k], xh, acc0, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(tb + ky * 80 + kx * 8);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[tap][s2], bv[s2], acc0, 0, 0, 0);
            }
        }
        *reinterpret_cast<f32x4*>(Pp + (((ch * 2 + mb0) * 4 + q) * 16 + l16) * 4) = acc0;
        if constexpr (CONF) {
            f32x4 accc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(tb + ky * 80 + kx * 8);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(WC + ((((tap * 2 + ch) * 4 + q) * 32) + mb0 * 16 + l16) * 4);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) accc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s2], bv[s2], accc, 0, 0, 0);
            }
            *reinterpret_cast<f32x4*>(PC + (((ch * 2 + mb0) * 4 + q) * 16 + l16) * 4) = accc;
        }
        // the next tile's halo (fetched one iteration ago) goes to the other buffer, the one after it into registers
        if (tile + (int)gridDim.x < tiles_total) {
            stash_tile(smem + (buf ^ 1) * kTT);
            if (tile + 2 * (int)gridDim.x < tiles_total) fetch_tile(tile + 2 * gridDim.x);
        }
        __syncthreads();

        // ---- 1x1 layer 32 -> 64: output block `wave`; B = relu(sum of the two chunk partials) ----
        f32x4 acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 pa = *reinterpret_cast<const f32x4*>(Pp + (((0 * 2 + u) * 4 + q) * 16 + l16) * 4);
            const f32x4 pb = *reinterpret_cast<const f32x4*>(Pp + (((1 * 2 + u) * 4 + q) * 16 + l16) * 4);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[u][s2], fmaxf(pa[s2] + pb[s2], 0.0f), acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[r] = fmaxf(acc1[r], 0.0f);
        if constexpr (W2B) {
            // channels wave*16 + 4q + r = slots (wave & 1) * 4 + r of k group wave >> 1: 8 bytes per plane
            uint32_t h0, m0, l0, h1, m1, l1;
            split_pair(acc1[0], acc1[1], h0, m0, l0);
            split_pair(acc1[2], acc1[3], h1, m1, l1);
            char* __restrict__ d = reinterpret_cast<char*>(Y) + (((wave >> 1) * 64 + lane) * 16) + (wave & 1) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * 64 * 16) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(d + 4 * 64 * 16) = u32x2{l0, l1};
        } else {
            *reinterpret_cast<f32x4*>(Y + ((wave * 4 + q) * 16 + l16) * 4) = acc1;
        }
        if constexpr (CONF) {
            if (wave == 0) {        // confidence = sigmoid(sum_c relu(conv)[c] w[c] + b): lane (q, l16) holds channels u*16 + q*4 + r
                float sdot = 0.0f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 cwv = *reinterpret_cast<const f32x4*>(WC + kCoWc + u * 16 + q * 4);
                    const f32x4 pa = *reinterpret_cast<const f32x4*>(PC + (((0 * 2 + u) * 4 + q) * 16 + l16) * 4);
                    const f32x4 pb = *reinterpret_cast<const f32x4*>(PC + (((1 * 2 + u) * 4 + q) * 16 + l16) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sdot = fmaf(fmaxf(pa[r] + pb[r], 0.0f), cwv[r], sdot);
                }
                sdot += __shfl_xor(sdot, 16);
                sdot += __shfl_xor(sdot, 32);
                sdot += WC[kCoWc + 32];
                const int pxc = x0 + l16;
                if (q == 0 && pxc < a.W) a.conf[b * a.conf_sb + y * a.W + pxc] = sigmoidf_(sdot);
            }
        }
        __syncthreads();

        // ---- 1x1 layer 64 -> 256: bins 64*wave .. 64*wave + 63 ----
        f32x4 acc2[4];
#pragma unroll
        for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = *reinterpret_cast<const f32x4*>(BS + (wave * 4 + mbl) * 16 + q * 4);
        if constexpr (W2B) {
            bf8 yh[2], ym[2], yl[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const char* __restrict__ sy = reinterpret_cast<const char*>(Y) + (g * 64 + lane) * 16;
                yh[g] = *reinterpret_cast<const bf8*>(sy);
                ym[g] = *reinterpret_cast<const bf8*>(sy + 2 * 64 * 16);
                yl[g] = *reinterpret_cast<const bf8*>(sy + 4 * 64 * 16);
            }
            // six cross products per k group, small terms first; the four output blocks interleaved (independent accumulators)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2b[mbl][g][2], yh[g], acc2[mbl], 0, 0, 0);
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2b[mbl][g][0], yl[g], acc2[mbl], 0, 0, 0);
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2b[mbl][g][1], ym[g], acc2[mbl], 0, 0, 0);
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2b[mbl][g][1], yh[g], acc2[mbl], 0, 0, 0);
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2b[mbl][g][0], ym[g], acc2[mbl], 0, 0, 0);
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2b[mbl][g][0], yh[g], acc2[mbl], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int m1 = 0; m1 < 4; ++m1) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(Y + ((m1 * 4 + q) * 16 + l16) * 4);
#pragma unroll
                for (int mbl = 0; mbl < 4; ++mbl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc2[mbl] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[mbl][m1][r], yv[r], acc2[mbl], 0, 0, 0);
            }
        }

        // ---- logits -> LDS [pixel][bin]; then wave w owns pixels 4w .. 4w+3 completely (16 lanes x 16 bins each): the
        //      softmax statistics, first arg-max and window regression need no further workgroup barrier ----
#pragma unroll
        for (int mbl = 0; mbl < 4; ++mbl)
            *reinterpret_cast<f32x4*>(LG + l16 * kCoLgStride + (wave * 4 + mbl) * 16 + q * 4) = acc2[mbl];
        __syncthreads();
        const int pl = wave * 4 + (lane >> 4), t = lane & 15;      // pixel of the tile, bin group (bins 16t .. 16t+15)
        float e[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(LG + pl * kCoLgStride + t * 16 + i * 4);
            e[4 * i] = v[0]; e[4 * i + 1] = v[1]; e[4 * i + 2] = v[2]; e[4 * i + 3] = v[3];
        }
        float m = e[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, e[i]);
        m = row16_max(m);             // (the 16 lanes of a pixel are one DPP row)
        float s = 0.0f, bv = -1.0f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            e[i] = __expf(e[i] - m);      // v_exp_f32 (1 ulp, like the vectorised exp of the reference's CPU softmax): 16 per lane
            s += e[i];
            if (e[i] > bv) {          // strict: the lowest own bin wins ties (bins are visited in increasing order)
                bv = e[i];
                bi = t * 16 + i;
            }
        }
        s = row16_sum(s);
        const float gm = row16_max(bv);
        // first arg-max of p = e / s: see head_tail (quotients compared only when another bin lies within 2^-22 of the maximum)
        const float thresh = gm * 0.99999976f;
        int close = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) close += e[i] >= thresh ? 1 : 0;
        close = row16_sum(close);
        float bp = bv;
        if (__any(close > 1)) {         // wave-uniform, rare
            bp = -1.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pk = e[i] / s;
                if (pk > bp) {
                    bp = pk;
                    bi = t * 16 + i;
                }
            }
        }
        auto take_better = [&](float op, int oi) {
            if (op > bp || (op == bp && oi < bi)) {
                bp = op;
                bi = oi;
            }
        };
        take_better(row_dpp<kDppXor1>(bp), row_dpp<kDppXor1>(bi));
        take_better(row_dpp<kDppXor2>(bp), row_dpp<kDppXor2>(bi));
        take_better(row_dpp<kDppHalfMirror>(bp), row_dpp<kDppHalfMirror>(bi));
        take_better(row_dpp<kDppMirror>(bp), row_dpp<kDppMirror>(bi));
        // window k*-4 .. k*+4 (clamped; border duplicates double-counted): lane i < 9 of the pixel's 16 lanes fetches its bin
        // back from LDS as e, the nine terms are then summed in window order by lane 0 (the arithmetic of prob_regress_kernel)
        const int lo = bi - ITERMVS_WINDOW_RADIUS;
        int kbin = lo + t;
        kbin = kbin < 0 ? 0 : (kbin > kHeadBins - 1 ? kHeadBins - 1 : kbin);
        const float pk_mine = __expf(LG[pl * kCoLgStride + kbin] - m) / s;
        float num = 0.0f, den = 1e-6f;   // itermvs.py:212
        float pkw[kHeadWin];            // lane 0 of the pixel's row: the probabilities of lanes 0 .. 8 (row_shl: lane l reads lane l + i)
        pkw[0] = pk_mine;
        pkw[1] = row_dpp<kDppRowShl + 1>(pk_mine); pkw[2] = row_dpp<kDppRowShl + 2>(pk_mine);
        pkw[3] = row_dpp<kDppRowShl + 3>(pk_mine); pkw[4] = row_dpp<kDppRowShl + 4>(pk_mine);
        pkw[5] = row_dpp<kDppRowShl + 5>(pk_mine); pkw[6] = row_dpp<kDppRowShl + 6>(pk_mine);
        pkw[7] = row_dpp<kDppRowShl + 7>(pk_mine); pkw[8] = row_dpp<kDppRowShl + 8>(pk_mine);
        static_assert(kHeadWin == 9, "window of nine bins");
#pragma unroll
        for (int i = 0; i < kHeadWin; ++i) {
            const float pk = pkw[i];
            int k = lo + i;
            k = k < 0 ? 0 : (k > kHeadBins - 1 ? kHeadBins - 1 : k);
            num = num + (float)k * pk;
            den = den + pk;
        }
        const int px = x0 + pl;
        if (t == 0 && px < a.W) {
            const int p = y * a.W + px;
            const float nd = (num / den) / (float)(kHeadBins - 1);
            if (a.out.nd0) a.out.nd0[b * a.out.nd_sb0 + p] = nd;
            if (a.out.nd1) a.out.nd1[b * a.out.nd_sb1 + p] = nd;
            if (a.out.best) a.out.best[(size_t)b * a.out.P + p] = bi;
        }
    }
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_head_regress(const float* x, int64_t x_sb, int32_t B, int32_t P, const float* w1_packed,
                                    const float* w2_packed, const float* bias2, float* nd_out0, int64_t nd_sb0,
                                    float* nd_out1, int64_t nd_sb1, int64_t* best, void* stream) {
    ITERMVS_RETURN_IF(!x || !w1_packed || !w2_packed || !bias2, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || P < 1, ITERMVS_ERR_DIMS);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(head_regress_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLds) == hipSuccess;
    ITERMVS_RETURN_IF(!attr_ok, ITERMVS_ERR_LAUNCH);
    HeadArgs a;
    a.x = x; a.x_sb = x_sb; a.w1p = w1_packed; a.w2p = w2_packed; a.bias2 = bias2;
    a.nd0 = nd_out0; a.nd1 = nd_out1; a.nd_sb0 = nd_sb0; a.nd_sb1 = nd_sb1; a.best = best; a.P = P;
    hipLaunchKernelGGL(head_regress_kernel, dim3((P + 63) / 64, B), dim3(256), kHeadLds, (hipStream_t)stream, a);
    return itermvs_launch_status();
}

template <bool CONF, bool W2B, bool C3 = false>
static int launch_head_coop(const FusedArgs& a, int grid, int tiles, hipStream_t stream) {
    // CONF: 80 KB, C3: 71 KB -- two workgroups per CU still fit the 160 KB
    constexpr int lds = (kCoLds + (CONF ? kCoWc + 36 : 0) + (C3 ? 2 * (kC3T - kCoT) : 0)) * 4 + (C3 ? kC3WlB : 0);
    auto kern = head_coop_kernel<CONF, W2B, C3>;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    if (!attr_ok) return ITERMVS_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a, tiles);
    return itermvs_launch_status();
}

static int launch_head_fused(const float* hidden, int64_t hidden_sb, int32_t B, int32_t H, int32_t W, const void* w0_tile, int32_t w0_format,
                             const float* w1_packed, const void* w2_packed, int32_t w2_format, const float* bias2, float* nd_out0, int64_t nd_sb0,
                             float* nd_out1, int64_t nd_sb1, int64_t* best, const float* wc_tile, const float* conf_dot, float* conf,
                             int64_t conf_sb, void* stream) {
    ITERMVS_RETURN_IF(!hidden || !w0_tile || !w1_packed || !w2_packed || !bias2, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || H < 1 || W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(w2_format != 0 && w2_format != 3, ITERMVS_ERR_LAYOUT);
    // the 3x3 layer in bf16x3 (w0_format 3): with the bf16x3 last layer, without the confidence rider
    ITERMVS_RETURN_IF(w0_format != 0 && !(w0_format == 3 && w2_format == 3 && !conf), ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF((((uintptr_t)w0_tile) & 15) != 0, ITERMVS_ERR_ALIGN);
    ITERMVS_RETURN_IF((((uintptr_t)w2_packed) & 15) != 0, ITERMVS_ERR_ALIGN);
    FusedArgs a;
    a.hidden = hidden; a.h_sb = hidden_sb; a.w0t = w0_tile; a.w1p = w1_packed; a.w2p = w2_packed; a.bias2 = bias2;
    a.out.nd0 = nd_out0; a.out.nd1 = nd_out1; a.out.nd_sb0 = nd_sb0; a.out.nd_sb1 = nd_sb1; a.out.best = best; a.out.P = H * W;
    a.H = H; a.W = W; a.tiles_x = (W + 15) / 16;
    a.wct = wc_tile; a.cdot = conf_dot; a.conf = conf; a.conf_sb = conf_sb;
    // one tile shared by the four waves of a persistent workgroup, two workgroups per CU
    const int cus = itermvs_num_cus();
    const int tiles = a.tiles_x * H * B;
    static const int wgs_per_cu = [] { const char* e = itermvs_tuning_env("ITERMVS_HEAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2; }();
    const int grid = tiles < wgs_per_cu * cus ? tiles : wgs_per_cu * cus;
    hipStream_t st = (hipStream_t)stream;
    if (conf) return w2_format ? launch_head_coop<true, true>(a, grid, tiles, st) : launch_head_coop<true, false>(a, grid, tiles, st);
    if (w0_format == 3) return launch_head_coop<false, true, true>(a, grid, tiles, st);
    return w2_format ? launch_head_coop<false, true>(a, grid, tiles, st) : launch_head_coop<false, false>(a, grid, tiles, st);
}

extern "C" int itermvs_head_fused(const float* hidden, int64_t hidden_sb, int32_t B, int32_t H, int32_t W,
                                  const void* w0_tile, int32_t w0_format, const float* w1_packed, const void* w2_packed, int32_t w2_format,
                                  const float* bias2, float* nd_out0, int64_t nd_sb0, float* nd_out1, int64_t nd_sb1, int64_t* best,
                                  void* stream) {
    return launch_head_fused(hidden, hidden_sb, B, H, W, w0_tile, w0_format, w1_packed, w2_packed, w2_format, bias2, nd_out0, nd_sb0, nd_out1, nd_sb1, best,
                             nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int itermvs_head_fused_conf(const float* hidden, int64_t hidden_sb, int32_t B, int32_t H, int32_t W,
                                       const float* w0_tile, const float* w1_packed, const void* w2_packed, int32_t w2_format, const float* bias2,
                                       float* nd_out0, int64_t nd_sb0, float* nd_out1, int64_t nd_sb1, int64_t* best,
                                       const float* wc_tile, const float* conf_dot, float* conf, int64_t conf_sb, void* stream) {
    ITERMVS_RETURN_IF(!wc_tile || !conf_dot || !conf, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF((((uintptr_t)wc_tile) & 15) != 0, ITERMVS_ERR_ALIGN);
    return launch_head_fused(hidden, hidden_sb, B, H, W, w0_tile, 0, w1_packed, w2_packed, w2_format, bias2, nd_out0, nd_sb0, nd_out1, nd_sb1, best,
                             wc_tile, conf_dot, conf, conf_sb, stream);
}
