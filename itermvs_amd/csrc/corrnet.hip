// itermvs_corrnet: the whole CorrNet (models/itermvs.py:352-381) in ONE launch.
//
//   c0 = relu(conv3x3(x, 8 -> 8))                    c1 = relu(conv3x3 s2 (c0, 8 -> 16))       c2 = relu(conv3x3 s2 (c1, 16 -> 32))
//   u1 = c1 + deconv3x3 s2 (c2, 32 -> 16)            u0 = c0 + deconv3x3 s2 (u1, 16 -> 8)      y  = conv3x3(u0, 8 -> 1) + bias
//
// As six launches (itermvs_conv2d per layer) one CorrNet costs ~55 us at cfg 1 for ~4 us of arithmetic: five launch
// boundaries and five ramp-ups (weight staging, first fetch, drain) on maps of 20k pixels.  Here a workgroup owns a 32 x 32
// output tile of one map and walks the U-Net with every intermediate in LDS; the halo each layer needs is recomputed
// (x 45x45 -> c0 43x43 -> c1 21x21 -> c2 10x10 -> u1 18x18 -> u0 34x34 -> y 32x32), positions outside the image are stored
// as zeros so every layer sees the zero padding the reference's layer sees.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32) as implicit GEMMs LDS -> LDS: a wave takes groups of 16 consecutive
// output positions (row-major over the layer's region), A = the layer's weights staged in LDS in operand order
// [tap][k-step][q][co], B = the activations read from the channel-planar LDS tile at the tap's displacement, D = 16 output
// channels x 16 positions.  The transposed convolutions run as four parity classes on the INPUT grid (1, 2, 2 and 4 taps).
// (A first version on packed vector FMAs with the weights fed through the scalar cache lost to the six-launch form,
// 68 vs 55 us: a vector FMA needs a fresh weight operand per 128 multiply-adds and neither the scalar cache nor LDS
// broadcasts deliver that; the matrix core reuses each operand register 16 times.)
#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int kCnTile = 32;
#ifndef ITERMVS_CORRNET_THREADS
#define ITERMVS_CORRNET_THREADS 1024
#endif
constexpr int kCnThreads = ITERMVS_CORRNET_THREADS;
constexpr int kCnWaves = kCnThreads / 64;
constexpr int XS = 45, XP = 46;      // region size / LDS row pitch of x
constexpr int C0S = 43, C0P = 44;    // c0 (later u0 in place, region rows / columns 6 .. 39)
constexpr int C1S = 21, C1P = 22;    // c1 (later u1 in place, 2 .. 19), half resolution
constexpr int C2S = 10, C2P = 11;    // c2, quarter resolution
// Channel-plane strides, padded against LDS bank conflicts of the B-operand reads (lane (q, l16) reads channel 4k + q at
// position l16 * stride): a buffer read at stride 1 wants its planes 16 banks apart (q = 0 / 1 fill the 32 banks of a
// ds_read_b32 lane group), a buffer read at stride 2 an ODD distance (q = 0 on the even banks, q = 1 on the odd ones).
// Measured before the padding: 39 % of the kernel's LDS cycles were conflict cycles (profiles/r02_pmc_kernels.json).
constexpr int XPL = XS * XP + 26;    // 2096 = 16 (mod 32): read at stride 1 by conv0
constexpr int C0PL = C0S * C0P + 13; // 1905 = 17 (mod 32): read at stride 2 by conv1 (and at stride 1 by conv5)
constexpr int C1PL = C1S * C1P + 3;  //  465 = 17 (mod 32): read at stride 2 by conv2, at stride 1 by conv4
constexpr int C2PL = C2S * C2P + 2;  //  112 = 16 (mod 32): read at stride 1 by conv3
static_assert(XPL % 32 == 16 && C0PL % 32 == 17 && C1PL % 32 == 17 && C2PL % 32 == 16, "plane strides");
constexpr int kSzX = 8 * XPL, kSzC0 = 8 * C0PL, kSzC1 = 16 * C1PL, kSzC2 = 32 * C2PL;
constexpr int kOffC0 = kSzX, kOffC1 = 0, kOffC2 = kSzC1;      // c1 / c2 reuse the x region once c0 is complete
static_assert(kSzC1 + kSzC2 <= kSzX, "c1 + c2 must fit the x region");
constexpr int kOffW = kSzX + kSzC0;                           // the current layer's weights
constexpr int kSzW = 6912;                                    // (bf16x3 form: the split weights of conv2 / conv3, 27 648 B)
constexpr int kCnLdsFloats = kOffW + kSzW;
static_assert(kCnLdsFloats * 4 <= 160 * 1024, "LDS budget");
// packed weight set of one CorrNet (floats), every layer in MFMA operand order [tap][k-step][q][co padded to 16 / 32]:
// see ops.pack_corrnet_weights
constexpr int kW0 = 0;                         // conv0, two output rows per MFMA (conv0_layer): 12 x 2 x 4 x 16
constexpr int kW1 = kW0 + 12 * 2 * 4 * 16;     // conv1: 9 x 2 x 4 x 16
constexpr int kW2 = kW1 + 9 * 2 * 4 * 16;      // conv2: 9 x 4 x 4 x 32
constexpr int kW3 = kW2 + 9 * 4 * 4 * 32;      // conv3 (transposed): 9 x 8 x 4 x 16
constexpr int kW4 = kW3 + 9 * 8 * 4 * 16;      // conv4 (transposed): 9 x 4 x 4 x 16
constexpr int kW5 = kW4 + 9 * 4 * 4 * 16;      // conv5: [ci 8][tap 9], then the bias
constexpr int kWBias = kW5 + 72;
// bf16x3 form (corrnet_kernel<true>): conv0's weights as bf16 A operands of v_mfma_f32_16x16x32_bf16 -- [18 MFMAs][64 lanes][8 bf16]
// = 4608 floats -- in front of the same layers 1..5 (ops.pack_corrnet_weights(split3=True))
constexpr int kW0Split = 18 * 64 * 4;
static_assert(kW0Split <= kSzW, "conv0's split weights must fit the weight buffer");
constexpr int kXPlaneB = XS * XP * 16;                     // bytes of one bf16 plane of the x tile: [row 45][col 46][8 channels]
static_assert(2 * kXPlaneB <= kSzX * 4 && kXPlaneB <= kSzC0 * 4, "planes h, m of x must fit the x region, plane l the c0 region");
// bf16x3 form, layers 2..4 (conv2, the two transposed convolutions): the split weights of itermvs_conv2d's weight_format 3,
// bf16 [tap 9][chunk = ci / 16][term h, m, l][row co, padded to 16 / 32][16 ci] -- conv2 [9][1][3][32][16], conv3 [9][2][3][16][16]
// (27 648 B each), conv4 [9][1][3][16][16] (13 824 B); conv1 stays on the fp32 instruction (its input c0 stays fp32 for the skip
// and the last layer).  Offsets in floats:
constexpr int kS3W1 = kW0Split;                // conv1, fp32 operand order as above
constexpr int kS3W2 = kS3W1 + (kW2 - kW1);     // conv2 split
constexpr int kS3W3 = kS3W2 + 6912;            // conv3 split
constexpr int kS3W4 = kS3W3 + 6912;            // conv4 split
constexpr int kS3W5 = kS3W4 + 3456;            // conv5 [ci 8][tap 9]
constexpr int kS3Bias = kS3W5 + 72;
constexpr int kS3WeightFloats = kS3Bias + 8;
static_assert(kS3WeightFloats == 23120, "packed set of the bf16x3 form (include/itermvs_hip.h)");
// c1 and c2 as bf16 triples (the B operands of conv2 and of the transposed convolutions) in the x region, which is free once conv0 has
// read it: [chunk = channel / 16][term h, m, l][half = 8 channels][position][16 B], positions padded to a multiple of 16 (256 B)
constexpr int kC1PP = (C1S * C1P + 15) / 16 * 16, kC1HalfB = kC1PP * 16, kC1PlaneB = 2 * kC1HalfB;          // 464 positions
constexpr int kC2PP = (C2S * C2P + 15) / 16 * 16, kC2HalfB = kC2PP * 16, kC2PlaneB = 2 * kC2HalfB, kC2ChunkB = 3 * kC2PlaneB;
constexpr int kOffC1sB = 0, kOffC2sB = 3 * kC1PlaneB;          // byte offsets inside the x region
static_assert(kOffC2sB + 2 * kC2ChunkB <= kSzX * 4, "c1 + c2 as bf16 triples must fit the x region");

struct CorrNetArgs {
    const float* x;
    const float* w[3];
    int seg_end[3];
    float* out;
    float* out2;
    int64_t x_sn, out_sn, out2_sn;
    int M, H, W, tiles_x;
    int vec4;          // x base and map stride 16-byte aligned: the tile is staged with float4 loads
};

// the next layer's weights: global -> registers now, registers -> LDS once the current layer is done with the buffer
template <int N>
struct WeightStage {
    static constexpr int PER = (N + kCnThreads - 1) / kCnThreads;
    float r[PER];
    __device__ __forceinline__ void fetch(const float* __restrict__ src, int tid) {
#pragma unroll
        for (int i = 0; i < PER; ++i) r[i] = tid + i * kCnThreads < N ? src[tid + i * kCnThreads] : 0.0f;
    }
    __device__ __forceinline__ void commit(float* __restrict__ dst, int tid) const {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (tid + i * kCnThreads < N) dst[tid + i * kCnThreads] = r[i];
    }
};

// 3x3 convolution LDS -> LDS on the matrix cores.  In: CIN planes of stride INPL, row pitch INP, whose (0,0) is tap (0,0) of
// output (0,0) (stride S); Out: COUT planes of stride OUTPL, OUTS x OUTS positions at row pitch OUTP; Wl: [9][CIN/4][4][MB*16].  Output position (oy, ox) has image coordinates
// (gy0 + oy, gx0 + ox) at this layer's resolution; outside [0,imgH) x [0,imgW) a zero is stored.
// four channels (4 qq .. 4 qq + 3 of a 16-channel chunk) of one position -> the three 8-byte entries of a split map chunk
__device__ __forceinline__ void split_put4(char* __restrict__ chunk, int half_b, int qq, int pos, float v0, float v1, float v2, float v3) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split_pair(v0, v1, h0, m0, l0);
    split_pair(v2, v3, h1, m1, l1);
    char* __restrict__ d = chunk + (qq >> 1) * half_b + pos * 16 + (qq & 1) * 8;
    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(d + 2 * half_b) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(d + 4 * half_b) = u32x2{l0, l1};
}
// ... and back: h + m + l, exact (8 + 8 + 8 significant bits)
__device__ __forceinline__ void split_get4(const char* __restrict__ chunk, int half_b, int qq, int pos, float (&v)[4]) {
    const char* __restrict__ s = chunk + (qq >> 1) * half_b + pos * 16 + (qq & 1) * 8;
    const u32x2 hh = *reinterpret_cast<const u32x2*>(s);
    const u32x2 mm = *reinterpret_cast<const u32x2*>(s + 2 * half_b);
    const u32x2 ll = *reinterpret_cast<const u32x2*>(s + 4 * half_b);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t hw = r < 2 ? hh[0] : hh[1], mw = r < 2 ? mm[0] : mm[1], lw = r < 2 ? ll[0] : ll[1];
        const float ah = __uint_as_float((r & 1) ? (hw & 0xffff0000u) : (hw << 16));
        const float am = __uint_as_float((r & 1) ? (mw & 0xffff0000u) : (mw << 16));
        const float al = __uint_as_float((r & 1) ? (lw & 0xffff0000u) : (lw << 16));
        v[r] = (ah + am) + al;
    }
}

// SPLIT_HALF_B > 0 (16 output channels): the results go to `OutS` as bf16 triples (position oy * OUTP + ox) instead of fp32 planes
template <int CIN, int COUT, int MB, int S, int INPL, int INP, int OUTS, int OUTPL, int OUTP, int SPLIT_HALF_B = 0>
__device__ __forceinline__ void conv_layer(const float* __restrict__ In, float* __restrict__ Out, const float* __restrict__ Wl,
                                           int gy0, int gx0, int imgH, int imgW, int wave, int lane, char* __restrict__ OutS = nullptr) {
    constexpr int KS = CIN / 4, NPOS = OUTS * OUTS, GROUPS = (NPOS + 15) / 16;
    const int q = lane >> 4, l16 = lane & 15;
    // the A operands (this lane's weight of every (tap, k-step, block)) are the same for every group: read them once
    float aw[9 * KS * MB];
#pragma unroll
    for (int i = 0; i < 9 * KS; ++i)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) aw[i * MB + mb] = Wl[(i * 4 + q) * (MB * 16) + mb * 16 + l16];
    for (int g = wave; g < GROUPS; g += kCnWaves) {
        const int pos = g * 16 + l16;
        const int pc = pos < NPOS ? pos : NPOS - 1;          // surplus lanes recompute the last position, never store
        const int oy = pc / OUTS, ox = pc - oy * OUTS;
        f32x4 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* __restrict__ bp = In + q * INPL + oy * S * INP + ox * S;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float b = bp[ks * 4 * INPL + ky * INP + kx];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(tap * KS + ks) * MB + mb], b, acc[mb], 0, 0, 0);
            }
        }
        const int gy = gy0 + oy, gx = gx0 + ox;
        const bool inside = gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
        if (pos < NPOS) {
            if constexpr (SPLIT_HALF_B > 0) {
                static_assert(MB == 1 && COUT == 16, "split output: one 16-channel chunk");
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inside ? fmaxf(acc[0][r], 0.0f) : 0.0f;
                split_put4(OutS, SPLIT_HALF_B, q, oy * OUTP + ox, v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = mb * 16 + q * 4 + r;
                        if (co < COUT) Out[co * OUTPL + oy * OUTP + ox] = inside ? fmaxf(acc[mb][r], 0.0f) : 0.0f;
                    }
            }
        }
    }
}

// ---- layers on v_mfma_f32_16x16x32_bf16 from split maps (bf16x3 arithmetic of conv_tile3.hip: K = 32 = two 16-channel terms,
//      A1 = [wh | wh], A2 = [wm | wm], A3 = [wl | wh]; B1 = [xh | xm], B3 = [xh | xl]; acc += A3 B3 + A2 B1 + A1 B1) ----
// a lane's B operands of one 16-channel chunk at a position, and its three A operands of (tap, chunk): Wl = bf16 [tap][chunk][3][ROWS][16]
struct SplitLane {
    int half, second, l16, q;
    __device__ __forceinline__ explicit SplitLane(int lane) : half((lane >> 4) & 1), second(lane >> 5), l16(lane & 15), q(lane >> 4) {}
    template <int HALF_B>
    __device__ __forceinline__ void b(const char* __restrict__ chunk, int pos, bf8& b1, bf8& b3) const {
        const char* __restrict__ p = chunk + half * HALF_B + pos * 16;
        b1 = *reinterpret_cast<const bf8*>(p + (second ? 2 * HALF_B : 0));
        b3 = *reinterpret_cast<const bf8*>(p + (second ? 4 * HALF_B : 0));
    }
    template <int ROWS>
    __device__ __forceinline__ void a(const char* __restrict__ wtap, int row0, bf8& a1, bf8& a2, bf8& a3) const {
        const char* __restrict__ p = wtap + (row0 + l16) * 32 + half * 16;
        a1 = *reinterpret_cast<const bf8*>(p);
        a2 = *reinterpret_cast<const bf8*>(p + ROWS * 32);
        a3 = *reinterpret_cast<const bf8*>(p + (second ? 0 : 2 * ROWS * 32));
    }
};

// conv2 (16 -> 32, stride 2, ReLU): c1 triples -> c2 triples.  Work unit = (group of 16 output positions, block of 16 output
// channels): 14 units over the 16 waves, 27 MFMAs of 16 cycles each (the fp32 form: 7 groups x 72 MFMAs of 40 cycles on 7 waves)
__device__ __forceinline__ void conv2_split(const char* __restrict__ In, char* __restrict__ Out, const char* __restrict__ Wl,
                                            int gy0, int gx0, int imgH, int imgW, int wave, int lane) {
    constexpr int NPOS = C2S * C2S, GROUPS = (NPOS + 15) / 16;
    const SplitLane L(lane);
    for (int u = wave; u < GROUPS * 2; u += kCnWaves) {
        const int g = u >> 1, mb = u & 1;
        const int pos = g * 16 + L.l16;
        const int pc = pos < NPOS ? pos : NPOS - 1;
        const int oy = pc / C2S, ox = pc - oy * C2S;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            bf8 a1, a2, a3, b1, b3;
            L.a<32>(Wl + tap * (3 * 32 * 32), mb * 16, a1, a2, a3);
            L.b<kC1HalfB>(In, (2 * oy + ky) * C1P + 2 * ox + kx, b1, b3);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
        }
        const int gy = gy0 + oy, gx = gx0 + ox;
        const bool inside = gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
        if (pos < NPOS)
            split_put4(Out + mb * kC2ChunkB, kC2HalfB, L.q, oy * C2P + ox, inside ? fmaxf(acc[0], 0.0f) : 0.0f, inside ? fmaxf(acc[1], 0.0f) : 0.0f,
                       inside ? fmaxf(acc[2], 0.0f) : 0.0f, inside ? fmaxf(acc[3], 0.0f) : 0.0f);
    }
}

// Transposed convolution + skip from a split map (see deconv_layer for the parity classes): NCH input chunks of 16 channels at
// In (chunk stride IN_CHUNK_B, half stride IN_HALF_B, pitch INP); the skip tensor is updated in place -- as bf16 triples
// (SKIP_HALF_B > 0: u1 over c1, 16 channels) or as fp32 planes (u0 over c0, COUT = 8).  Wl = bf16 [tap][chunk][3][16][16].
template <int NCH, int IN_HALF_B, int IN_CHUNK_B, int COUT, int NB, int IO, int INP, int OO, int OUTPL, int OUTP, int SKIP_HALF_B>
__device__ __forceinline__ void deconv_split(const char* __restrict__ In, float* __restrict__ Skip, char* __restrict__ SkipS,
                                             const char* __restrict__ Wl, int gy0, int gx0, int imgH, int imgW, int wave, int lane) {
    constexpr int NPOS = NB * NB, GROUPS = (NPOS + 15) / 16;
    const SplitLane L(lane);
    for (int g = wave; g < GROUPS; g += kCnWaves) {
        const int pos = g * 16 + L.l16;
        const int pc = pos < NPOS ? pos : NPOS - 1;
        const int ba = pc / NB, bb = pc - ba * NB;
        f32x4 acc[4];                                        // class py * 2 + px
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    bf8 b1, b3;
                    L.b<IN_HALF_B>(In + ch * IN_CHUNK_B, (ba + IO + dy) * INP + bb + IO + dx, b1, b3);
#pragma unroll
                    for (int py = 0; py <= dy; ++py)
#pragma unroll
                        for (int px = 0; px <= dx; ++px) {
                            const int ky = dy == 1 ? py : 2, kx = dx == 1 ? px : 2;
                            bf8 a1, a2, a3;
                            L.a<16>(Wl + ((ky * 3 + kx) * NCH + ch) * (3 * 16 * 32), 0, a1, a2, a3);
                            f32x4& c = acc[py * 2 + px];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c, 0, 0, 0);
                        }
                }
        if (pos < NPOS) {
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int o = 2 * ba + py, p = 2 * bb + px;
                    const int gy = gy0 + o, gx = gx0 + p;
                    const bool inside = gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
                    if constexpr (SKIP_HALF_B > 0) {
                        static_assert(COUT == 16, "split skip: one 16-channel chunk");
                        const int sp = (o + OO) * OUTP + p + OO;
                        float v[4];
                        split_get4(SkipS, SKIP_HALF_B, L.q, sp, v);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = inside ? v[r] + acc[py * 2 + px][r] : 0.0f;
                        split_put4(SkipS, SKIP_HALF_B, L.q, sp, v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = L.q * 4 + r;
                            if (co < COUT) {
                                float* __restrict__ d = Skip + co * OUTPL + (o + OO) * OUTP + p + OO;
                                *d = inside ? *d + acc[py * 2 + px][r] : 0.0f;
                            }
                        }
                    }
                }
        }
    }
}

// conv0 (8 -> 8 channels) would fill only half of the 16 MFMA rows.  Here one MFMA tile holds TWO output rows of 16
// positions: rows m = 0..7 are the 8 channels of output row 2p, rows 8..15 those of output row 2p + 1, contracted over the
// 4-row x 3-column window both need (K = 4 x 3 x 8 = 96 = 24 steps; the row a half does not use has zero weights) --
// 24 MFMAs per 32 positions instead of 36.  Wl: [window row 4][kx 3][k-step 2][q 4][16] (ops.pack_corrnet_weights).
template <int INPL, int INR, int INP, int OUTS, int OUTPL, int OUTP>
__device__ __forceinline__ void conv0_layer(const float* __restrict__ In, float* __restrict__ Out, const float* __restrict__ Wl,
                                            int gy0, int gx0, int imgH, int imgW, int wave, int lane) {
    constexpr int PAIRS = (OUTS + 1) / 2, NPOS = PAIRS * OUTS, GROUPS = (NPOS + 15) / 16;
    const int q = lane >> 4, l16 = lane & 15;
    float aw[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) aw[i] = Wl[(i * 4 + q) * 16 + l16];
    for (int g = wave; g < GROUPS; g += kCnWaves) {
        const int pos = g * 16 + l16;
        const int pc = pos < NPOS ? pos : NPOS - 1;
        const int rp = pc / OUTS, ox = pc - rp * OUTS;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* __restrict__ bp = In + q * INPL + 2 * rp * INP + ox;
        // window row 3 of the last pair lies past the region (only the unused second output row would read it): re-read the
        // last row instead of whatever follows it in LDS -- its weights for the first output row are zero, 0 * NaN is not
        const int w3 = (min(2 * rp + 3, INR - 1) - 2 * rp) * INP;
#pragma unroll
        for (int wr = 0; wr < 4; ++wr)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(wr * 3 + kx) * 2 + ks],
                                                               bp[ks * 4 * INPL + (wr < 3 ? wr * INP : w3) + kx], acc, 0, 0, 0);
        const int oy = 2 * rp + (q >> 1);
        const int gy = gy0 + oy, gx = gx0 + ox;
        const bool inside = gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
        if (pos < NPOS && oy < OUTS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Out[((q & 1) * 4 + r) * OUTPL + oy * OUTP + ox] = inside ? fmaxf(acc[r], 0.0f) : 0.0f;
        }
    }
}

// ---- conv0 on the bf16 matrix instruction (corrnet_kernel<true>) ----
// x has 8 channels: one 16-byte LDS entry per position and term (h, m, l of the exact split, common.hpp).  The K = 32 of
// v_mfma_f32_16x16x32_bf16 carries TWO positions of the 4 x 3 window (the two-rows-per-tile form of conv0_layer: 12 window
// positions, rows 0..7 of the tile = output row 2p, rows 8..15 = output row 2p + 1) times TWO terms:
//     lanes q = 0, 1: window positions 2v, 2v + 1, first term;   q = 2, 3: the same positions, second term
//     B1 = [xh xh | xm xm]   B3 = [xh xh | xl xl]        A1 = wh   A2 = wm   A3 = [wl wl | wh wh]
//     acc += A1 B1 (wh xh + wh xm) + A2 B1 (wm xh + wm xm) + A3 B3 (wl xh + wh xl)          -- the six largest of nine products
// 18 MFMAs of 16 cycles per 32 output positions instead of 24 of 40.  Planes h and m of x (2 x 33 KB) fill the x region; the l
// plane lies in the c0 region, which is empty until conv0's results are stored: the accumulators of a wave's groups stay in
// registers until every wave is done reading (one barrier), then c0 overwrites the l plane.
template <int INR, int INP, int OUTS, int OUTPL, int OUTP>
struct Conv0Split {
    static constexpr int PAIRS = (OUTS + 1) / 2, NPOS = PAIRS * OUTS, GROUPS = (NPOS + 15) / 16;
    static constexpr int PER = (GROUPS + kCnWaves - 1) / kCnWaves;
    f32x4 acc[PER];
    // the B-operand byte offset of window-position pair v for this lane at group position (rp, ox)
    static __device__ __forceinline__ int boff(int v, int sel, int rp, int ox) {
        const int wp = 2 * v + sel, wr = wp / 3, kx = wp - wr * 3;
        const int row = 2 * rp + wr < INR ? 2 * rp + wr : INR - 1;       // window row 3 of the last pair: re-read the last row (zero weights)
        return (row * INP + ox + kx) * 16;
    }
    // Xb: planes h, m (kXPlaneB apart); Lb: plane l; Wb: the 18 A operands
    __device__ __forceinline__ void run(const char* __restrict__ Xb, const char* __restrict__ Lb, const char* __restrict__ Wb, int wave, int lane) {
        const int q = lane >> 4, l16 = lane & 15;
        const int sel = q & 1, second = q >> 1;
        const char* __restrict__ x1 = Xb + second * kXPlaneB;            // B1: h (q = 0, 1) / m (q = 2, 3)
        const char* __restrict__ x3 = second ? Lb : Xb;                  // B3: h / l
        bf8 aw[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) aw[i] = *reinterpret_cast<const bf8*>(Wb + (i * 64 + lane) * 16);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int g = wave + k * kCnWaves;
            f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
            if (g < GROUPS) {                    // wave-uniform
                const int pos = g * 16 + l16;
                const int pc = pos < NPOS ? pos : NPOS - 1;
                const int rp = pc / OUTS, ox = pc - rp * OUTS;
#pragma unroll
                for (int v = 0; v < 6; ++v) {
                    const int o = boff(v, sel, rp, ox);
                    const bf8 b1 = *reinterpret_cast<const bf8*>(x1 + o);
                    const bf8 b3 = *reinterpret_cast<const bf8*>(x3 + o);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[12 + v], b3, c, 0, 0, 0);         // A3 B3 (smaller terms first)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[2 * v + 1], b1, c, 0, 0, 0);      // A2 B1
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[2 * v], b1, c, 0, 0, 0);          // A1 B1
                }
            }
            acc[k] = c;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ Out, int gy0, int gx0, int imgH, int imgW, int wave, int lane) const {
        const int q = lane >> 4, l16 = lane & 15;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int g = wave + k * kCnWaves;
            const int pos = g * 16 + l16;
            if (g < GROUPS && pos < NPOS) {
                const int rp = pos / OUTS, ox = pos - rp * OUTS;
                const int oy = 2 * rp + (q >> 1);
                const int gy = gy0 + oy, gx = gx0 + ox;
                const bool inside = gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
                if (oy < OUTS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Out[((q & 1) * 4 + r) * OUTPL + oy * OUTP + ox] = inside ? fmaxf(acc[k][r], 0.0f) : 0.0f;
                }
            }
        }
    }
};

// 8 channels of one position of the x tile -> its three 16-byte bf16 entries (planes h, m at Xb, plane l at Lb)
__device__ __forceinline__ void x_split_put(char* __restrict__ Xb, char* __restrict__ Lb, int p, const float (&v)[8]) {
    u32x4 Hh, Mm, Ll;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t h, m, l;
        split_pair(v[2 * k], v[2 * k + 1], h, m, l);
        Hh[k] = h; Mm[k] = m; Ll[k] = l;
    }
    *reinterpret_cast<u32x4*>(Xb + p * 16) = Hh;
    *reinterpret_cast<u32x4*>(Xb + kXPlaneB + p * 16) = Mm;
    *reinterpret_cast<u32x4*>(Lb + p * 16) = Ll;
}

// ConvTranspose2d(3, stride 2, pad 1, out_pad 1) + skip, LDS -> LDS in place: out[o] = sum_i in[i] w[o - 2i + 1].
// The output region's origin is an ODD coordinate, so output (2a + py, 2b + px) of the region takes
//   py = 0: rows a+1 (ky 0) and a (ky 2);  py = 1: row a+1 (ky 1)           -- same for columns --
// i.e. parity class (py, px) is a small convolution on the input grid with (2 - py)(2 - px) taps.  NB x NB blocks; input
// block (a, b) sits at In[..][a + IO][b + IO], output (o, p) at Skip[..][o + OO][p + OO] (holding the skip tensor, updated
// in place).  Wl: [9][CIN/4][4][16].
template <int CIN, int COUT, int NB, int IO, int INPL, int INP, int OO, int OUTPL, int OUTP>
__device__ __forceinline__ void deconv_layer(const float* __restrict__ In, float* __restrict__ Skip, const float* __restrict__ Wl,
                                             int gy0, int gx0, int imgH, int imgW, int wave, int lane) {
    constexpr int KS = CIN / 4, NPOS = NB * NB, GROUPS = (NPOS + 15) / 16;
    const int q = lane >> 4, l16 = lane & 15;
    // unit = 16 input blocks with all four output parities: every input value is read once and feeds the 1 / 2 / 2 / 4
    // parity classes that use it (9 MFMAs per k-step from 4 LDS reads of B)
    for (int g = wave; g < GROUPS; g += kCnWaves) {
        const int pos = g * 16 + l16;
        const int pc = pos < NPOS ? pos : NPOS - 1;
        const int ba = pc / NB, bb = pc - ba * NB;
        f32x4 acc[4];                                        // class py * 2 + px
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* __restrict__ bp = In + q * INPL + (ba + IO) * INP + bb + IO;
        const float* __restrict__ ap = Wl + q * 16 + l16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float b = bp[ks * 4 * INPL + dy * INP + dx];
                    // input (a + dy, b + dx) reaches output row parity py through ky: dy = 1 -> ky = py (0 or 1), dy = 0 -> ky = 2 (py = 0 only)
#pragma unroll
                    for (int py = 0; py <= dy; ++py)
#pragma unroll
                        for (int px = 0; px <= dx; ++px) {
                            const int ky = dy == 1 ? py : 2, kx = dx == 1 ? px : 2;
                            acc[py * 2 + px] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[((ky * 3 + kx) * KS + ks) * 64], b, acc[py * 2 + px], 0, 0, 0);
                        }
                }
        if (pos < NPOS) {
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int o = 2 * ba + py, p = 2 * bb + px;
                    const int gy = gy0 + o, gx = gx0 + p;
                    const bool inside = gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = q * 4 + r;
                        if (co < COUT) {
                            float* __restrict__ d = Skip + co * OUTPL + (o + OO) * OUTP + p + OO;
                            *d = inside ? *d + acc[py * 2 + px][r] : 0.0f;
                        }
                    }
                }
        }
    }
}

template <bool S3>
__global__ void __launch_bounds__(kCnThreads) corrnet_kernel(const CorrNetArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* __restrict__ X = lds;
    float* __restrict__ C0 = lds + kOffC0;
    float* __restrict__ C1 = lds + kOffC1;
    float* __restrict__ C2 = lds + kOffC2;
    float* __restrict__ WL = lds + kOffW;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int m = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int X0 = tx * kCnTile, Y0 = ty * kCnTile;
    const float* __restrict__ wt = a.w[m < a.seg_end[0] ? 0 : (m < a.seg_end[1] ? 1 : 2)];
    const int H = a.H, W = a.W, H2 = H >> 1, W2 = W >> 1, H4 = H >> 2, W4 = W >> 2;

    if constexpr (S3) {
        // ---- x tile -> two bf16 planes (h, m) + conv0's split weights -> LDS; conv0 in two passes (see Conv0Split) ----
        const float* __restrict__ xm = a.x + (int64_t)m * a.x_sn;
        char* __restrict__ Xb = reinterpret_cast<char*>(X);
        WeightStage<kW0Split> w0s;
        w0s.fetch(wt, tid);
        char* __restrict__ Lb = reinterpret_cast<char*>(C0);
        // item = one position, 8 dword loads (lanes = consecutive columns of one plane row); 45 x 46 positions (the pad column as
        // zeros), up to 3 per thread.  (A form with aligned 16-byte loads -- 540 threads x 8 planes x float4, four positions
        // split per thread -- measured 25.2 us per launch against 23.9 us: the split arithmetic of a position is the
        // long pole of this phase and spreads better over all 1024 threads.)
        // (buffer loads: one vector offset per position -- out of range where the zero padding is --, the channel plane in the scalar
        //  offset: the phase is bound by its vector instructions, 16 waves per CU computing 64-bit addresses and predicates)
        float v[3][8];
        const uint32_t plane_b = (uint32_t)(H * W) * 4u;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xm, 0, (int)(8u * plane_b), 0x00020000);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = tid + k * kCnThreads;
            const int ry = p / XP, rx = p - ry * XP;
            const int gy = Y0 - 8 + ry, gx = X0 - 8 + rx;
            const bool ok = p < XS * XP && rx < XS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const uint32_t vo = ok ? (uint32_t)(gy * W + gx) * 4u : 0x7fffffffu;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) v[k][ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, (uint32_t)ci * plane_b, 0));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + k * kCnThreads < XS * XP) x_split_put(Xb, Lb, tid + k * kCnThreads, v[k]);
        w0s.commit(WL, tid);
        __syncthreads();
        Conv0Split<XS, XP, C0S, C0PL, C0P> c0;
        WeightStage<kW2 - kW1> nw;
        nw.fetch(wt + kS3W1, tid);
        c0.run(Xb, Lb, reinterpret_cast<const char*>(WL), wave, lane);
        __syncthreads();                                   // every wave is done with the l plane (c0 region) and conv0's weights
        c0.store(C0, Y0 - 7, X0 - 7, H, W, wave, lane);
        nw.commit(WL, tid);
        __syncthreads();
    } else {
        // ---- x tile (+8 / +12 halo) and conv0's weights -> LDS; zeros outside the image and in the pad column ----
        {
            const float* __restrict__ xm = a.x + (int64_t)m * a.x_sn;
            // all of a wave's row loads (and conv0's weights) are issued before the first LDS write: one HBM round trip instead
            // of one per batch
            WeightStage<kW1 - kW0> w0s;
            w0s.fetch(wt + kW0, tid);
            if (a.vec4) {
                // 16-byte loads: the region starts at column X0 - 8 (a multiple of 4, like W), so a row is 12 aligned float4 that
                // lie entirely inside or outside the image; item = (channel, row, float4) -- 4 320 items, 5 per thread, instead of
                // 23 dword row loads per wave (dword loads of unaligned tile rows stream at ~2.5 TB/s, aligned 16-byte ones at
                // ~7.6: tools/ubench/tile_read.hip)
                constexpr int ITEMS = 8 * XS * 12, PER4 = (ITEMS + kCnThreads - 1) / kCnThreads;
                f32x4 v4[PER4];
    #pragma unroll
                for (int i = 0; i < PER4; ++i) {
                    const int id = min(tid + i * kCnThreads, ITEMS - 1);
                    const int r = id / 12, j = id - r * 12;
                    const int ci = r / XS, ry = r - ci * XS;
                    const int gy = Y0 - 8 + ry, gx = X0 - 8 + 4 * j;
                    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    v4[i] = ok ? *reinterpret_cast<const f32x4*>(xm + (int64_t)ci * H * W + gy * W + gx) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
    #pragma unroll
                for (int i = 0; i < PER4; ++i) {
                    const int id = tid + i * kCnThreads;
                    if (id < ITEMS) {
                        const int r = id / 12, j = id - r * 12;
                        const int ci = r / XS, ry = r - ci * XS;
                        float* __restrict__ d = X + ci * XPL + ry * XP + 4 * j;       // even offset: two 8-byte stores
                        if (j < 11) {
                            *reinterpret_cast<float2*>(d) = float2{v4[i][0], v4[i][1]};
                            *reinterpret_cast<float2*>(d + 2) = float2{v4[i][2], v4[i][3]};
                        } else {
                            *reinterpret_cast<float2*>(d) = float2{v4[i][0], 0.0f};    // column 44 and the pad column; 46, 47 are the next row's
                        }
                    }
                }
            } else {
                // one (channel, row) of the region per wave iteration, lanes = columns (any alignment of x)
                const int gx = X0 - 8 + lane;
                const bool okx = lane < XS && gx >= 0 && gx < W;
                constexpr int PER = (8 * XS + kCnWaves - 1) / kCnWaves;
                float v[PER];
    #pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int r = min(wave + i * kCnWaves, 8 * XS - 1);
                    const int ci = r / XS, ry = r - ci * XS;             // wave-uniform
                    const int gy = Y0 - 8 + ry;
                    const bool ok = okx && gy >= 0 && gy < H;
                    v[i] = ok ? xm[(int64_t)ci * H * W + gy * W + gx] : 0.0f;
                }
    #pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int r = wave + i * kCnWaves;
                    if (r < 8 * XS && lane < XP) {
                        const int ci = r / XS, ry = r - ci * XS;
                        X[ci * XPL + ry * XP + lane] = v[i];
                    }
                }
            }
            w0s.commit(WL, tid);
        }
        __syncthreads();
        {   // c0 = relu(conv(x)): 43 x 43
            WeightStage<kW2 - kW1> nw;
            nw.fetch(wt + kW1, tid);
            conv0_layer<XPL, XS, XP, C0S, C0PL, C0P>(X, C0, WL, Y0 - 7, X0 - 7, H, W, wave, lane);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
    }
    if constexpr (S3) {
        // ---- conv2 and the two transposed convolutions on the bf16 matrix instruction (the four fp32 layers were 31.5 k of the
        //      launch's 52 k cycles, profiles/r06/r06k_*): c1 and c2 live as bf16 triples in the x region, c0 / u0 stay fp32 ----
        char* __restrict__ C1s = reinterpret_cast<char*>(X) + kOffC1sB;
        char* __restrict__ C2s = reinterpret_cast<char*>(X) + kOffC2sB;
        const char* __restrict__ WLb = reinterpret_cast<const char*>(WL);
        {   // c1 = relu(conv s2 (c0)): 21 x 21 at half resolution, fp32 instruction (c0 is fp32), stored as triples
            WeightStage<6912> nw;
            nw.fetch(wt + kS3W2, tid);
            conv_layer<8, 16, 1, 2, C0PL, C0P, C1S, C1PL, C1P, kC1HalfB>(C0, nullptr, WL, (Y0 >> 1) - 3, (X0 >> 1) - 3, H2, W2, wave, lane, C1s);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
        {   // c2 = relu(conv s2 (c1)): 10 x 10 at quarter resolution
            WeightStage<6912> nw;
            nw.fetch(wt + kS3W3, tid);
            conv2_split(C1s, C2s, WLb, (Y0 >> 2) - 1, (X0 >> 2) - 1, H4, W4, wave, lane);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
        {   // u1 = c1 + deconv(c2): 18 x 18 = c1 rows / columns 2 .. 19, in place (triples)
            WeightStage<3456> nw;
            nw.fetch(wt + kS3W4, tid);
            deconv_split<2, kC2HalfB, kC2ChunkB, 16, 9, 0, C2P, 2, 0, C1P, kC1HalfB>(C2s, nullptr, C1s, WLb, (Y0 >> 1) - 1, (X0 >> 1) - 1, H2, W2, wave, lane);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
        // u0 = c0 + deconv(u1): 34 x 34 = c0 rows / columns 6 .. 39, in place (fp32 planes)
        deconv_split<1, kC1HalfB, 0, 8, 17, 2, C1P, 6, C0PL, C0P, 0>(C1s, C0, nullptr, WLb, Y0 - 1, X0 - 1, H, W, wave, lane);
        __syncthreads();
    } else {
        {   // c1 = relu(conv s2 (c0)): 21 x 21 at half resolution (over the x region)
            WeightStage<kW3 - kW2> nw;
            nw.fetch(wt + kW2, tid);
            conv_layer<8, 16, 1, 2, C0PL, C0P, C1S, C1PL, C1P>(C0, C1, WL, (Y0 >> 1) - 3, (X0 >> 1) - 3, H2, W2, wave, lane);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
        {   // c2 = relu(conv s2 (c1)): 10 x 10 at quarter resolution
            WeightStage<kW4 - kW3> nw;
            nw.fetch(wt + kW3, tid);
            conv_layer<16, 32, 2, 2, C1PL, C1P, C2S, C2PL, C2P>(C1, C2, WL, (Y0 >> 2) - 1, (X0 >> 2) - 1, H4, W4, wave, lane);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
        {   // u1 = c1 + deconv(c2): 18 x 18 = c1 rows / columns 2 .. 19, in place
            WeightStage<kW5 - kW4> nw;
            nw.fetch(wt + kW4, tid);
            deconv_layer<32, 16, 9, 0, C2PL, C2P, 2, C1PL, C1P>(C2, C1, WL, (Y0 >> 1) - 1, (X0 >> 1) - 1, H2, W2, wave, lane);
            __syncthreads();
            nw.commit(WL, tid);
        }
        __syncthreads();
        // u0 = c0 + deconv(u1): 34 x 34 = c0 rows / columns 6 .. 39, in place
        deconv_layer<16, 8, 17, 2, C1PL, C1P, 6, C0PL, C0P>(C1, C0, WL, Y0 - 1, X0 - 1, H, W, wave, lane);
        __syncthreads();

    }

    // ---- y = conv(u0, 8 -> 1) + bias on the vector ALUs: 32 x 32, a thread owns 2 neighbouring pixels; the 72 weights are
    //      wave-uniform (scalar loads) ----
    if (tid < 512) {
        const int oy = tid >> 4, ox = (tid & 15) * 2;
        float y0 = wt[S3 ? kS3Bias : kWBias], y1 = y0;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* __restrict__ row = C0 + ci * C0PL + (oy + ky + 6) * C0P + ox + 6;
                const float in[4] = {row[0], row[1], row[2], row[3]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wt[(S3 ? kS3W5 : kW5) + ci * 9 + ky * 3 + kx];
                    y0 = fmaf(wv, in[kx], y0);
                    y1 = fmaf(wv, in[kx + 1], y1);
                }
            }
        const int gy = Y0 + oy, gx = X0 + ox;
        if (gy < H) {
            float* __restrict__ o = a.out + (int64_t)m * a.out_sn + gy * W + gx;
            if (gx < W) o[0] = y0;
            if (gx + 1 < W) o[1] = y1;
            if (a.out2) {
                float* __restrict__ o2 = a.out2 + (int64_t)m * a.out2_sn + gy * W + gx;
                if (gx < W) o2[0] = y0;
                if (gx + 1 < W) o2[1] = y1;
            }
        }
    }
}

}  // namespace itermvs

using namespace itermvs;

static int launch_corrnet(const float* x, int64_t x_sn, const float* const* weights, const int32_t* seg_end, int32_t n_seg,
                          int32_t M, int32_t H, int32_t W, float* out, int64_t out_sn, float* out2, int64_t out2_sn, bool split3,
                          void* stream) {
    ITERMVS_RETURN_IF(!x || !weights || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(M < 1 || H < 4 || W < 4 || (H & 3) || (W & 3) || n_seg < 1 || n_seg > 3, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((int64_t)8 * H * W * 4 >= ((int64_t)1 << 31), ITERMVS_ERR_DIMS);      // 32-bit buffer offsets within a map
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
    a.vec4 = (((uintptr_t)x) % 16 == 0 && x_sn % 4 == 0) ? 1 : 0;
    a.M = M; a.H = H; a.W = W; a.tiles_x = (W + kCnTile - 1) / kCnTile;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(corrnet_kernel<false>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kCnLdsFloats * 4) == hipSuccess &&
                                hipFuncSetAttribute(reinterpret_cast<const void*>(corrnet_kernel<true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kCnLdsFloats * 4) == hipSuccess;
    ITERMVS_RETURN_IF(!attr_ok, ITERMVS_ERR_LAUNCH);
    const dim3 grid(a.tiles_x * ((H + kCnTile - 1) / kCnTile), M);
    if (split3) hipLaunchKernelGGL(corrnet_kernel<true>, grid, dim3(kCnThreads), kCnLdsFloats * 4, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(corrnet_kernel<false>, grid, dim3(kCnThreads), kCnLdsFloats * 4, (hipStream_t)stream, a);
    return itermvs_launch_status();
}

extern "C" int itermvs_corrnet(const float* x, int64_t x_sn, const float* const* weights, const int32_t* seg_end, int32_t n_seg,
                               int32_t M, int32_t H, int32_t W, float* out, int64_t out_sn, float* out2, int64_t out2_sn,
                               void* stream) {
    return launch_corrnet(x, x_sn, weights, seg_end, n_seg, M, H, W, out, out_sn, out2, out2_sn, false, stream);
}

extern "C" int itermvs_corrnet_bf16x3(const float* x, int64_t x_sn, const float* const* weights, const int32_t* seg_end, int32_t n_seg,
                                      int32_t M, int32_t H, int32_t W, float* out, int64_t out_sn, float* out2, int64_t out2_sn,
                                      void* stream) {
    return launch_corrnet(x, x_sn, weights, seg_end, n_seg, M, H, W, out, out_sn, out2, out2_sn, true, stream);
}
