// itermvs_fpn_level: one level of FeatureNet's top-down path in ONE launch (models/net.py:46-49 in training form, :60-63 in
// test form):
//
//   t   = F.interpolate(top, scale_factor=2, mode="bilinear") + inner(lat)       1x1, CL -> 48, + bias
//   out = output(t)                                                               3x3, 48 -> COUT, + bias
//
// As two launches, t (48 channels: 79 MB at level 1 of cfg 1) is written once and read once; here it only exists as an
// (8+2) x (32+2) tile in LDS (level 2 also writes its interior: it is level 1's `top`).  A persistent workgroup of eight
// waves per CU walks 8 x 32 output tiles:
//   1. the 6 x 18 patch of `top` the tile's up-sampling touches -> LDS (fetched into registers during the previous tile);
//   2. GEMM 1 on v_mfma_f32_16x16x4_f32: A = inner's weights (registers), B = `lat` straight from its NCHW planes (16
//      consecutive positions of the flattened 10 x 34 region per lane group), D = 48 channels x 16 positions; the epilogue
//      adds the bilinear sample of the LDS patch (same arithmetic as bilinear_up_kernel) and stores t to LDS, zeros outside
//      the image (the 3x3 layer's padding);
//   3. GEMM 2: A = output's weights in LDS (operand order [tap][k-step][q][co], staged once per workgroup), B = t from LDS
//      at the tap's displacement; a wave owns one output row (two 16-position groups share every A read); D is written
//      channels-last in the feature storage type (fp32 / fp16 / bf16, itermvs_dtype) and, optionally, as NCHW planes.
#include "conv_epilogue.hpp"

namespace itermvs {

constexpr int kFpTH = 8, kFpTW = 32, kFpThreads = 512, kFpWaves = 8;
constexpr int kFpMid = 48;                                   // channels of t
constexpr int kFpRR = kFpTH + 2, kFpRC = kFpTW + 2;          // t region
constexpr int kFpTP = 36, kFpTPL = kFpRR * kFpTP + 8;        // row pitch / plane stride 368 = 16 (mod 32): stride-1 B reads
static_assert(kFpTPL % 32 == 16, "t plane stride");
constexpr int kFpCR = kFpTH / 2 + 2, kFpCC = kFpTW / 2 + 2, kFpCP = 19;       // patch of `top`: 6 x 18, pitch 19
constexpr int kFpNB1 = (kFpRR * kFpRC + 15) / 16;            // 22 position groups of GEMM 1
constexpr int kFpNB1W = (kFpNB1 + kFpWaves - 1) / kFpWaves;  // per wave: 3
constexpr int kFpTopPer = (kFpMid * kFpCR * kFpCC + kFpThreads - 1) / kFpThreads;

struct FpnArgs {
    const float* lat;       // [N, CL, H, W]
    const float* top;       // [N, 48, H/2, W/2]
    const float* w_in;      // [CL/4][4][48] operand order, then 48 biases
    const float* w_out;     // [9][12][4][COUT] operand order, then COUT biases
    void* out;              // [N, H, W, COUT] channels-last, storage type FT
    float* out_planar;      // optional [N, COUT, H, W]
    float* t_out;           // optional [N, 48, H, W]
    int N, H, W, tiles_x, tiles_y;
};

template <int CL, int MB2, int FT>
__global__ __launch_bounds__(kFpThreads) void fpn_kernel(FpnArgs a, int tiles) {
    constexpr int KS1 = CL / 4, COUT = 16 * MB2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* __restrict__ T = lds;                                   // [48][kFpTPL]
    float* __restrict__ TOP = lds + kFpMid * kFpTPL;               // [48][6][19]
    float* __restrict__ WL = TOP + kFpMid * kFpCR * kFpCP;         // [9][12][4][COUT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int H = a.H, W = a.W, Hc = H >> 1, Wc = W >> 1;
    const int plane = H * W, cplane = Hc * Wc;

    // ---- once per workgroup: output's weights -> LDS, inner's weights and both biases -> registers ----
    for (int i = tid; i < 9 * 12 * 4 * COUT; i += kFpThreads) WL[i] = a.w_out[i];
    float a1[KS1][3];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) a1[ks][mb] = a.w_in[(ks * 4 + q) * kFpMid + mb * 16 + l16];
    float b1[3][4], b2[MB2][4];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) b1[mb][r] = a.w_in[CL * kFpMid + mb * 16 + q * 4 + r];
#pragma unroll
    for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) b2[mb][r] = a.w_out[9 * 12 * 4 * COUT + mb * 16 + q * 4 + r];

    // next tile's operands, fetched into registers while the current tile computes
    float topv[kFpTopPer];
    float latv[kFpNB1W][KS1];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int oy0 = ty * kFpTH, ox0 = tx * kFpTW;
        const float* __restrict__ tp = a.top + (int64_t)n * kFpMid * cplane;
        const int cy0 = (oy0 >> 1) - 1, cx0 = (ox0 >> 1) - 1;
#pragma unroll
        for (int i = 0; i < kFpTopPer; ++i) {
            const int e = min(tid + i * kFpThreads, kFpMid * kFpCR * kFpCC - 1);
            const int c = e / (kFpCR * kFpCC), rem = e - c * (kFpCR * kFpCC);
            const int r = rem / kFpCC, x = rem - r * kFpCC;
            const int gy = min(max(cy0 + r, 0), Hc - 1), gx = min(max(cx0 + x, 0), Wc - 1);
            topv[i] = tp[c * cplane + gy * Wc + gx];
        }
        const float* __restrict__ lp = a.lat + (int64_t)n * CL * plane;
#pragma unroll
        for (int j = 0; j < kFpNB1W; ++j) {
            const int p = min((wave + j * kFpWaves) * 16 + l16, kFpRR * kFpRC - 1);
            const int ry = p / kFpRC, rx = p - ry * kFpRC;
            const int gy = min(max(oy0 - 1 + ry, 0), H - 1), gx = min(max(ox0 - 1 + rx, 0), W - 1);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) latv[j][ks] = lp[(ks * 4 + q) * plane + gy * W + gx];
        }
    };
    fetch(blockIdx.x);

#pragma unroll 1
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int oy0 = ty * kFpTH, ox0 = tx * kFpTW;
        const int cy0 = (oy0 >> 1) - 1, cx0 = (ox0 >> 1) - 1;

        // 1. patch of `top` -> LDS
#pragma unroll
        for (int i = 0; i < kFpTopPer; ++i) {
            const int e = tid + i * kFpThreads;
            if (e < kFpMid * kFpCR * kFpCC) {
                const int c = e / (kFpCR * kFpCC), rem = e - c * (kFpCR * kFpCC);
                const int r = rem / kFpCC, x = rem - r * kFpCC;
                TOP[(c * kFpCR + r) * kFpCP + x] = topv[i];
            }
        }
        __syncthreads();      // TOP complete; every wave is past the previous tile's GEMM 2 (T may be overwritten)

        // 2. t = inner(lat) + bias + up2(top) on the 10 x 34 region -> LDS
#pragma unroll
        for (int j = 0; j < kFpNB1W; ++j) {
            const int nb = wave + j * kFpWaves;                    // wave-uniform
            if (nb < kFpNB1) {
                f32x4 acc[3];
#pragma unroll
                for (int mb = 0; mb < 3; ++mb) acc[mb] = f32x4{b1[mb][0], b1[mb][1], b1[mb][2], b1[mb][3]};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                    for (int mb = 0; mb < 3; ++mb)
                        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks][mb], latv[j][ks], acc[mb], 0, 0, 0);
                const int p = nb * 16 + l16;
                const int pc = min(p, kFpRR * kFpRC - 1);
                const int ry = pc / kFpRC, rx = pc - ry * kFpRC;
                const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
                const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
                // F.interpolate(scale_factor=2, bilinear, align_corners=False): the arithmetic of conv_epilogue.hpp (ADD == 2)
                float sy = ((float)gy + 0.5f) * 0.5f - 0.5f, sx = ((float)gx + 0.5f) * 0.5f - 0.5f;
                sy = sy < 0.0f ? 0.0f : sy;
                sx = sx < 0.0f ? 0.0f : sx;
                int y0 = (int)sy, x0 = (int)sx;
                y0 = y0 > Hc - 1 ? Hc - 1 : y0;
                x0 = x0 > Wc - 1 ? Wc - 1 : x0;
                const int y1 = y0 + (y0 < Hc - 1 ? 1 : 0), x1 = x0 + (x0 < Wc - 1 ? 1 : 0);
                const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
                const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
                // local patch coordinates; positions outside the image are never used, keep their indices in range
                const int j0 = min(max(y0 - cy0, 0), kFpCR - 1), j1 = min(max(y1 - cy0, 0), kFpCR - 1);
                const int i0 = min(max(x0 - cx0, 0), kFpCC - 1), i1 = min(max(x1 - cx0, 0), kFpCC - 1);
                const float* __restrict__ tq = TOP + (q * 4) * (kFpCR * kFpCP);
                const int o00 = j0 * kFpCP + i0, o01 = j0 * kFpCP + i1, o10 = j1 * kFpCP + i0, o11 = j1 * kFpCP + i1;
                float* __restrict__ td = T + (q * 4) * kFpTPL + ry * kFpTP + rx;
                const bool interior = inside && ry >= 1 && ry <= kFpTH && rx >= 1 && rx <= kFpTW && p < kFpRR * kFpRC;
                float* __restrict__ tg = a.t_out ? a.t_out + ((int64_t)n * kFpMid + q * 4) * plane + gy * W + gx : nullptr;
#pragma unroll
                for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* __restrict__ tc = tq + (mb * 16 + r) * (kFpCR * kFpCP);
                        const float upper = tc[o00] * lx0 + tc[o01] * lx1;
                        const float lower = tc[o10] * lx0 + tc[o11] * lx1;
                        const float v = acc[mb][r] + (upper * ly0 + lower * ly1);
                        if (p < kFpRR * kFpRC) td[(mb * 16 + r) * kFpTPL] = inside ? v : 0.0f;
                        if (tg && interior) tg[(int64_t)(mb * 16 + r) * plane] = v;
                    }
            }
        }
        __syncthreads();      // T complete

        if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);

        // 3. out = output(t) + bias: wave w -> output row w, two groups of 16 positions
        {
            f32x4 acc[2][MB2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int mb = 0; mb < MB2; ++mb) acc[cb][mb] = f32x4{b2[mb][0], b2[mb][1], b2[mb][2], b2[mb][3]};
            const float* __restrict__ bp = T + q * kFpTPL + wave * kFpTP + l16;
            const float* __restrict__ ap = WL + q * COUT + l16;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    float av[MB2];
#pragma unroll
                    for (int mb = 0; mb < MB2; ++mb) av[mb] = ap[(tap * 12 + ks) * 4 * COUT + mb * 16];
                    const float bv0 = bp[ks * 4 * kFpTPL + ky * kFpTP + kx];
                    const float bv1 = bp[ks * 4 * kFpTPL + ky * kFpTP + kx + 16];
#pragma unroll
                    for (int mb = 0; mb < MB2; ++mb) {
                        acc[0][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv0, acc[0][mb], 0, 0, 0);
                        acc[1][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv1, acc[1][mb], 0, 0, 0);
                    }
                }
            }
            const int gy = oy0 + wave;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int gx = ox0 + cb * 16 + l16;
                if (gy < H && gx < W) {
                    const int64_t px = ((int64_t)n * H + gy) * W + gx;
#pragma unroll
                    for (int mb = 0; mb < MB2; ++mb) {
                        const f32x4 v = acc[cb][mb];
                        const int co = mb * 16 + q * 4;
                        if constexpr (FT == ITERMVS_F32) {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + px * COUT + co) = v;
                        } else {
                            uint2 pk;
                            if constexpr (FT == ITERMVS_F16) {
                                pk.x = epi_to_f16(v[0]) | (epi_to_f16(v[1]) << 16);
                                pk.y = epi_to_f16(v[2]) | (epi_to_f16(v[3]) << 16);
                            } else {
                                pk.x = epi_to_bf16(v[0]) | (epi_to_bf16(v[1]) << 16);
                                pk.y = epi_to_bf16(v[2]) | (epi_to_bf16(v[3]) << 16);
                            }
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + px * COUT + co) = pk;
                        }
                        if (a.out_planar) {
                            float* __restrict__ op = a.out_planar + ((int64_t)n * COUT + co) * plane + gy * W + gx;
#pragma unroll
                            for (int r = 0; r < 4; ++r) op[(int64_t)r * plane] = v[r];
                        }
                    }
                }
            }
        }
    }
}

template <int CL, int MB2>
static int launch_fpn(const FpnArgs& a, int dtype, int tiles, hipStream_t stream) {
    constexpr int kLds = (kFpMid * kFpTPL + kFpMid * kFpCR * kFpCP + 9 * 12 * 4 * 16 * MB2) * 4;
    static_assert(kLds <= 160 * 1024, "LDS budget");
    const int grid = tiles < itermvs_num_cus() ? tiles : itermvs_num_cus();
#define ITERMVS_FPN_LAUNCH(FT)                                                                                          \
    {                                                                                                                   \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(fpn_kernel<CL, MB2, FT>),             \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;    \
        if (!ok) return ITERMVS_ERR_LAUNCH;                                                                             \
        hipLaunchKernelGGL((fpn_kernel<CL, MB2, FT>), dim3(grid), dim3(kFpThreads), kLds, stream, a, tiles);           \
    }
    if (dtype == ITERMVS_F32) ITERMVS_FPN_LAUNCH(ITERMVS_F32)
    else if (dtype == ITERMVS_F16) ITERMVS_FPN_LAUNCH(ITERMVS_F16)
    else ITERMVS_FPN_LAUNCH(ITERMVS_BF16)
#undef ITERMVS_FPN_LAUNCH
    return itermvs_launch_status();
}

}  // namespace itermvs

extern "C" int itermvs_fpn_level(const float* lat, int32_t CL, const float* top, int32_t N, int32_t H, int32_t W, const float* w_in,
                                 const float* w_out, int32_t COUT, void* out, int32_t out_dtype, float* out_planar, float* t_out,
                                 void* stream) {
    using namespace itermvs;
    ITERMVS_RETURN_IF(!lat || !top || !w_in || !w_out || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1), ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(!((CL == 16 && COUT == 16) || (CL == 32 && COUT == 32)), ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(out_dtype != ITERMVS_F32 && out_dtype != ITERMVS_F16 && out_dtype != ITERMVS_BF16, ITERMVS_ERR_DTYPE);
    ITERMVS_RETURN_IF((int64_t)48 * H * W > 0x7fffffff, ITERMVS_ERR_DIMS);
    FpnArgs a;
    a.lat = lat; a.top = top; a.w_in = w_in; a.w_out = w_out; a.out = out; a.out_planar = out_planar; a.t_out = t_out;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + kFpTW - 1) / kFpTW; a.tiles_y = (H + kFpTH - 1) / kFpTH;
    const int64_t tiles = (int64_t)N * a.tiles_x * a.tiles_y;
    ITERMVS_RETURN_IF(tiles > 0x7fffffff, ITERMVS_ERR_DIMS);
    if (CL == 16) return launch_fpn<16, 1>(a, out_dtype, (int)tiles, (hipStream_t)stream);
    return launch_fpn<32, 2>(a, out_dtype, (int)tiles, (hipStream_t)stream);
}
