// itermvs_conv2d, matrix-core path with LDS-staged input tiles.  EXPERIMENTAL: selected with
// ITERMVS_CONV_MFMA=lds; measured slower than the direct-gather kernel (conv_mfma.hip) on most layers
// of the path at cfg 1 (3.4 ms vs 2.9 ms per depth map), kept for the next round's tuning.
//
// PMC on the gather version (conv_mfma.hip) showed the waves parked in s_waitcnt 60-80 % of the time
// and 9-12 L1 accesses per MFMA: every tap re-gathers its B operand from global memory in four
// unaligned 64-byte pieces.  Here a workgroup owns a spatial tile of output pixels and, per chunk of
// input channels, copies the input tile + halo into LDS ONCE (coalesced rows, zero padding applied
// while copying); all k*k taps then read their B operands from LDS (ds_read_b32, conflict-free:
// the channel-plane stride is chosen so the four k-slots of a wave fall on disjoint banks).
// Global traffic per tile drops from 9x to (1 + halo)x and the k-loop has no vector-memory waits
// except the small, L1-resident weight (A operand) loads.
//   block = 4 waves; tile = TH rows x (16*TWT) columns of output pixels; wave w owns NB of the
//   TH*TWT 16-pixel row segments; MB x NB accumulators of v_mfma_f32_16x16x4_f32 per wave.
//   large tile: TH=8, TWT=2, NB=4 (256 pixels)   small tile: TH=4, TWT=1, NB=1 (64 pixels, used when
//   the layer has too few pixels to fill 256 CUs with large tiles).
#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct LdsConvArgs {
    const float* in;
    float* out;
    float* out2;
    const float* add;
    const float* aux1;
    const float* aux2;
    int64_t in_sn, out_sn, add_sn, aux1_sn, aux2_sn;
    const float* weight[3];   // packed [k*k][CinPad][CoutPad]
    const float* bias[3];
    int seg_end[3];
    int N, Cin, CinPad, Hin, Win, Cout, CoutPad, Hout, Wout;
    int stride, pad, dil, act;
    int tiles_x, in_h, in_w, plane;   // tile grid and staged-tile geometry (floats)
};

__device__ __forceinline__ float lds_epilogue(float v, int act, float add, float a1, float a2) {
    v += add;
    switch (act) {
        case 1: return fmaxf(v, 0.0f);
        case 2: return sigmoidf_(v);
        case 3: return tanhf(v);
        case 4: return sigmoidf_(v) * a1;                  // r * h            (module.py:63-64)
        case 5: return (1.0f - a2) * a1 + a2 * tanhf(v);   // (1-z) h + z q    (module.py:64-65)
        default: return v;
    }
}

constexpr int kChunk = 16;   // input channels staged per pass (missing channels are zero-filled in LDS)

// exact r / d for 0 <= r < 2^20 with a precomputed float reciprocal (staging index math without
// integer division)
__device__ __forceinline__ int fast_div(int r, int d, float inv) {
    int q = (int)(((float)r + 0.5f) * inv);
    q -= (q * d > r);
    q += ((q + 1) * d <= r);
    return q;
}

template <int MB, int NB, int KS, int TH, int TWT>
__global__ void __launch_bounds__(256) conv_mfma_lds_kernel(const LdsConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ tile = smem;                                   // [kChunk][plane]
    float* __restrict__ wl = smem + kChunk * a.plane;                  // [KS*KS][kChunk][16*MB]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kslot = lane >> 4, l16 = lane & 15;
    const int n = blockIdx.z;
    const int seg = (n >= a.seg_end[0]) + (n >= a.seg_end[1]);
    const int m0 = blockIdx.y * (MB * 16);
    const int tile_y = blockIdx.x / a.tiles_x, tile_x = blockIdx.x - tile_y * a.tiles_x;
    const int oy0 = tile_y * TH, ox0 = tile_x * (16 * TWT);
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;   // input coords of the staged tile origin

    // this wave's 16-pixel row segments: segment id = wave * NB + nb -> (row, column block)
    int ly[NB], lx[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int sg = wave * NB + nb;
        ly[nb] = sg / TWT;
        lx[nb] = (sg - ly[nb] * TWT) * 16 + l16;
    }
    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int hw = a.Hin * a.Win;
    const float* __restrict__ inb = a.in + (int64_t)n * a.in_sn;
    const float* __restrict__ wg = a.weight[seg] + m0;
    const int tile_px = a.in_h * a.in_w;
    const float inv_px = 1.0f / (float)tile_px, inv_w = 1.0f / (float)a.in_w;
    constexpr int WROW = 16 * MB;                        // staged weight row: this block's output channels
    constexpr int WTOT = KS * KS * kChunk * WROW;

    for (int c0 = 0; c0 < a.CinPad; c0 += kChunk) {
        __syncthreads();                                     // previous chunk fully consumed
        // ---- stage the input tile [16][in_h][in_w] (zero outside the image / beyond Cin) ---------
        for (int e = tid; e < kChunk * tile_px; e += 256) {
            const int c = fast_div(e, tile_px, inv_px);
            const int r = e - c * tile_px;
            const int y = fast_div(r, a.in_w, inv_w), x = r - y * a.in_w;
            const int gy = iy0 + y, gx = ix0 + x, gc = c0 + c;
            const bool ok = gc < a.Cin && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            const float v = ok ? inb[(int64_t)gc * hw + gy * a.Win + gx] : 0.0f;
            tile[c * a.plane + r] = v;
        }
        // ---- stage this chunk's weights [tap][16 ci][16*MB co] --------------------------------------
        for (int e = tid; e < WTOT; e += 256) {
            const int row = e / WROW, col = e - row * WROW;          // row = tap*16 + ci (compile-time divisor)
            const int tap = row / kChunk, ci = row - tap * kChunk;
            wl[e] = (c0 + ci < a.CinPad) ? wg[((size_t)tap * a.CinPad + c0 + ci) * a.CoutPad + col] : 0.0f;
        }
        __syncthreads();
        // ---- k*k taps x 4 k-steps, every operand from LDS --------------------------------------------
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            float av[4][MB], bv[4][NB];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[st][mb] = wl[(tap * kChunk + st * 4 + kslot) * WROW + mb * 16 + l16];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    bv[st][nb] = tile[(st * 4 + kslot) * a.plane + (ly[nb] * a.stride + ky * a.dil) * a.in_w +
                                      lx[nb] * a.stride + kx * a.dil];
            }
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][mb], bv[st][nb], acc[mb][nb], 0, 0, 0);
        }
    }
    // ---- epilogue: D col (pixel) = lane & 15, row (cout) = (lane >> 4) * 4 + r -----------------------
    const int P = a.Hout * a.Wout;
    const float* __restrict__ bias = a.bias[seg];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + mb * 16 + kslot * 4 + r;
            if (co >= a.Cout) continue;
            const float bs = bias ? bias[co] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int oy = oy0 + ly[nb], ox = ox0 + lx[nb];
                if (oy >= a.Hout || ox >= a.Wout) continue;
                const int64_t ch = (int64_t)co * P + oy * a.Wout + ox;
                const float ad = a.add ? a.add[(int64_t)n * a.add_sn + ch] : 0.0f;
                const float a1 = a.aux1 ? a.aux1[(int64_t)n * a.aux1_sn + ch] : 0.0f;
                const float a2 = a.aux2 ? a.aux2[(int64_t)n * a.aux2_sn + ch] : 0.0f;
                const float v = lds_epilogue(acc[mb][nb][r] + bs, a.act, ad, a1, a2);
                a.out[(int64_t)n * a.out_sn + ch] = v;
                if (a.out2) a.out2[((int64_t)n * a.Cout) * P + ch] = v;
            }
        }
}

// tile shapes: 0 = large (8 rows x 32 px, NB = 4), 1 = medium (8 rows x 16 px, NB = 2)
template <int MB, int KS>
static void launch_lds(const LdsConvArgs& a0, int shape, int mt, hipStream_t stream) {
    LdsConvArgs a = a0;
    const int th = 8, twt = shape == 0 ? 2 : 1;
    a.tiles_x = (a.Wout + 16 * twt - 1) / (16 * twt);
    const int tiles_y = (a.Hout + th - 1) / th;
    a.in_h = (th - 1) * a.stride + (KS - 1) * a.dil + 1;
    a.in_w = (16 * twt - 1) * a.stride + (KS - 1) * a.dil + 1;
    // channel-plane stride: the four k-slots (lanes 0-15, 16-31 | 32-47, 48-63) must hit disjoint banks.
    // stride 1: 16 consecutive banks per k-slot -> plane = 16 (mod 32); stride 2: every other bank -> odd plane.
    int plane = a.in_h * a.in_w;
    if (a.stride == 1) plane += (16 - (plane % 32) + 32) % 32;
    else plane |= 1;
    a.plane = plane;
    const size_t lds = ((size_t)kChunk * plane + (size_t)KS * KS * kChunk * 16 * MB) * sizeof(float);
    const dim3 grid(a.tiles_x * tiles_y, mt / MB, a.N);
    if (shape == 0) hipLaunchKernelGGL((conv_mfma_lds_kernel<MB, 4, KS, 8, 2>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((conv_mfma_lds_kernel<MB, 2, KS, 8, 1>), grid, dim3(256), lds, stream, a);
}

}  // namespace itermvs

using namespace itermvs;

// called from itermvs_conv2d (conv.hip); returns 1 when this variant does not apply (too few
// workgroups for the layer: the direct-gather kernel with 16-pixel tiles takes over)
int itermvs_conv2d_mfma_lds(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream) {
    if (p->stride > 2 || p->dilation > 2) return 1;
    LdsConvArgs a;
    a.in = p->in; a.out = p->out; a.out2 = p->out2; a.add = p->add; a.aux1 = p->aux1; a.aux2 = p->aux2;
    a.in_sn = p->in_sn; a.out_sn = p->out_sn; a.add_sn = p->add_sn; a.aux1_sn = p->aux1_sn; a.aux2_sn = p->aux2_sn;
    for (int i = 0; i < 3; ++i) {
        const int k = i < p->n_seg ? i : p->n_seg - 1;
        a.weight[i] = p->weight[k];
        a.bias[i] = p->bias[k];
        a.seg_end[i] = i < p->n_seg - 1 ? p->seg_end[i] : p->N;
    }
    a.N = p->N; a.Cin = p->Cin; a.CinPad = (p->Cin + 3) / 4 * 4; a.Hin = p->Hin; a.Win = p->Win;
    a.Cout = p->Cout; a.CoutPad = (p->Cout + 15) / 16 * 16; a.Hout = hout; a.Wout = wout;
    a.stride = p->stride; a.pad = p->pad; a.dil = p->dilation; a.act = p->act;
    const int mt = a.CoutPad / 16;
    const int mb = (mt % 3 == 0) ? 3 : ((mt % 2 == 0) ? 2 : 1);   // 48 / 32 / 16 output channels per wave
    auto blocks = [&](int twt) { return (int64_t)((wout + 16 * twt - 1) / (16 * twt)) * ((hout + 7) / 8) * (mt / mb) * p->N; };
    int shape;
    if (blocks(2) >= 512) shape = 0;
    else if (blocks(1) >= 512) shape = 1;
    else return 1;
#define ITERMVS_LDS_LAUNCH(MB_)                                                \
    if (mb == MB_) {                                                           \
        if (p->ksize == 3) launch_lds<MB_, 3>(a, shape, mt, stream);           \
        else launch_lds<MB_, 1>(a, shape, mt, stream);                         \
    }
    ITERMVS_LDS_LAUNCH(3) ITERMVS_LDS_LAUNCH(2) ITERMVS_LDS_LAUNCH(1)
#undef ITERMVS_LDS_LAUNCH
    return itermvs_launch_status();
}
