// itermvs_conv2d, matrix-core path: implicit-GEMM 2-D convolution on v_mfma_f32_16x16x4_f32.
//
// The convolutions of the path are small dense contractions (K = Cin*k*k = 27..432, Cout = 8..256,
// 5k..400k pixels).  A one-thread-per-pixel VALU kernel cannot fill 256 CUs on them (the ConvGRU
// gate conv is 253 MFLOP-equivalents but only 320 waves), so they run on the matrix cores:
//   D[cout, pixel] += W[cout, k] * X[k, pixel],  k = (tap, ci),  16 x 16 x 4 per instruction,
// f32 in / f32 accumulate -- bit-for-bit a k-ordered fmaf chain (exact fp32, no TF32 on gfx950).
//   * A operand = weights: lane l holds W[cout = m0 + (l & 15)][k = 4t + (l >> 4)], read from the packed
//     [tap][Cin_pad][Cout_pad] layout (64-byte runs, identical for every wave -> L1/L2 resident);
//   * B operand = inputs: lane l holds X[k][pixel = p0 + (l & 15)], gathered straight from the NCHW
//     planes (16 consecutive pixels = one 64-byte run; zero for out-of-image taps);
//   * a wave owns MB x NB tiles (16*MB output channels x 16*NB pixels): MB + NB loads feed MB*NB MFMAs;
//   * D has pixels along lanes (col = lane & 15), so every store instruction writes 64-byte runs of
//     one output plane; bias / residual / activation / ConvGRU gate math are applied on the
//     accumulators (same epilogues as conv.hip).
// K is ordered (tap, ci) with Cin padded to a multiple of 4 (zero weights), so the four k-slots of a
// step share one tap: the spatial offset is computed once per tap and the inner loop is
// `load A, load B, mfma` with two pointer bumps.
#include <stdlib.h>

#include "common.hpp"
#include "conv_epilogue.hpp"

namespace itermvs {

struct MfmaArgs {
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
    int ksize, stride, pad, dil, act, add_mode, out_nhwc;
};

__device__ __forceinline__ EpilogueArgs make_epilogue(const MfmaArgs& a, int n, int P) {
    EpilogueArgs e;
    e.out = epi_out_base(a.out, (int64_t)n * a.out_sn, a.out_nhwc);
    e.out2 = a.out2 ? a.out2 + (int64_t)n * a.Cout * P : nullptr;
    e.add = a.add ? a.add + (int64_t)n * a.add_sn : nullptr;
    e.aux1 = a.aux1 ? a.aux1 + (int64_t)n * a.aux1_sn : nullptr;
    e.aux2 = a.aux2 ? a.aux2 + (int64_t)n * a.aux2_sn : nullptr;
    e.Cout = a.Cout; e.P = P; e.act = a.act;
    e.add_mode = a.add_mode; e.Hout = a.Hout; e.Wout = a.Wout; e.out_nhwc = a.out_nhwc;
    return e;
}

// Operand fetch: every load is a BUFFER load through a wave-uniform 128-bit descriptor --
//   address = descriptor base (SGPRs) + per-lane 32-bit byte offset (VGPR, constant per tap) + SGPR offset
//   (advances with the k-step) -- so the k-loop carries NO vector address arithmetic, and the hardware
//   bounds check returns 0 for (a) taps outside the image (their lane offset is set out of range) and
//   (b) the zero-padded channels beyond Cin.  (PMC on the flat-pointer version: 11-21 VALU instructions
//   per MFMA -- 64-bit address math, clamps and selects -- made the kernels VALU-issue bound.)
constexpr uint32_t kOob = 0x7fffffffu;   // lane offset that is out of range for every descriptor used here

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// U consecutive k-steps of one tap: all loads first, then the MFMAs
template <int MB, int NB, int U>
__device__ __forceinline__ void k_group(f32x4 (&acc)[MB][NB], __amdgpu_buffer_rsrc_t wr, uint32_t wv, uint32_t ws,
                                        uint32_t wstep_b, __amdgpu_buffer_rsrc_t ir, const uint32_t (&iv)[NB],
                                        uint32_t is, uint32_t istep_b) {
    float av[U][MB], bv[U][NB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[u][mb] = bload(wr, wv + mb * 64, ws + u * wstep_b);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[u][nb] = bload(ir, iv[nb], is + u * istep_b);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mb], bv[u][nb], acc[mb][nb], 0, 0, 0);
}

template <int MB, int NB, int KS>
__global__ void __launch_bounds__(256) conv_mfma_kernel(const MfmaArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kslot = lane >> 4, l16 = lane & 15;
    const int n = blockIdx.z;
    const int seg = (n >= a.seg_end[0]) + (n >= a.seg_end[1]);
    const int P = a.Hout * a.Wout;
    const int m0 = blockIdx.y * (MB * 16);
    const int pbase = (blockIdx.x * 4 + wave) * (NB * 16);
    if (pbase >= P) return;   // whole wave; no block-level synchronisation in this kernel

    int oy[NB], ox[NB];
    bool pv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int p = pbase + nb * 16 + l16;
        pv[nb] = p < P;
        const int pc = pv[nb] ? p : 0;
        oy[nb] = pc / a.Wout;
        ox[nb] = pc - oy[nb] * a.Wout;
    }
    f32x4 acc[MB][NB];
    conv_bias_init<MB, NB>(acc, a.bias[seg], a.Cout, m0, kslot);

    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
    // descriptors from wave-uniform values only (kernel arguments, blockIdx)
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (int64_t)n * a.in_sn), 0, (int)(a.Cin * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.weight[seg] + m0), 0, (int)((uint32_t)(KS * KS) * a.CinPad * a.CoutPad * 4u), 0x00020000);
    const uint32_t wv = (uint32_t)(kslot * a.CoutPad + l16) * 4u;
    const uint32_t wstep_b = 16u * a.CoutPad;          // bytes per k-step in the packed weights
    const uint32_t istep_b = 16u * plane;              // bytes per k-step in the input planes (4 channels)
    const uint32_t kplane_b = (uint32_t)kslot * plane * 4u;
    const int steps = a.CinPad >> 2;
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        uint32_t iv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int iy = oy[nb] * a.stride - a.pad + ky * a.dil;
            const int ix = ox[nb] * a.stride - a.pad + kx * a.dil;
            const bool ok = pv[nb] && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            iv[nb] = ok ? kplane_b + (uint32_t)(iy * a.Win + ix) * 4u : kOob;
        }
        uint32_t ws = (uint32_t)tap * a.CinPad * a.CoutPad * 4u, is = 0;
        // k-steps in groups of 4: the 4*(MB+NB) loads of a group are in flight before its first MFMA
        int st = 0;
        for (; st + 4 <= steps; st += 4) {
            k_group<MB, NB, 4>(acc, wr, wv, ws, wstep_b, ir, iv, is, istep_b);
            ws += 4 * wstep_b;
            is += 4 * istep_b;
        }
        for (; st < steps; ++st) {
            k_group<MB, NB, 1>(acc, wr, wv, ws, wstep_b, ir, iv, is, istep_b);
            ws += wstep_b;
            is += istep_b;
        }
    }
    // D: col (pixel) = lane & 15, row (cout) = (lane >> 4) * 4 + r
    uint32_t pix_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) pix_off[nb] = pv[nb] ? (uint32_t)(pbase + nb * 16 + l16) * 4u : kEpiOob;
    conv_epilogue<MB, NB>(make_epilogue(a, n, P), acc, m0, kslot, pix_off, oy, ox);
}

// ---------------------------------------------------------------------------------------------
// FeatureNet's lateral layers (models/net.py:46,49): out = conv1x1(x) + bias + F.interpolate(coarse, x2, bilinear).
// In the generic epilogue above every output element gathers its four coarse taps from global memory (192 gather loads
// per lane against 28 operand loads and 48 MFMAs: 49 us for a layer whose traffic is worth 23 us, and 2.3x its bytes
// moved through the L1).  Here a workgroup owns a 4 x 64 output tile (one row per wave, four 16-pixel slots); the
// (4/2 + 2) x (64/2 + 2) coarse patch of all 48 channels is staged ONCE in LDS, border-replicated (so every tap pair is
// (j, j + 1) / (i, i + 1) in patch coordinates), the operand roles of the MFMA are swapped so that a lane holds four
// consecutive pixels of one channel (dwordx4 stores, eight LDS reads for its twelve taps), and the operand / patch loads
// are batched (two memory round trips per tile).  The arithmetic and its order are those of the generic epilogue
// (== bilinear_up_kernel): results are bit-identical.  Measured 49 -> 36 us (level 1), 17.9 -> 14 us (level 2).
// ---------------------------------------------------------------------------------------------
constexpr int kLatTH = 4, kLatTW = 64;                       // output tile (rows = waves)
constexpr int kLatPH = kLatTH / 2 + 2, kLatPW = kLatTW / 2 + 2;   // coarse patch 4 x 34
constexpr int kLatRS = 37;                                   // LDS row stride; channel stride 4 * 37 = 148 floats: the two
constexpr int kLatCS = kLatPH * kLatRS;                      // channel quads of a half-wave land 16 banks apart

template <int MB>
__global__ void __launch_bounds__(256, 4) lateral_up2_kernel(const MfmaArgs a, const int tiles_x) {
    __shared__ float patch[MB * 16 * kLatCS];
    constexpr int NB = kLatTW / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kslot = lane >> 4, l16 = lane & 15;
    const int n = blockIdx.z;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int P = a.Hout * a.Wout;
    const int Hc = a.Hout >> 1, Wc = a.Wout >> 1;

    // ---- operand set-up of this wave's row: 16 * MB channels x 64 pixels ----
    // Operand roles are SWAPPED against conv_mfma_kernel: the pixels are the A operand (rows), the weights the B operand
    // (columns), so D[pixel, channel] puts FOUR CONSECUTIVE PIXELS of one channel into a lane (pixel 4*kslot + r, channel
    // l16): one dwordx4 store per accumulator into the NCHW plane instead of four dword stores into four planes, and the
    // twelve taps of those four pixels are eight LDS reads.  The k-ordered sums are the same, element for element.
    const int oy = ty * kLatTH + wave;
    const bool row_ok = oy < a.Hout;
    uint32_t iv[NB];
    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ox = tx * kLatTW + nb * 16 + l16;
        iv[nb] = (row_ok && ox < a.Wout) ? (uint32_t)kslot * plane * 4u + (uint32_t)(oy * a.Win + ox) * 4u : kOob;
    }
    f32x4 acc[MB][NB];
    {
        const __amdgpu_buffer_rsrc_t rb = epi_rsrc(a.bias[0], a.bias[0] ? (uint32_t)a.Cout * 4u : 0u);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const float bs = a.bias[0] ? epi_load(rb, (uint32_t)(mb * 16 + l16) * 4u) : 0.0f;    // 0 beyond Cout
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{bs, bs, bs, bs};
        }
    }
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (int64_t)n * a.in_sn), 0, (int)(a.Cin * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight[0], 0, (int)((uint32_t)a.CinPad * a.CoutPad * 4u), 0x00020000);
    const uint32_t wv = (uint32_t)(kslot * a.CoutPad + l16) * 4u;
    const uint32_t wstep_b = 16u * a.CoutPad, istep_b = 16u * plane;
    const int steps = a.CinPad >> 2;
    // four k-steps: loads, then (later) their MFMAs -- the loads of the first group and of the first patch half are issued
    // back to back, so a workgroup pays two memory round trips before its epilogue instead of five
    float gw[4][MB], gx[4][NB];
    auto gemm_load = [&](int st0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) gw[u][mb] = bload(wr, wv + mb * 64, (uint32_t)(st0 + u) * wstep_b);   // beyond CinPad: 0
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) gx[u][nb] = bload(ir, iv[nb], (uint32_t)(st0 + u) * istep_b);        // beyond Cin: 0
        }
    };
    auto gemm_mfma = [&](int st0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (st0 + u < steps) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gx[u][nb], gw[u][mb], acc[mb][nb], 0, 0, 0);
            }
    };

    // ---- the coarse patch: wave w takes channels w + 4u (u = 0..11) in two halves; lanes 0..33 hold one patch row per load ----
    const uint32_t cplane_b = (uint32_t)(Hc * Wc) * 4u;
    const __amdgpu_buffer_rsrc_t rc = epi_rsrc(a.add + (int64_t)n * a.add_sn, (uint32_t)a.Cout * cplane_b);
    uint32_t pvoff, rowoff[kLatPH];
    {
        const int cy0 = ty * (kLatTH / 2) - 1, cx0 = tx * (kLatTW / 2) - 1;
        int cx = cx0 + lane;
        cx = cx < 0 ? 0 : (cx > Wc - 1 ? Wc - 1 : cx);
        pvoff = lane < kLatPW ? (uint32_t)cx * 4u : kOob;
#pragma unroll
        for (int i = 0; i < kLatPH; ++i) {
            int cy = cy0 + i;
            cy = cy < 0 ? 0 : (cy > Hc - 1 ? Hc - 1 : cy);
            rowoff[i] = (uint32_t)(cy * Wc) * 4u;
        }
    }
    constexpr int kHalf = MB * 2;                      // channels per wave and half (MB * 16 / 4 waves / 2)
    float pv_[kHalf][kLatPH];
    auto patch_load = [&](int half) {
#pragma unroll
        for (int u = 0; u < kHalf; ++u)
#pragma unroll
            for (int i = 0; i < kLatPH; ++i)           // channels >= Cout lie beyond the descriptor: 0
                pv_[u][i] = bload(rc, pvoff, (uint32_t)(wave + 4 * (half * kHalf + u)) * cplane_b + rowoff[i]);
    };
    auto patch_store = [&](int half) {
        if (lane < kLatPW) {
#pragma unroll
            for (int u = 0; u < kHalf; ++u)
#pragma unroll
                for (int i = 0; i < kLatPH; ++i) patch[(wave + 4 * (half * kHalf + u)) * kLatCS + i * kLatRS + lane] = pv_[u][i];
        }
    };
    gemm_load(0);
    patch_load(0);
    gemm_mfma(0);
    for (int st0 = 4; st0 < steps; st0 += 4) {
        gemm_load(st0);
        gemm_mfma(st0);
    }
    patch_store(0);
    patch_load(1);
    patch_store(1);
    __syncthreads();                                      // the patch is complete

    // ---- epilogue: + x2 bilinear up-sampling of the patch (same arithmetic as conv_epilogue_act<0, 2>) ----
    float sy = ((float)oy + 0.5f) * 0.5f - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    int y0 = (int)sy;
    y0 = y0 > Hc - 1 ? Hc - 1 : y0;
    const float ly1 = sy - (float)y0, ly0 = 1.0f - ly1;
    const int i0 = row_ok ? y0 - (ty * (kLatTH / 2) - 1) : 0;
    const bool vec_ok = (a.Wout & 3) == 0 && (((uintptr_t)a.out | (uintptr_t)(a.out_sn * 4)) & 15) == 0;   // uniform
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int oxb = tx * kLatTW + nb * 16 + kslot * 4;                  // this lane's four pixels: oxb .. oxb + 3
        // Their coarse columns x0 are (2m - 1, 2m, 2m, 2m + 1) for oxb = 4m, i.e. patch columns jb + (0, 1, 1, 2) with
        // jb = nb * 8 + kslot * 2: a STATIC tap pattern over four patch values per row.  At the left image border sx clamps
        // to 0 (x0 = 0, weights (1, 0)); the static pattern then reads the replicated border column with weight 1 and the
        // true column 0 with weight 0 -- the same value as F.interpolate's v[0] * 1 + v[1] * 0.
        float lx0[4], lx1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sx = ((float)(oxb + r) + 0.5f) * 0.5f - 0.5f;
            sx = sx < 0.0f ? 0.0f : sx;
            int x0 = (int)sx;
            x0 = x0 > Wc - 1 ? Wc - 1 : x0;
            lx1[r] = sx - (float)x0;
            lx0[r] = 1.0f - lx1[r];
        }
        const bool px_ok = row_ok && oxb < a.Wout;
        const int jb = nb * 8 + kslot * 2;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int co = mb * 16 + l16;
            const float* __restrict__ pc = patch + co * kLatCS + i0 * kLatRS + jb;
            const float t0[4] = {pc[0], pc[1], pc[2], pc[3]};
            const float t1[4] = {pc[kLatRS], pc[kLatRS + 1], pc[kLatRS + 2], pc[kLatRS + 3]};
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                constexpr int kTap[4] = {0, 1, 1, 2};
                const float top = t0[kTap[r]] * lx0[r] + t0[kTap[r] + 1] * lx1[r];
                const float bot = t1[kTap[r]] * lx0[r] + t1[kTap[r] + 1] * lx1[r];
                o[r] = acc[mb][nb][r] + (top * ly0 + bot * ly1);
            }
            if (px_ok && co < a.Cout) {
                float* __restrict__ dst = a.out + (int64_t)n * a.out_sn + (int64_t)co * P + (int64_t)oy * a.Wout + oxb;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(dst) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (oxb + r < a.Wout) dst[r] = o[r];
                }
            }
        }
    }
}

// Split-K variant for layers with too few output tiles to fill the chip (ConvGRU gates, heads, the
// coarse CorrNet layers: a few hundred 16x16 tiles, each a serial chain of up to 100 dependent k-steps).
// The four waves of a block share ONE 16-pixel x 16*MB-channel tile and each takes every fourth k-step;
// partial accumulators are summed through LDS in a fixed order (deterministic) and wave 0 runs the
// epilogue.  4x the waves, 4x shorter dependency chains.
template <int MB, int KS>
__global__ void __launch_bounds__(256) conv_mfma_splitk_kernel(const MfmaArgs a) {
    __shared__ float red[3][MB][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kslot = lane >> 4, l16 = lane & 15;
    const int n = blockIdx.z;
    const int seg = (n >= a.seg_end[0]) + (n >= a.seg_end[1]);
    const int P = a.Hout * a.Wout;
    const int m0 = blockIdx.y * (MB * 16);
    const int pbase = blockIdx.x * 16;
    const int p = pbase + l16;
    const bool pv = p < P;
    const int pc = pv ? p : 0;
    const int oy = pc / a.Wout, ox = pc - oy * a.Wout;
    f32x4 acc[MB][1];
    conv_bias_init<MB, 1>(acc, wave == 0 ? a.bias[seg] : nullptr, a.Cout, m0, kslot);   // the bias enters the sum once
    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (int64_t)n * a.in_sn), 0, (int)(a.Cin * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.weight[seg] + m0), 0, (int)((uint32_t)(KS * KS) * a.CinPad * a.CoutPad * 4u), 0x00020000);
    const uint32_t wv = (uint32_t)(kslot * a.CoutPad + l16) * 4u;
    const uint32_t wstep_b = 16u * a.CoutPad, istep_b = 16u * plane;
    const uint32_t kplane_b = (uint32_t)kslot * plane * 4u;
    const int steps = a.CinPad >> 2;
    // the wave id selects this wave's k-steps; made provably uniform for the scalar offset math
    const int wu = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const int iy = oy * a.stride - a.pad + ky * a.dil;
        const int ix = ox * a.stride - a.pad + kx * a.dil;
        const bool ok = pv && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        uint32_t iv[1];
        iv[0] = ok ? kplane_b + (uint32_t)(iy * a.Win + ix) * 4u : kOob;
        const uint32_t wt = (uint32_t)tap * a.CinPad * a.CoutPad * 4u;
        // this wave's k-steps of the tap: (tap * steps + st) % 4 == wave keeps the four waves balanced
        int st = (wu - tap * steps) & 3;
        for (; st + 4 < steps; st += 8) {        // two of this wave's steps per trip
            float av0[MB], av1[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                av0[mb] = bload(wr, wv + mb * 64, wt + st * wstep_b);
                av1[mb] = bload(wr, wv + mb * 64, wt + (st + 4) * wstep_b);
            }
            const float b0 = bload(ir, iv[0], st * istep_b), b1 = bload(ir, iv[0], (st + 4) * istep_b);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[mb], b0, acc[mb][0], 0, 0, 0);
                acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[mb], b1, acc[mb][0], 0, 0, 0);
            }
        }
        for (; st < steps; st += 4) k_group<MB, 1, 1>(acc, wr, wv, wt + st * wstep_b, wstep_b, ir, iv, st * istep_b, istep_b);
    }
    if (wave > 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][mb][r][lane] = acc[mb][0][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[mb][0][r] = ((acc[mb][0][r] + red[0][mb][r][lane]) + red[1][mb][r][lane]) + red[2][mb][r][lane];
    const uint32_t pix_off[1] = {pv ? (uint32_t)p * 4u : kEpiOob};
    const int py[1] = {oy}, px[1] = {ox};
    conv_epilogue<MB, 1>(make_epilogue(a, n, P), acc, m0, kslot, pix_off, py, px);
}

}  // namespace itermvs

using namespace itermvs;

// called from itermvs_conv2d (conv.hip) for the non-transposed convolutions when weight_format == 1
int itermvs_conv2d_mfma(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream) {
    MfmaArgs a;
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
    a.ksize = p->ksize; a.stride = p->stride; a.pad = p->pad; a.dil = p->dilation; a.act = p->act;
    a.add_mode = p->add_mode; a.out_nhwc = p->out_layout;
    const int P = hout * wout;
    const int mt = a.CoutPad / 16;
    // FeatureNet's lateral layers: 1x1 + bias + x2 bilinear up-sampled residual with the coarse patch staged in LDS
    if (p->ksize == 1 && p->add_mode == 1 && p->add && p->out_layout == 0 && !p->out2 && p->act == 0 && p->stride == 1 && p->pad == 0 &&
        p->n_seg == 1 && mt == 3 && !p->split_cout) {
        const int tiles_x = (wout + kLatTW - 1) / kLatTW, tiles_y = (hout + kLatTH - 1) / kLatTH;
        hipLaunchKernelGGL((lateral_up2_kernel<3>), dim3(tiles_x * tiles_y, 1, p->N), dim3(256), 0, stream, a, tiles_x);
        return itermvs_launch_status();
    }
    // largest register blocking (MB x NB tiles of 16 channels x 16 pixels per wave) that still yields
    // >= 2048 waves (2 per SIMD): bigger tiles need fewer loads per MFMA, more waves hide latency
    struct Cfg { int mb, nb; };
    const Cfg cfgs[] = {{3, 4}, {2, 4}, {3, 2}, {2, 2}, {1, 4}, {1, 2}, {3, 1}, {2, 1}, {1, 1}};
    Cfg pick = {1, 1};
    int64_t best_waves = -1;
    for (const Cfg& c : cfgs) {
        if (mt % c.mb != 0) continue;
        const int64_t waves = (int64_t)p->N * ((P + 16 * c.nb - 1) / (16 * c.nb)) * (mt / c.mb);
        if (waves >= 2048) { pick = c; best_waves = waves; break; }
        if (waves > best_waves) { pick = c; best_waves = waves; }   // otherwise: the most waves available
    }
    // too few tiles even at 16x16 and a k-loop long enough to split: four waves per tile (split-K)
    static const bool no_splitk = [] { const char* e = itermvs_tuning_env("ITERMVS_CONV_SPLITK"); return e && e[0] == '0'; }();
    const int ksteps = p->ksize * p->ksize * (a.CinPad / 4);
    const int64_t tiles16 = (int64_t)p->N * ((P + 15) / 16) * mt;       // waves of the <1,1> configuration
    if (!no_splitk && tiles16 < 8192 && ksteps >= 16) {
        const int mb = (mt % 2 == 0 && tiles16 / 2 >= 2048) ? 2 : 1;
        const dim3 grid((P + 15) / 16, mt / mb, p->N);
        if (mb == 2) {
            if (p->ksize == 3) hipLaunchKernelGGL((conv_mfma_splitk_kernel<2, 3>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((conv_mfma_splitk_kernel<2, 1>), grid, dim3(256), 0, stream, a);
        } else {
            if (p->ksize == 3) hipLaunchKernelGGL((conv_mfma_splitk_kernel<1, 3>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((conv_mfma_splitk_kernel<1, 1>), grid, dim3(256), 0, stream, a);
        }
        return itermvs_launch_status();
    }
    const int px_per_block = 4 * 16 * pick.nb;
    const dim3 grid((P + px_per_block - 1) / px_per_block, mt / pick.mb, p->N);
#define ITERMVS_LAUNCH(MB_, NB_)                                                                              \
    if (pick.mb == MB_ && pick.nb == NB_) {                                                                    \
        if (p->ksize == 3) hipLaunchKernelGGL((conv_mfma_kernel<MB_, NB_, 3>), grid, dim3(256), 0, stream, a); \
        else hipLaunchKernelGGL((conv_mfma_kernel<MB_, NB_, 1>), grid, dim3(256), 0, stream, a);               \
    }
    ITERMVS_LAUNCH(3, 4) ITERMVS_LAUNCH(2, 4) ITERMVS_LAUNCH(3, 2) ITERMVS_LAUNCH(2, 2) ITERMVS_LAUNCH(1, 4)
    ITERMVS_LAUNCH(1, 2) ITERMVS_LAUNCH(3, 1) ITERMVS_LAUNCH(2, 1) ITERMVS_LAUNCH(1, 1)
#undef ITERMVS_LAUNCH
    return itermvs_launch_status();
}
