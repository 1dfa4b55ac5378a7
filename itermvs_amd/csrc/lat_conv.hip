// itermvs_lateral_conv3x3: a level of FeatureNet's top-down path in ONE launch (models/net.py:48-50, test mode :62-63)
//     intra = F.interpolate(coarse, scale_factor=2, mode="bilinear") + inner(fine)        1x1, Cf -> 48, bias
//     out   = output(intra)                                                                3x3, 48 -> Cout, bias, padding 1
// As two launches (lateral_up2_kernel, then conv_tile3) the 48-channel `intra` map of level 1 (78 MB at cfg 1) was written by
// the first and fetched -- plane by plane, 255 wave-level dword loads per 8 x 32 tile -- by the second: 35.6 + 49.3 us of the
// depth map.  Only the output convolution reads it (net.py:50), so here it never leaves the chip.  A persistent 8-wave workgroup
// owns an 8 x 32 tile of `out`; per 16-channel chunk c of `intra`:
//   A(c)  intra chunk on the (8+2) x (32+2) halo tile: the 1x1 layer on the exact fp32 matrix instruction straight from registers
//         (a lane fetched its pixel's fine channels in operand order: no LDS hop), plus the four bilinear taps of a lane's
//         four channels from a border-replicated 6 x 18 coarse patch in LDS (one ds_read_b128 per tap), zero outside the image
//         (the 3x3 layer's padding), split into three bf16 terms and stored to an LDS tile in conv_tile3's operand layout;
//   B(c)  the chunk's 9 taps of the 3x3 layer on v_mfma_f32_16x16x32_bf16 (bf16x3 arithmetic of conv_tile3.hip: six cross products of
//         the exact three-term splits, fp32 accumulation).
// The chunk tile is double-buffered, so A(c+1) of one wave runs beside B(c) of another (4 barriers per tile); the next tile's
// fine / coarse values are fetched into registers while B(2) multiplies.  The up-sampling arithmetic is operation for operation
// that of lateral_up2_kernel / bilinear_up_kernel (the fused F.interpolate, align_corners=False).
#include <type_traits>

#include "common.hpp"
#include "conv_epilogue.hpp"

namespace itermvs {

using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int kLcThreads = 512, kLcWaves = 8;
constexpr int kLcTH = 8, kLcTW = 32;
constexpr int kLcMW = kLcTW + 2, kLcMH = kLcTH + 2, kLcMPX = kLcMW * kLcMH;      // intra tile incl. halo: 10 x 34
constexpr int kLcMG = (kLcMPX + 15) / 16;                                        // 22 groups of 16 positions
constexpr int kLcGPW = (kLcMG + kLcWaves - 1) / kLcWaves;                        // 3 groups per wave (waves 6, 7: 2)
constexpr int kLcPLB = (kLcMPX * 16 + 255) / 256 * 256;                          // bytes per (plane, half) of a chunk tile
static_assert(kLcPLB >= kLcMG * 16 * 16, "the last group's surplus positions must land in the planes' padding");
constexpr int kLcChunkB = 6 * kLcPLB;
constexpr int kLcPW = kLcTW / 2 + 2, kLcPH = kLcTH / 2 + 2, kLcPPX = kLcPW * kLcPH;   // coarse patch 6 x 18
constexpr int kLcPS = 52;          // floats per patch position (48 channels + 4): 16 positions' b128 accesses cover the 64 banks once
constexpr int kLcPQ = 5;                                                         // 16-byte pieces per (channel, patch row): 20 columns fetched
constexpr int kLcPItems = 48 * kLcPH * kLcPQ;                                    // (channel, patch row, piece) staging items
constexpr int kLcPIT = (kLcPItems + kLcThreads - 1) / kLcThreads;                // 3 per thread
constexpr int kLcFQ = (kLcMW + 3) / 4, kLcFW = 4 * kLcFQ;                        // fine tile rows: 9 pieces = 36 columns (34 used)
constexpr int kLcWChunk = 9 * 3 * 16 * 32;                                       // split weights of (chunk, 16 output channels)
constexpr uint32_t kLcOob = 0x7fffffffu;

struct LatConvArgs {
    const float* fine;
    const float* coarse;
    const float* w_lat;      // [Cf][48] fp32 (weight_format 1 of the 1x1 layer)
    const float* b_lat;      // [48] or nullptr
    const void* w_out;       // bf16 [9][3][3][CoutPad][16] (weight_format 3)
    const float* b_out;      // [Cout] or nullptr
    float* out;
    float* out2;
    int64_t fine_sn, coarse_sn, out_sn;
    int N, H, W, Cout, CoutPad, out_nhwc, tiles_x, tiles_y, total, banded;
};

template <int KS, int MBO>
__global__ void __launch_bounds__(kLcThreads) lat_conv_kernel(const LatConvArgs a) {
    constexpr int NB = 2;                              // 16 output groups of 16 pixels over 8 waves
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    char* __restrict__ T0 = lc_smem;                                   // chunk tile, two buffers
    char* __restrict__ Wt = lc_smem + 2 * kLcChunkB;                   // [mb][chunk][tap][plane][16 rows][32 B]
    float* __restrict__ Pt = reinterpret_cast<float*>(Wt + MBO * 3 * kLcWChunk);      // coarse patch [position][kLcPS]
    float* __restrict__ Ft = Pt + kLcPPX * kLcPS;                                     // fine tile [channel][row][kLcFW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int half = q & 1, second = q >> 1;
    const uint32_t plane = (uint32_t)(a.H * a.W);
    const int Hc = a.H >> 1, Wc = a.W >> 1;
    const uint32_t cplane = (uint32_t)(Hc * Wc);
    const int P = a.H * a.W;

    // the 1x1 layer: A operand of v_mfma_f32_16x16x4_f32 = weight[out channel c*16 + l16][fine channel 4 s + q], resident
    float wl[3][KS];
    f32x4 bl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int s = 0; s < KS; ++s) wl[c][s] = a.w_lat[(4 * s + q) * 48 + c * 16 + l16];
#pragma unroll
        for (int r = 0; r < 4; ++r) bl[c][r] = a.b_lat ? a.b_lat[c * 16 + q * 4 + r] : 0.0f;
    }

    f32x4 bo[MBO];                   // the 3x3 layer's bias: accumulators start from it
#pragma unroll
    for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[mb][r] = a.b_out && mb * 16 + q * 4 + r < a.Cout ? a.b_out[mb * 16 + q * 4 + r] : 0.0f;

    struct Work { int n, oy0, ox0; };
    auto decode = [&](int w) {
        Work k;
        const int t2 = w / a.tiles_x;
        k.n = t2 / a.tiles_y;
        k.oy0 = (t2 - k.n * a.tiles_y) * kLcTH;
        k.ox0 = (w - t2 * a.tiles_x) * kLcTW;
        return k;
    };
    // this lane's intra positions: group wave + 8 g, position l16 (row-major over the 10 x 34 halo tile)
    int mpos[kLcGPW], my[kLcGPW], mx[kLcGPW];
#pragma unroll
    for (int g = 0; g < kLcGPW; ++g) {
        mpos[g] = (wave + g * kLcWaves) * 16 + l16;
        const int pc = min(mpos[g], kLcMPX - 1);
        my[g] = pc / kLcMW;
        mx[g] = pc - my[g] * kLcMW;
    }
    // Staging: every global access is a 16-byte load of four consecutive pixels of one plane.  fine: (channel, tile row, piece of
    // 4 columns) -> LDS [channel][row][36]; coarse: (channel, patch row, piece) -> LDS [position][channel].  The pieces start at
    // column ox0 - 1 (cx0 = ox0/2 - 1 for the patch): dword-aligned only; in the first tile column the first piece is fetched one
    // pixel to the right and shifted.
    constexpr int kFItems = 4 * KS * kLcMH * kLcFQ;
    constexpr int FIT = (kFItems + kLcThreads - 1) / kLcThreads;
    u32x4 fv[FIT], pv[kLcPIT];
    // piece i of tile k's loads: i < FIT the fine items, then the coarse items.  The loads of the NEXT tile are issued one or
    // two per phase: issued together (48 wave-level loads of 1 KB per workgroup) they filled the CU's address path and the
    // issuing waves stood at the instruction for 3-7 k cycles while the others waited at the next barrier
    // (profiles/r06/r06k_lat_conv_phase_stamps.txt)
    auto fetch_part = [&](const Work& k, int i) __attribute__((always_inline)) {
        if (i < FIT) {
            const int j = i;
            const __amdgpu_buffer_rsrc_t fr_ =
                __builtin_amdgcn_make_buffer_rsrc((void*)(a.fine + (int64_t)k.n * a.fine_sn), 0, (int)((uint32_t)(4 * KS) * plane * 4u), 0x00020000);
            const int item = tid + j * kLcThreads;
            const int ch = item / (kLcMH * kLcFQ), rem = item - ch * (kLcMH * kLcFQ);
            const int row = rem / kLcFQ, jq = rem - row * kLcFQ;
            const int gy = k.oy0 - 1 + row, gx0 = k.ox0 - 1 + 4 * jq;
            const bool ok = item < kFItems && gy >= 0 && gy < a.H && gx0 < a.W;
            const uint32_t go = ok ? ((uint32_t)ch * plane + (uint32_t)(gy * a.W + max(gx0, 0))) * 4u : kLcOob;
            fv[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(fr_, go, 0, 0));
        } else {
            const int j = i - FIT;
            const __amdgpu_buffer_rsrc_t cr_ =
                __builtin_amdgcn_make_buffer_rsrc((void*)(a.coarse + (int64_t)k.n * a.coarse_sn), 0, (int)(48u * cplane * 4u), 0x00020000);
            const int cy0 = (k.oy0 >> 1) - 1, cx0 = (k.ox0 >> 1) - 1;
            // piece fastest, then patch row, then channel: a wave's 64 pieces are ~13 runs of 80 contiguous bytes (13-26 cache
            // lines).  With the channel fastest every piece of a wave-level load lay in another line, the tile's 288 lines
            // (37-74 KB) did not survive in the 32 KB L1 until their other four pieces came by, and the kernel moved 227 MB
            // through the L2 for 33 MB of patches (profiles/r06/r06k_*)
            const int item = tid + j * kLcThreads;
            const int rem = item / kLcPQ, jq = item - rem * kLcPQ;
            const int ch = rem / kLcPH, pr = rem - ch * kLcPH;
            const int cy = min(max(cy0 + pr, 0), Hc - 1);                // rows: border-replicated here; columns: clamped when read
            const uint32_t go = item < kLcPItems ? ((uint32_t)ch * cplane + (uint32_t)(cy * Wc + max(cx0 + 4 * jq, 0))) * 4u : kLcOob;
            pv[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(cr_, go, 0, 0));
        }
    };
    constexpr int kParts = FIT + kLcPIT;
    static_assert(kParts == 6, "two loads per step");
    // the fetched registers of tile k -> LDS
    auto stage = [&](const Work& k) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < FIT; ++j) {
            const int item = tid + j * kLcThreads;
            if (j < FIT - 1 || item < kFItems) {
                const int ch = item / (kLcMH * kLcFQ), rem = item - ch * (kLcMH * kLcFQ);
                const int row = rem / kLcFQ, jq = rem - row * kLcFQ;
                u32x4 v = fv[j];
                if (k.ox0 == 0 && jq == 0) v = u32x4{0u, v[0], v[1], v[2]};             // fetched from column 0 instead of -1
                *reinterpret_cast<u32x4*>(Ft + (ch * kLcMH + row) * kLcFW + 4 * jq) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < kLcPIT; ++j) {
            const int item = tid + j * kLcThreads;
            if (j < kLcPIT - 1 || item < kLcPItems) {
                const int rem = item / kLcPQ, jq = item - rem * kLcPQ;
                const int ch = rem / kLcPH, pr = rem - ch * kLcPH;
                // element e is patch column 4 jq + e; in the first tile column piece 0 was fetched from image column 0 = patch column 1
                const int pc0 = 4 * jq + (k.ox0 == 0 && jq == 0 ? 1 : 0);
                float* __restrict__ d = Pt + (pr * kLcPW + pc0) * kLcPS + ch;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (pc0 + e < kLcPW) d[e * kLcPS] = __uint_as_float(pv[j][e]);
            }
        }
    };

    // B phase operand bases (rc_layer of res_chain.hip; K = 32 holds two 16-channel terms: B1 = [xh | xm], B3 = [xh | xl])
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pos = (wave + nb * kLcWaves) * 16 + l16;
        boff[nb] = ((pos / kLcTW) * kLcMW + (pos % kLcTW)) * 16;
    }
    const int x1o = (second ? 2 * kLcPLB : 0) + half * kLcPLB;
    const int x3o = (second ? 4 * kLcPLB : 0) + half * kLcPLB;
    const char* __restrict__ wa = Wt + l16 * 32 + half * 16;             // A1 = [wh | wh]; A2 = + 512
    const char* __restrict__ wa3 = wa + (second ? 0 : 1024);             // A3 = [wl | wh]

    // ---- per-tile state of the A phase: up-sampling weights (F.interpolate, align_corners=False, like bilinear_up_kernel), tap
    // positions in the patch (columns clamped to the image), the 1x1 layer's B operand (channel 4 s + q of the lane's positions) ----
    float lx0[kLcGPW], lx1[kLcGPW], ly0[kLcGPW], ly1[kLcGPW];
    bool inside[kLcGPW];
    int offa[kLcGPW], offb[kLcGPW];
    float fr[kLcGPW][KS];
    auto setup_a = [&](const Work& k) __attribute__((always_inline)) {
        const int jlo = k.ox0 == 0 ? 1 : 0, jhi = min(kLcPW - 1, Wc - 1 - ((k.ox0 >> 1) - 1));
#pragma unroll
        for (int g = 0; g < kLcGPW; ++g) {
            const int ja = min(max(mx[g] >> 1, jlo), jhi), jb = min(max((mx[g] >> 1) + 1, jlo), jhi);
            offa[g] = ((my[g] >> 1) * kLcPW + ja) * kLcPS + q * 4;
            offb[g] = ((my[g] >> 1) * kLcPW + jb) * kLcPS + q * 4;
            const int gy = k.oy0 - 1 + my[g], gx = k.ox0 - 1 + mx[g];
            inside[g] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            float sy = ((float)gy + 0.5f) * 0.5f - 0.5f, sx = ((float)gx + 0.5f) * 0.5f - 0.5f;
            sy = sy < 0.0f ? 0.0f : sy;
            sx = sx < 0.0f ? 0.0f : sx;
            int y0 = (int)sy, x0 = (int)sx;
            y0 = y0 > Hc - 1 ? Hc - 1 : y0;
            x0 = x0 > Wc - 1 ? Wc - 1 : x0;
            ly1[g] = sy - (float)y0;
            lx1[g] = sx - (float)x0;
            ly0[g] = 1.0f - ly1[g];
            lx0[g] = 1.0f - lx1[g];
#pragma unroll
            for (int s = 0; s < KS; ++s) fr[g][s] = Ft[((4 * s + q) * kLcMH + my[g]) * kLcFW + mx[g]];
        }
    };
    // A: chunk c of the intra tile -> chunk buffer `buf`.  The NG groups of this wave are interleaved: their matrix-instruction
    // chains and 4 NG patch reads are independent and issued before the first result is needed.  No lane guard on the stores:
    // group 21's positions 340..351 land in the padding of the planes (kLcPLB holds 352 positions).
    auto phase_a_n = [&](int c, int buf, auto ng_tag) __attribute__((always_inline)) {
        constexpr int NG = decltype(ng_tag)::value;
        char* __restrict__ tb = T0 + buf * kLcChunkB;
        f32x4 m[NG], t00[NG], t01[NG], t10[NG], t11[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // taps: patch rows (my >> 1, + 1) -- replicated at the image border when fetched --, columns clamped in setup_a
            t00[g] = *reinterpret_cast<const f32x4*>(Pt + offa[g] + c * 16);
            t01[g] = *reinterpret_cast<const f32x4*>(Pt + offb[g] + c * 16);
            t10[g] = *reinterpret_cast<const f32x4*>(Pt + offa[g] + kLcPW * kLcPS + c * 16);
            t11[g] = *reinterpret_cast<const f32x4*>(Pt + offb[g] + kLcPW * kLcPS + c * 16);
            m[g] = bl[c];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < NG; ++g) m[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[c][s], fr[g][s], m[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float top = t00[g][r] * lx0[g] + t01[g][r] * lx1[g];
                const float bot = t10[g][r] * lx0[g] + t11[g][r] * lx1[g];
                const float x = m[g][r] + (top * ly0[g] + bot * ly1[g]);
                v[r] = inside[g] ? x : 0.0f;
            }
            uint32_t h0, m0, l0, h1, m1, l1;
            split_pair(v[0], v[1], h0, m0, l0);
            split_pair(v[2], v[3], h1, m1, l1);
            char* __restrict__ d = tb + (q >> 1) * kLcPLB + mpos[g] * 16 + (q & 1) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * kLcPLB) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(d + 4 * kLcPLB) = u32x2{l0, l1};
        }
    };
    auto phase_a = [&](int c, int buf) __attribute__((always_inline)) {
        if (wave < kLcMG - (kLcGPW - 1) * kLcWaves) phase_a_n(c, buf, std::integral_constant<int, kLcGPW>{});      // wave-uniform
        else phase_a_n(c, buf, std::integral_constant<int, kLcGPW - 1>{});
    };
    // B: the 9 taps of chunk c from chunk buffer `buf`
    f32x4 acc[MBO][NB];
    auto phase_b = [&](int c, int buf) __attribute__((always_inline)) {
        const char* __restrict__ tb = T0 + buf * kLcChunkB;
        const char* __restrict__ x1 = tb + x1o;
        const char* __restrict__ x3 = tb + x3o;
        bf8 a1[2][MBO], a2[2][MBO], a3[2][MBO], b1[2][NB], b3[2][NB];
        auto read = [&](int tap, int set) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int to = (ky * kLcMW + kx) * 16;
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb) {
                const int o = ((mb * 3 + c) * 9 + tap) * 1536;
                a1[set][mb] = *reinterpret_cast<const bf8*>(wa + o);
                a2[set][mb] = *reinterpret_cast<const bf8*>(wa + o + 512);
                a3[set][mb] = *reinterpret_cast<const bf8*>(wa3 + o);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                b1[set][nb] = *reinterpret_cast<const bf8*>(x1 + boff[nb] + to);
                b3[set][nb] = *reinterpret_cast<const bf8*>(x3 + boff[nb] + to);
            }
        };
        read(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = tap & 1;
            if (tap + 1 < 9) read(tap + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[s][mb], b3[s][nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s][mb], b1[s][nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s][mb], b1[s][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // results of tile k (D: column = pixel l16, row = output channel 4 q + r: the shared epilogue's layout)
    auto store = [&](const Work& k) __attribute__((always_inline)) {
        uint32_t pix_off[NB];
        int py[NB], px[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int pos = (wave + nb * kLcWaves) * 16 + l16;
            const int oy = k.oy0 + pos / kLcTW, ox = k.ox0 + pos % kLcTW;
            pix_off[nb] = oy < a.H && ox < a.W ? (uint32_t)(oy * a.W + ox) * 4u : kEpiOob;
            py[nb] = oy;
            px[nb] = ox;
        }
        EpilogueArgs e;
        e.out = epi_out_base(a.out, (int64_t)k.n * a.out_sn, a.out_nhwc);
        e.out2 = a.out2 ? a.out2 + (int64_t)k.n * a.Cout * P : nullptr;
        e.add = nullptr; e.aux1 = nullptr; e.aux2 = nullptr;
        e.Cout = a.Cout; e.P = P; e.act = 0;
        e.add_mode = 0; e.Hout = a.H; e.Wout = a.W; e.out_nhwc = a.out_nhwc;
        conv_epilogue<MBO, NB>(e, acc, 0, q, pix_off, py, px);
    };
    // One step = B of a finished chunk and A of the next one.  The two waves that share a SIMD (w and w + 4) take the halves in
    // opposite order, so that the bf16 matrix work of one runs under the vector work of the other (they overlap across waves,
    // tools/ubench/mfma_bf16_rate.hip); in lock step both would be in the same half at the same time.
    const bool b_first = wave < kLcWaves / 2;

    // Tile order: workgroup b runs on XCD b % 8 (round-robin dispatch); the tiles of one XCD are a contiguous run of the tile list
    // (whole image bands), so that tiles sharing halo rows / patch cache lines meet in the same 4 MB L2
    int w, wstep, wend;
    if (a.banded) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        w = (int)((int64_t)a.total * xcd / 8) + slot;
        wend = (int)((int64_t)a.total * (xcd + 1) / 8);
        wstep = gridDim.x >> 3;
    } else {
        w = blockIdx.x;
        wend = a.total;
        wstep = gridDim.x;
    }
    if (w >= wend) return;
    Work cur = decode(w);
#pragma unroll
    for (int i = 0; i < kParts; ++i) fetch_part(cur, i);
    // split weights of the 3x3 layer -> LDS, once per workgroup (all loads in flight, then the stores; behind the first tile's fetch)
    {
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(a.w_out);
        constexpr int total = MBO * 3 * kLcWChunk / 16;
        constexpr int kBatch = (total + kLcThreads - 1) / kLcThreads;
        u32x4 t[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int pc = tid + i * kLcThreads;
            const int grp = pc >> 5, within = pc & 31;               // 32 pieces per (mb, chunk, tap, plane)
            const int mb = grp / 81, r0 = grp - mb * 81;
            const int c = r0 / 27, r1 = r0 - c * 27;
            const int tap = r1 / 3, pl = r1 - tap * 3;
            if (pc < total) t[i] = src[((int64_t)((tap * 3 + c) * 3 + pl) * a.CoutPad + mb * 16) * 2 + within];
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int pc = tid + i * kLcThreads;
            if (pc < total) reinterpret_cast<u32x4*>(Wt)[pc] = t[i];
        }
    }
    stage(cur);
    __syncthreads();
    setup_a(cur);
    int wn = w + wstep;
    bool more = wn < wend;          // a tile follows `cur`: its loads are issued during cur's steps
    Work nxt = cur;
    if (more) {
        nxt = decode(wn);
        fetch_part(nxt, 0);
        fetch_part(nxt, 1);
    }
    int buf = 0;                    // chunk buffer the NEXT A phase writes (alternates per chunk, across tiles)
    phase_a(0, buf);
    __syncthreads();
    while (true) {
#pragma unroll
        for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = bo[mb];
        // chunk 0 | chunk 1
        if (more) {
            fetch_part(nxt, 2);
            fetch_part(nxt, 3);
        }
        if (b_first) { phase_b(0, buf); phase_a(1, buf ^ 1); } else { phase_a(1, buf ^ 1); phase_b(0, buf); }
        __syncthreads();
        if (more) {
            fetch_part(nxt, 4);
            fetch_part(nxt, 5);
        }
        if (b_first) { phase_b(1, buf ^ 1); phase_a(2, buf); } else { phase_a(2, buf); phase_b(1, buf ^ 1); }
        __syncthreads();            // every wave is done with the patch and the fine tile of this tile
        const bool had_more = more;
        const Work done = cur;
        if (had_more) {
            stage(nxt);             // (waits for the loads issued over the last three steps)
            __syncthreads();
            cur = nxt;
            w = wn;
            wn = w + wstep;
            setup_a(cur);
            more = wn < wend;
            if (more) {
                nxt = decode(wn);
                fetch_part(nxt, 0);
                fetch_part(nxt, 1);
            }
        }
        // the last chunk of this tile | chunk 0 of the next one
        if (b_first || !had_more) {
            phase_b(2, buf);
            store(done);
            if (had_more) phase_a(0, buf ^ 1);
        } else {
            phase_a(0, buf ^ 1);
            phase_b(2, buf);
            store(done);
        }
        if (!had_more) break;
        buf ^= 1;
        __syncthreads();
    }
}

template <int KS, int MBO>
static int launch_lat_conv(LatConvArgs& a, hipStream_t stream) {
    constexpr int lds = 2 * kLcChunkB + MBO * 3 * kLcWChunk + kLcPPX * kLcPS * 4 + 4 * KS * kLcMH * kLcFW * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = lat_conv_kernel<KS, MBO>;
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    if (!attr_ok) return ITERMVS_ERR_LAUNCH;
    const int cus = itermvs_num_cus();
    const int grid = a.total < cus ? a.total : cus;
    a.banded = grid % 8 == 0 && a.total >= grid ? 1 : 0;          // (grid < 8 or ragged: plain order)
    itermvs_profile_begin(3, stream);           // bench.py's convolution roofline brackets this launch like an itermvs_conv2d one
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kLcThreads), lds, stream, a);
    itermvs_profile_end(3, stream);
    return itermvs_launch_status();
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_lateral_conv3x3(const float* fine, int64_t fine_sn, int32_t Cf, const float* coarse, int64_t coarse_sn,
                                       int32_t N, int32_t H, int32_t W, const float* w_lat, const float* b_lat, const void* w_out,
                                       const float* b_out, int32_t Cout, void* out, int64_t out_sn, int32_t out_layout, float* out2,
                                       void* stream) {
    ITERMVS_RETURN_IF(!fine || !coarse || !w_lat || !w_out || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || H > 8190 || W > 8190, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((int64_t)48 * H * W * 4 >= ((int64_t)1 << 31), ITERMVS_ERR_DIMS);        // 32-bit byte offsets inside one image
    ITERMVS_RETURN_IF(Cf != 16 || Cout != 16, ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(out_layout < 0 || out_layout > 3, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(((uintptr_t)w_out) % 16, ITERMVS_ERR_ALIGN);
    ITERMVS_RETURN_IF(out_layout == 1 && (((uintptr_t)out) % 16 || out_sn % 4), ITERMVS_ERR_ALIGN);
    ITERMVS_RETURN_IF(out_layout >= 2 && (((uintptr_t)out) % 8 || out_sn % 4), ITERMVS_ERR_ALIGN);
    LatConvArgs a;
    a.fine = fine; a.coarse = coarse; a.w_lat = w_lat; a.b_lat = b_lat; a.w_out = w_out; a.b_out = b_out;
    a.out = reinterpret_cast<float*>(out); a.out2 = out2;
    a.fine_sn = fine_sn; a.coarse_sn = coarse_sn; a.out_sn = out_sn;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.CoutPad = (Cout + 15) / 16 * 16; a.out_nhwc = out_layout;
    a.tiles_x = (W + kLcTW - 1) / kLcTW; a.tiles_y = (H + kLcTH - 1) / kLcTH;
    const int64_t total = (int64_t)N * a.tiles_x * a.tiles_y;
    ITERMVS_RETURN_IF(total >= ((int64_t)1 << 31), ITERMVS_ERR_DIMS);
    a.total = (int)total;
    return launch_lat_conv<4, 1>(a, (hipStream_t)stream);
}
