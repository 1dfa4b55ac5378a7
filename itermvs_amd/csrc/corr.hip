// Fused homography warp + bilinear gather + group-wise correlation (+ view-weighted mean).
// Replaces models/module.py:68-125 + models/itermvs.py:48-69 / :86-120 without ever writing a
// warped volume or a per-view cost volume (iteration branch) to memory.
//
// Work decomposition (wave64, channels-last source maps):
//   a work ITEM is (pixel, hypothesis, channel chunk); the 4 (C=16) or 8 (C=32,48) lanes that
//   share one (pixel, hypothesis) read one CONTIGUOUS 64/128/192-byte feature vector per
//   bilinear tap (dwordx4 / 3x dwordx2 per lane), so a wave touches 8-16 cache lines per load
//   instruction instead of 64.  A chunk always covers whole correlation groups (G=8):
//     C=16 -> 4 lanes x float4, two groups of 2 per lane
//     C=32 -> 8 lanes x float4, one group of 4 per lane
//     C=48 -> 8 lanes x 6 floats, one group of 6 per lane
//   so no cross-lane reduction is needed.  The loop over source views runs inside the lane and
//   carries the view-weighted accumulators in registers.
//   Results are transposed through LDS so the [B,N,8,H,W] planes CorrNet / PixelViewWeight
//   consume are written as full rows of TILE pixels.
#include "corr_common.hpp"

namespace itermvs {

// Pixel tile of the two fused correlation kernels: 16 x 2 pixels.  Measured on MI355X at cfg 1 (profiles/r03, tools/
// kernel_bench.py; -DITERMVS_CORR_TW=<8|16|32> builds the other shapes): 8 x 4 / 16 x 2 / 32 x 1 tiles run the iteration
// kernel in 24.3 / 24.1 / 23.9 us on a noise depth map and 21.9 / 21.5 / 21.7 us on a smooth one, the initialisation
// kernel in 24.6 / 23.4 / 25.3 us -- the tile shape is NOT what bounds them (vector-L1 hits were already 85 %).
#ifndef ITERMVS_CORR_TW
#define ITERMVS_CORR_TW 16
#endif
constexpr int kIterTW = ITERMVS_CORR_TW;      // a power of two, >= 4, dividing 32

// ---------------------------------------------------------------------------------------------
// iteration branch
// ---------------------------------------------------------------------------------------------
template <int CPG, int TILE, int FT>
__device__ __forceinline__ void corr_iter_level(const IterArgs& a, const IterLevel& L, int lvl, float* __restrict__ lds) {
    using K = Chunk<CPG>;
    constexpr int LS = TILE + 1;  // padded LDS row: the transposed writes hit distinct banks
    const int N = L.N;
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    // pixel tile = kIterTW x (TILE / kIterTW) pixels: a 2-D patch of the sample grid maps to a compact patch of every
    // source map, so the rows its bilinear taps touch are shared by the vertically adjacent pixels of the SAME workgroup
    // (vector-L1 hits) instead of being fetched again by whichever workgroup owns the next row strip
    constexpr int TW = kIterTW, TH = TILE / kIterTW;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tile = a.band > 0 ? xcd_tile_chunked(a.band) : xcd_tile(tiles_x * tiles_y);
    if (tile >= tiles_x * tiles_y) return;          // padding blocks of the XCD-aligned grid (uniform per block)
    const int tile_ty = tile / tiles_x, tile_tx = tile - tile_ty * tiles_x;
    const int x0 = tile_tx * TW, y0 = tile_ty * TH;
    constexpr int LPT = K::LPT, NGL = K::NG;      // lanes per (pixel, hypothesis), groups finalised per lane
    const int per_px = N * LPT;
    const int items = TILE * per_px;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = a.proj + ((size_t)(lvl * a.B + b) * a.S) * 12;
    const uint32_t sy = (uint32_t)L.sy * feat_bytes<FT>(), sx = (uint32_t)L.sx * feat_bytes<FT>();   // byte strides (chunk_corr)

    // index arithmetic without integer division on the common path (an unsigned division costs ~25 vector instructions,
    // and these gather kernels are bound by instruction issue): items per pixel is a power of two for the reference's
    // hypothesis counts, the tile width is a compile-time power of two
    const int px_shift = (per_px & (per_px - 1)) == 0 ? 31 - __clz(per_px) : -1;
#pragma unroll 1
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int px = px_shift >= 0 ? item >> px_shift : item / per_px;
        const int rem = item - px * per_px;
        const int n = rem / LPT;
        const int j = rem - n * LPT;         // the LPT lanes of one (pixel, hypothesis) are adjacent lanes
        const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
        if (x >= a.W || y >= a.H) continue;  // whole lane groups drop out together
        const int p = y * a.W + x;

        float d;
        if (L.depth) {
            d = L.depth[((size_t)b * N + n) * P + p];
        } else {  // itermvs.py:291-293
            d = iter_hypothesis(a.nd[b * a.nd_sb + p], L.offs[n], inv_min, inv_max);
        }
        float refv[K::VEC];
        if constexpr (FT == ITERMVS_F32) load_vec<K::VEC>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff + j * 4, refv);
        else load_ref16<CPG>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff, j, refv);      // 16-byte lanes (corr_common.hpp)

        const float xs = (float)x * g.xr, ys = (float)y * g.yr;
        float acc[NGL];
#pragma unroll
        for (int q = 0; q < NGL; ++q) acc[q] = 0.0f;
        float wsum = 1e-5f;  // itermvs.py:88
        const uint32_t joff = (uint32_t)(j * 4) * feat_bytes<FT>();
        // The projection, the bilinear footprint and the view weight of (pixel, hypothesis) in view s are the same for the
        // four chunk lanes: lane j evaluates them for view s0 + j, then the quad walks the batch of views and every lane
        // takes view s0 + k's from lane k with DPP quad_perm moves.
        for (int s0 = 0; s0 < a.S; s0 += LPT) {
            Footprint mine = {0u, 0u, 0u, 0u, 0.0f, 0.0f, 0.0f, 0.0f};
            float w_mine = 0.0f;
            if (s0 + j < a.S) {
                const float* m = proj + (s0 + j) * 12;
                float rx, ry, rz, ix, iy;
                ray_dir(m, xs, ys, rx, ry, rz);
                project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
                mine = make_footprint(ix, iy, L.W1, L.H1, sy, sx);
                // planar [B,S,H,W]: a scattered dword per lane; interleaved [B,H,W,S] (the engine's layout): the quad's four
                // views are one 16-byte run and the wave's pixels one 64-byte run
                w_mine = a.view_w[(int64_t)b * a.vw_sb + (int64_t)(s0 + j) * a.vw_ss + (int64_t)p * a.vw_sp];
            }
            const int nb = min(LPT, a.S - s0);        // wave-uniform
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                if (k < nb) {
                    const Footprint tp = quad_footprint(mine, k);
                    const float wv = quad_bcast(w_mine, k);
                    float corr[NGL];
                    if constexpr (FT == ITERMVS_F32) chunk_corr<CPG, FT>(feat_base<FT>(L.src[s0 + k], (int64_t)b * L.sb), joff, tp, refv, corr);
                    else chunk_corr16<CPG, FT>(feat_base<FT>(L.src[s0 + k], (int64_t)b * L.sb), j, tp, refv, corr);
#pragma unroll
                    for (int q = 0; q < NGL; ++q) acc[q] = acc[q] + corr[q] * wv;  // itermvs.py:115
                    wsum = wsum + wv;                                                // itermvs.py:116
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NGL; ++q)
            lds[(n * ITERMVS_GROUPS + (FT == ITERMVS_F32 ? K::group(j, q) : group16<CPG>(j, q))) * LS + px] = acc[q] / wsum;
    }
    __syncthreads();
    const int rows = N * ITERMVS_GROUPS;
    // 16-byte stores where the rows allow it (a wave-level store costs the CU about the same whatever its width): W a
    // multiple of 4 keeps every quad of a tile row inside one plane row and aligned
    if ((a.W & 3) == 0 && ((uintptr_t)L.out & 15) == 0) {
        for (int idx = threadIdx.x; idx < rows * (TILE / 4); idx += kThreads) {
            const int row = idx / (TILE / 4), px = (idx - row * (TILE / 4)) * 4;
            const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
            if (x < a.W && y < a.H) {
                const float* __restrict__ l = lds + row * LS + px;
                *reinterpret_cast<float4*>(L.out + ((size_t)b * rows + row) * P + (size_t)y * a.W + x) = make_float4(l[0], l[1], l[2], l[3]);
            }
        }
        return;
    }
    for (int idx = threadIdx.x; idx < rows * TILE; idx += kThreads) {
        const int row = idx / TILE, px = idx - row * TILE;
        const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
        if (x < a.W && y < a.H) L.out[((size_t)b * rows + row) * P + (size_t)y * a.W + x] = lds[row * LS + px];
    }
}

template <int TILE, int FT>
__global__ void __launch_bounds__(kThreads) corr_iter_kernel(const IterArgs a) {
    __shared__ float lds[ITERMVS_MAX_HYP * ITERMVS_GROUPS * (TILE + 1)];
    const int lvl = blockIdx.y;      // (dispatching the heaviest level first measured slower: profiles/r05/r05m_corr_iter_level_order.txt)
    const IterLevel& L = a.lv[lvl];
    switch (L.C) {
        case 16: corr_iter_level<2, TILE, FT>(a, L, lvl, lds); break;
        case 32: corr_iter_level<4, TILE, FT>(a, L, lvl, lds); break;
        default: corr_iter_level<6, TILE, FT>(a, L, lvl, lds); break;
    }
}

#ifdef ITERMVS_ITER_TWO_PHASE       // A/B builds only: measured and not shipped (experiments/corr_iter_two_phase.inc)
#include "experiments/corr_iter_two_phase.inc"
#endif

// ---------------------------------------------------------------------------------------------
// initialisation branch: per-view correlation volume for PixelViewWeight (itermvs.py:48-53)
// grid = (pixel tiles, S * hypothesis blocks, B)
// ---------------------------------------------------------------------------------------------
struct InitArgs {
    const float* src[ITERMVS_MAX_SRC];
    int64_t sb, sy, sx;
    itermvs_fmap ref;
    const float* proj;
    const float* depth;
    const float* inv_min;
    const float* inv_max;
    float* out;
    int B, S, H, W, N, C, H1, W1, NB;  // NB = hypotheses per block
    int out_cl;                        // out as [B,S,N,H,W,8] (groups last) instead of [B,S,N,8,H,W]
};

constexpr int kInitNB = 8;  // hypotheses per block
// LDS row pitch = TILE + kInitPad.  A store instruction of the gather loop writes, per half wave, rows {32 grp + 2j (+q)} x 4
// consecutive pixels (lane = pixel * 8 + grp * 4 + j): with a pitch of 41 dwords the row offsets 82 j + 1312 grp fall on banks
// 18 j + 32 grp (mod 64) -- eight disjoint runs of four banks.  (Pitch 33 put 2j + px of one grp on the same bank for up to
// four lanes: 60 % of this kernel's LDS cycles were conflict cycles, profiles/r04_pmc_kernels.json.  The kernel's time did
// not change -- its LDS pipeline is busy 1.3 us of 24: profiles/r05/r05ad_corr_init_lds_pitch.txt.)
#ifndef ITERMVS_INIT_PAD
#define ITERMVS_INIT_PAD 9
#endif
constexpr int kInitPad = ITERMVS_INIT_PAD;

template <int CPG, int TILE, int FT>
__device__ __forceinline__ void corr_init_body(const InitArgs& a, float* __restrict__ lds) {
    using K = Chunk<CPG>;
    constexpr int LS = TILE + kInitPad;
    const int nblocks = (a.N + a.NB - 1) / a.NB;
    const int s = blockIdx.y / nblocks;
    const int n0 = (blockIdx.y - s * nblocks) * a.NB;
    const int nb = min(a.NB, a.N - n0);
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    constexpr int TW = kIterTW, TH = TILE / kIterTW;       // 2-D pixel tile (see corr_iter_level)
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tile = xcd_tile(tiles_x * tiles_y);
    if (tile >= tiles_x * tiles_y) return;
    const int tile_ty = tile / tiles_x, tile_tx = tile - tile_ty * tiles_x;
    const int x0 = tile_tx * TW, y0 = tile_ty * TH;
    constexpr int LPT = K::LPT, NGL = K::NG;
    const int ngrp = (nb + LPT - 1) / LPT;   // groups of LPT hypotheses
    const int per_px = ngrp * LPT;
    const int items = TILE * per_px;
    const WarpGeom g = make_geom(a.W, a.H, a.W1, a.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* m = a.proj + ((size_t)b * a.S + s) * 12;
    const float* fsrc = feat_base<FT>(a.src[s], (int64_t)b * a.sb);
    const uint32_t sy = (uint32_t)a.sy * feat_bytes<FT>(), sx = (uint32_t)a.sx * feat_bytes<FT>();   // byte strides (chunk_corr)

    const int px_shift = (per_px & (per_px - 1)) == 0 ? 31 - __clz(per_px) : -1;      // (no integer division on the common path)
#pragma unroll 1
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int px = px_shift >= 0 ? item >> px_shift : item / per_px;
        const int rem = item - px * per_px;
        const int grp = rem / LPT;
        const int j = rem - grp * LPT;
        const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
        if (x >= a.W || y >= a.H) continue;
        const int p = y * a.W + x;
        float refv[K::VEC];
        if constexpr (FT == ITERMVS_F32) {
            if (a.ref.sc == 1) {       // channels-last reference (the engine's layout): the chunk is VEC/4 vector loads off one address
                load_feat<K::VEC, FT>(feat_base<FT>((const float*)a.ref.data, (int64_t)b * a.ref.sb),
                                      (uint32_t)(y * (int)a.ref.sy + x * (int)a.ref.sx) + (uint32_t)(j * 4), refv);
            } else {
#pragma unroll
                for (int c = 0; c < K::VEC; ++c)
                    refv[c] = ld_feat<FT>((const float*)a.ref.data, b * a.ref.sb + chunk_channel<K::VEC>(j, c) * a.ref.sc + y * a.ref.sy + x * a.ref.sx);
            }
        } else {                       // 16-byte lanes (corr_common.hpp): another channel -> lane assignment
            const char* rb = reinterpret_cast<const char*>(feat_base<FT>((const float*)a.ref.data, (int64_t)b * a.ref.sb));
            const uint32_t ro = 2u * (uint32_t)(y * (int)a.ref.sy + x * (int)a.ref.sx);
            const bool vec = a.ref.sc == 1 && !(a.ref.sx & 7) && !(a.ref.sy & 7) && !(a.ref.sb & 7);
            if (vec) {
                load_ref16_stored<CPG, FT>(rb, ro, j, refv);
            } else {
#pragma unroll
                for (int c = 0; c < K::VEC; ++c)
                    refv[c] = ld_feat<FT>((const float*)a.ref.data, b * a.ref.sb + chunk16_channel<CPG>(j, c) * a.ref.sc + y * a.ref.sy + x * a.ref.sx);
            }
        }
        // lane j projects hypothesis grp*LPT + j once; the quad then walks its LPT hypotheses and
        // every lane takes the footprint of hypothesis k from lane k (DPP quad_perm moves)
        const int nl_mine = grp * LPT + j;
        Footprint mine = {0u, 0u, 0u, 0u, 0.0f, 0.0f, 0.0f, 0.0f};
        if (nl_mine < nb) {
            const int n = n0 + nl_mine;
            float d;
            if (a.depth) {
                d = a.depth[((size_t)b * a.N + n) * P + p];
            } else {  // itermvs.py:13-17
                d = init_hypothesis(n, a.N, inv_min, inv_max);
            }
            float rx, ry, rz, ix, iy;
            ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
            project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
            mine = make_footprint(ix, iy, a.W1, a.H1, sy, sx);
        }
        const uint32_t joff = (uint32_t)(j * 4) * feat_bytes<FT>();
        const int cnt = min(LPT, nb - grp * LPT);      // uniform per lane group; whole groups take the branch together
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            if (k < cnt) {
                const Footprint tp = quad_footprint(mine, k);
                float corr[NGL];
                if constexpr (FT == ITERMVS_F32) chunk_corr<CPG, FT>(fsrc, joff, tp, refv, corr);
                else chunk_corr16<CPG, FT>(fsrc, j, tp, refv, corr);
                const int nl = grp * LPT + k;
#pragma unroll
                for (int q = 0; q < NGL; ++q)
                    lds[(nl * ITERMVS_GROUPS + (FT == ITERMVS_F32 ? K::group(j, q) : group16<CPG>(j, q))) * LS + px] = corr[q];
            }
        }
    }
    __syncthreads();
    const int rows = nb * ITERMVS_GROUPS;
    float* o = a.out + (((size_t)b * a.S + s) * a.N + n0) * ITERMVS_GROUPS * P;
    if (a.out_cl) {
        // groups-last output [B,S,N,H,W,8] (what itermvs_conv2d's in_layout 1 stages with two 16-byte loads per pixel): item =
        // (hypothesis, pixel, half of the 8 groups) = one 16-byte store; a tile row of 16 pixels is one 512-byte run
        for (int idx = threadIdx.x; idx < nb * TILE * 2; idx += kThreads) {
            const int n = idx / (TILE * 2), r = idx - n * (TILE * 2);
            const int px = r >> 1, hf = r & 1;
            const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
            if (x < a.W && y < a.H) {
                const float* __restrict__ l = lds + (n * ITERMVS_GROUPS + hf * 4) * LS + px;
                *reinterpret_cast<float4*>(o + ((size_t)n * P + (size_t)y * a.W + x) * ITERMVS_GROUPS + hf * 4) = make_float4(l[0], l[LS], l[2 * LS], l[3 * LS]);
            }
        }
        return;
    }
    if ((a.W & 3) == 0 && ((uintptr_t)a.out & 15) == 0) {      // 16-byte stores (see corr_iter_level)
        for (int idx = threadIdx.x; idx < rows * (TILE / 4); idx += kThreads) {
            const int row = idx / (TILE / 4), px = (idx - row * (TILE / 4)) * 4;
            const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
            if (x < a.W && y < a.H) {
                const float* __restrict__ l = lds + row * LS + px;
                *reinterpret_cast<float4*>(o + (size_t)row * P + (size_t)y * a.W + x) = make_float4(l[0], l[1], l[2], l[3]);
            }
        }
        return;
    }
    for (int idx = threadIdx.x; idx < rows * TILE; idx += kThreads) {
        const int row = idx / TILE, px = idx - row * TILE;
        const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
        if (x < a.W && y < a.H) o[(size_t)row * P + (size_t)y * a.W + x] = lds[row * LS + px];
    }
}


#ifdef ITERMVS_INIT_SWEEP_BUILD     // A/B builds only: measured and not shipped (experiments/corr_init_plane_sweep.inc)
#include "experiments/corr_init_plane_sweep.inc"
#endif


template <int TILE, int FT>
__global__ void __launch_bounds__(kThreads) corr_init_kernel(const InitArgs a) {
    __shared__ float lds[kInitNB * ITERMVS_GROUPS * (TILE + kInitPad)];
    switch (a.C) {
        case 16: corr_init_body<2, TILE, FT>(a, lds); break;
        case 32: corr_init_body<4, TILE, FT>(a, lds); break;
        default: corr_init_body<6, TILE, FT>(a, lds); break;
    }
}

// out[b,n,g,p] = sum_s corr[b,s,n,g,p]*w[b,s,p] / (1e-5 + sum_s w[b,s,p])     (itermvs.py:59-69)
// Blocks [0, n_agg) aggregate; the blocks after them (itermvs_view_aggregate_up) up-sample the view weights x2 for the
// iterations (itermvs.py:56-57,71): both only read `w`, one launch instead of two.
__global__ void view_aggregate_kernel(const float* __restrict__ corr, const float* __restrict__ w, int S, int B, int NG,
                                      int P, float* __restrict__ out, int n_agg, int H3, int W3, float* __restrict__ w_up, int vec4,
                                      int up_interleaved, int corr_cl) {
    if ((int)blockIdx.x >= n_agg) {
        bilinear_up_body(w, B * S, H3, W3, 2, 0, w_up, (int64_t)(blockIdx.x - n_agg) * blockDim.x + threadIdx.x, up_interleaved ? S : 0);
        return;
    }
    const int64_t per = (int64_t)NG * P;
    if (corr_cl) {
        // corr stored groups last, [B,S,N,P,8] (itermvs_corr_init's out_layout 1): a thread owns one (hypothesis, pixel) and its 8
        // groups -- two 16-byte loads per view; out stays [B,N,8,P] (lanes = consecutive pixels: every store instruction one run).
        // The same per-element arithmetic as the forms below.
        const int N = NG / ITERMVS_GROUPS;
        const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= (int64_t)B * N * P) return;
        const int p = (int)(t % P);
        const int bn = (int)(t / P), b = bn / N, n = bn - b * N;
        float acc[ITERMVS_GROUPS], wsum = 1e-5f;
#pragma unroll
        for (int g = 0; g < ITERMVS_GROUPS; ++g) acc[g] = 0.0f;
        for (int s = 0; s < S; ++s) {
            const float ws = w[((size_t)b * S + s) * P + p];
            const float4* __restrict__ c = reinterpret_cast<const float4*>(corr + ((((size_t)b * S + s) * N + n) * P + p) * ITERMVS_GROUPS);
            const float4 c0 = c[0], c1 = c[1];
            acc[0] = acc[0] + c0.x * ws; acc[1] = acc[1] + c0.y * ws; acc[2] = acc[2] + c0.z * ws; acc[3] = acc[3] + c0.w * ws;
            acc[4] = acc[4] + c1.x * ws; acc[5] = acc[5] + c1.y * ws; acc[6] = acc[6] + c1.z * ws; acc[7] = acc[7] + c1.w * ws;
            wsum = wsum + ws;
        }
#pragma unroll
        for (int g = 0; g < ITERMVS_GROUPS; ++g) out[((size_t)b * NG + (size_t)n * ITERMVS_GROUPS + g) * P + p] = acc[g] / wsum;
        return;
    }
    if (vec4) {                    // four consecutive pixels per thread: 16-byte loads and stores (same arithmetic per element)
        const int64_t t4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int64_t t = t4 * 4;
        if (t >= (int64_t)B * per) return;
        const int p = (int)(t % P);
        const int b = (int)(t / per);
        const int64_t r = t - (int64_t)b * per;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wsum[4] = {1e-5f, 1e-5f, 1e-5f, 1e-5f};
        for (int s = 0; s < S; ++s) {
            const float4 ws = *reinterpret_cast<const float4*>(w + ((size_t)b * S + s) * P + p);
            const float4 c = *reinterpret_cast<const float4*>(corr + ((size_t)b * S + s) * per + r);
            acc[0] = acc[0] + c.x * ws.x; acc[1] = acc[1] + c.y * ws.y; acc[2] = acc[2] + c.z * ws.z; acc[3] = acc[3] + c.w * ws.w;
            wsum[0] = wsum[0] + ws.x; wsum[1] = wsum[1] + ws.y; wsum[2] = wsum[2] + ws.z; wsum[3] = wsum[3] + ws.w;
        }
        *reinterpret_cast<float4*>(out + t) = make_float4(acc[0] / wsum[0], acc[1] / wsum[1], acc[2] / wsum[2], acc[3] / wsum[3]);
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * per) return;
    const int p = (int)(t % P);
    const int b = (int)(t / per);
    const int64_t r = t - (int64_t)b * per;
    float acc = 0.0f, wsum = 1e-5f;
    for (int s = 0; s < S; ++s) {
        const float ws = w[((size_t)b * S + s) * P + p];
        acc = acc + corr[((size_t)b * S + s) * per + r] * ws;
        wsum = wsum + ws;
    }
    out[t] = acc / wsum;
}

// out[m,p] = max_n softmax_n(x[m,n,p])                                        (itermvs.py:347-348)
// N <= 32 (the reference's 32 initial hypotheses): all logits of a pixel are fetched in one round of loads and reduced in
// registers; the generic form below walks the hypotheses twice.
template <int NMAX>
__global__ void softmax_max_small_kernel(const float* __restrict__ x, int M, int N, int P, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * P) return;
    const int p = (int)(t % P);
    const int m = (int)(t / P);
    const float* xp = x + (size_t)m * N * P + p;
    float v[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) v[n] = n < N ? xp[(size_t)n * P] : -INFINITY;
    float mx = v[0];
#pragma unroll
    for (int n = 1; n < NMAX; ++n) mx = fmaxf(mx, v[n]);
    float sum = 0.0f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) sum += expf(v[n] - mx);      // exp(-inf) = 0 for the padding
    out[t] = 1.0f / sum;       // the largest probability belongs to the largest logit: exp(0) / sum
}

__global__ void softmax_max_kernel(const float* __restrict__ x, int M, int N, int P, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * P) return;
    const int p = (int)(t % P);
    const int m = (int)(t / P);
    const float* xp = x + (size_t)m * N * P + p;
    float mx = -INFINITY;
    for (int n = 0; n < N; ++n) mx = fmaxf(mx, xp[(size_t)n * P]);
    float sum = 0.0f;
    for (int n = 0; n < N; ++n) sum += expf(xp[(size_t)n * P] - mx);
    // the largest probability belongs to the largest logit: exp(0)/sum
    out[t] = 1.0f / sum;
}

// PixelViewWeight tail (itermvs.py:343-348): 1x1 convolution C -> 1 (+bias), softmax over the N hypotheses, max.
// x [M*N, C, P] planes (output of the 3x3 layer).  A block owns 64 pixels of one m; wave g evaluates the logits
// of hypotheses g*N/8 .. (g+1)*N/8 - 1 (64 coalesced plane loads per lane for N = 32, C = 16), the eight waves
// combine max and sum through LDS.  Replaces a 1x1 convolution launch + softmax_max_kernel (two passes of
// strided loads by only M*P threads).
template <int C, int NPW>
__global__ void __launch_bounds__(512) pvw_tail_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int N, int P,
                                                       float* __restrict__ out) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int m = blockIdx.y;
    const int p = blockIdx.x * 64 + lane;
    const bool live = p < P;
    const float b0 = bias ? bias[0] : 0.0f;
    float wt[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wt[c] = w[c];
    float v[NPW];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int n = g * NPW + i;
        // hypotheses past N (N not a multiple of 8 * NPW) re-read the last plane; their logit is masked below
        const float* xp = x + ((size_t)(m * N + min(n, N - 1)) * C) * P + (live ? p : 0);
        float acc = b0;
#pragma unroll
        for (int c = 0; c < C; ++c) acc = fmaf(xp[(size_t)c * P], wt[c], acc);
        v[i] = n < N ? acc : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    red[g][lane] = mx;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) mx = fmaxf(mx, red[k][lane]);
    __syncthreads();
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NPW; ++i) sum += expf(v[i] - mx);
    red[g][lane] = sum;
    __syncthreads();
    if (g == 0 && live) {
        float tot = red[0][lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += red[k][lane];
        out[(size_t)m * P + p] = 1.0f / tot;      // the largest probability belongs to the largest logit: exp(0) / sum
    }
}


// ---------------------------------------------------------------------------------------------
// itermvs_tap_indices: the sampling decisions of the fused kernels, observable.  One thread per (b, view, hypothesis,
// pixel) evaluates the hypothesis, ray, projection and footprint with the SAME inline functions, in the same order, as
// corr_iter_level / corr_init_body / the gradient kernels (iter_hypothesis / init_hypothesis, ray_dir, project_fast,
// make_taps) and stores floor(ix), floor(iy) and the validity bits instead of gathering.
// ---------------------------------------------------------------------------------------------
struct TapArgs {
    const float* proj;
    const float* depth;
    const float* nd;
    int64_t nd_sb;
    float offs[ITERMVS_MAX_HYP];
    const float* inv_min;
    const float* inv_max;
    int32_t* out;
    float* coords;
    int B, S, H, W, N, H1, W1, init;
};

__device__ __forceinline__ int32_t tap_floor_to_int(float f) {
    if (!(f == f)) return INT32_MIN;                       // NaN
    if (f >= 1073741824.0f) return 1073741824;             // +-2^30 saturation (inf included)
    if (f <= -1073741824.0f) return -1073741824;
    return (int32_t)f;
}

__global__ void __launch_bounds__(256) tap_indices_kernel(const TapArgs a) {
    const int P = a.H * a.W;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)a.B * a.S * a.N * P) return;
    const int p = (int)(t % P);
    const int n = (int)((t / P) % a.N);
    const int s = (int)((t / ((int64_t)P * a.N)) % a.S);
    const int b = (int)(t / ((int64_t)P * a.N * a.S));
    const int y = p / a.W, x = p - y * a.W;
    const WarpGeom g = make_geom(a.W, a.H, a.W1, a.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    float d;
    if (a.depth) d = a.depth[((size_t)b * a.N + n) * P + p];
    else if (a.init) d = init_hypothesis(n, a.N, inv_min, inv_max);
    else d = iter_hypothesis(a.nd[b * a.nd_sb + p], a.offs[n], inv_min, inv_max);
    const float* m = a.proj + ((size_t)b * a.S + s) * 12;
    const float xs = (float)x * g.xr, ys = (float)y * g.yr;
    float rx, ry, rz, ix, iy;
    ray_dir(m, xs, ys, rx, ry, rz);
    project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
    TapDiag dg;
    (void)make_taps(ix, iy, a.W1, a.H1, &dg);
    int32_t* o = a.out + (((size_t)b * a.S + s) * a.N + n) * 3 * P + p;
    o[0] = tap_floor_to_int(dg.fx0);
    o[P] = tap_floor_to_int(dg.fy0);
    o[2 * (size_t)P] = dg.bits;
    if (a.coords) {
        float* c = a.coords + (((size_t)b * a.S + s) * a.N + n) * 2 * P + p;
        c[0] = ix;
        c[P] = iy;
    }
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_tap_indices(const itermvs_tap_params* p, void* stream) {
    ITERMVS_RETURN_IF(!p, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1 || p->H1 < 1 || p->W1 < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->proj || !p->inv_depth_min || !p->inv_depth_max || !p->out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->N < 1 || (p->init && p->N < 2), ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(!p->depth && !p->init && (!p->norm_depth || p->N > ITERMVS_MAX_HYP), p->norm_depth ? ITERMVS_ERR_DIMS : ITERMVS_ERR_NULL);
    TapArgs a;
    a.proj = p->proj; a.depth = p->depth; a.nd = p->norm_depth; a.nd_sb = p->norm_depth_sb;
    for (int n = 0; n < ITERMVS_MAX_HYP; ++n) a.offs[n] = p->offsets[n];
    a.inv_min = p->inv_depth_min; a.inv_max = p->inv_depth_max; a.out = p->out; a.coords = p->coords;
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.N = p->N; a.H1 = p->H1; a.W1 = p->W1; a.init = p->init;
    const int64_t total = (int64_t)p->B * p->S * p->N * p->H * p->W;
    ITERMVS_RETURN_IF(total >= ((int64_t)1 << 31) * 256, ITERMVS_ERR_DIMS);
    hipLaunchKernelGGL(tap_indices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return itermvs_launch_status();
}

extern "C" int itermvs_pvw_tail(const float* x, const float* w, const float* bias, int32_t M, int32_t N, int32_t C,
                                int32_t P, float* out, void* stream) {
    ITERMVS_RETURN_IF(!x || !w || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(M < 1 || P < 1 || N < 1 || N > 32, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(C != 16, ITERMVS_ERR_CHANNELS);
    const dim3 grid((P + 63) / 64, M);
    if (N <= 8) hipLaunchKernelGGL((pvw_tail_kernel<16, 1>), grid, dim3(512), 0, (hipStream_t)stream, x, w, bias, N, P, out);
    else if (N <= 16) hipLaunchKernelGGL((pvw_tail_kernel<16, 2>), grid, dim3(512), 0, (hipStream_t)stream, x, w, bias, N, P, out);
    else hipLaunchKernelGGL((pvw_tail_kernel<16, 4>), grid, dim3(512), 0, (hipStream_t)stream, x, w, bias, N, P, out);
    return itermvs_launch_status();
}

// Tile order of the iteration kernel (see xcd_tile_chunked): tiles per XCD band, 0 = one contiguous band per XCD.
// A TUNING build takes ITERMVS_ITER_BAND_ROWS (tile rows per band; 0 = contiguous, -1 = plain interleaving) from the environment.
static int iter_band_tiles(int tiles_x, int tiles, const itermvs_corr_iter_params* p) {
    if (const char* e = itermvs_tuning_env("ITERMVS_ITER_BAND_ROWS")) {
        const int rows = atoi(e);
        return rows < 0 ? 1 : rows * tiles_x;
    }
    (void)tiles; (void)p;
    return 0;
}

extern "C" int itermvs_corr_iter(const itermvs_corr_iter_params* p, void* stream) {
    ITERMVS_RETURN_IF(!p, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->ref_q || !p->proj || !p->view_w || !p->inv_depth_min || !p->inv_depth_max, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(((uintptr_t)p->ref_q) % 16, ITERMVS_ERR_ALIGN);
    for (int l = 0; l < 3; ++l) {
        const int rc = itermvs_check_level(p->src[l], p->S);
        if (rc) return rc;
        ITERMVS_RETURN_IF(p->N[l] < 1 || p->N[l] > ITERMVS_MAX_HYP, ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(!p->out[l], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(!p->depth[l] && !p->norm_depth, ITERMVS_ERR_NULL);
    }
    IterArgs a;
    int coff = 0;
    for (int l = 0; l < 3; ++l) {
        IterLevel& L = a.lv[l];
        for (int v = 0; v < ITERMVS_MAX_SRC; ++v) L.src[v] = (const float*)p->src[l].view[v < p->S ? v : 0];
        L.sb = p->src[l].sb; L.sy = p->src[l].sy; L.sx = p->src[l].sx;
        L.depth = p->depth[l];
        L.out = p->out[l];
        for (int n = 0; n < ITERMVS_MAX_HYP; ++n) L.offs[n] = p->offsets[l][n];
        L.C = p->src[l].C; L.H1 = p->src[l].H; L.W1 = p->src[l].W; L.N = p->N[l];
        L.coff = coff;
        coff += L.C;
    }
    a.ref_q = p->ref_q; a.proj = p->proj; a.view_w = p->view_w; a.nd = p->norm_depth; a.nd_sb = p->norm_depth_sb;
    a.inv_min = p->inv_depth_min; a.inv_max = p->inv_depth_max;
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.CQ = coff;
    a.band = 0;
    if (p->view_w_sb == 0 && p->view_w_ss == 0 && p->view_w_sp == 0) {      // default: planar [B,S,H,W]
        a.vw_sp = 1; a.vw_ss = (int64_t)p->H * p->W; a.vw_sb = a.vw_ss * p->S;
    } else {
        ITERMVS_RETURN_IF(p->view_w_ss < 1 || p->view_w_sp < 1 || p->view_w_sb < 0, ITERMVS_ERR_LAYOUT);
        a.vw_sb = p->view_w_sb; a.vw_ss = p->view_w_ss; a.vw_sp = p->view_w_sp;
    }
    // One form: source views walked inside the lane.  (A views-across-waves form issued 14 % fewer vector instructions but
    // missed the vector L1 31 % more often -- 33.4 vs 28.9 us, profiles/r02 -- and was removed; `impl` is reserved.)
    ITERMVS_RETURN_IF(p->impl != 0, ITERMVS_ERR_DIMS);
    const int dtype = p->src[0].dtype;
    ITERMVS_RETURN_IF(p->src[1].dtype != dtype || p->src[2].dtype != dtype, ITERMVS_ERR_DTYPE);
    itermvs_profile_begin(1, (hipStream_t)stream);
    {
        constexpr int TILE = 32;
        const int tiles_x = (p->W + kIterTW - 1) / kIterTW;
        const int tiles = tiles_x * ((p->H + TILE / kIterTW - 1) / (TILE / kIterTW));
        a.band = iter_band_tiles(tiles_x, tiles, p);
        const dim3 grid(a.band > 0 ? xcd_chunked_grid(tiles, a.band) : (unsigned)(((tiles + 7) / 8) * 8), 3, p->B);
        switch (dtype) {
            case ITERMVS_F16: hipLaunchKernelGGL((corr_iter_kernel<TILE, ITERMVS_F16>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
            case ITERMVS_BF16: hipLaunchKernelGGL((corr_iter_kernel<TILE, ITERMVS_BF16>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
#ifdef ITERMVS_ITER_TWO_PHASE
            default: {
                int nmax = 1;
                for (int l = 0; l < 3; ++l) nmax = p->N[l] > nmax ? p->N[l] : nmax;
                const size_t shm = (size_t)nmax * (ITERMVS_GROUPS * (TILE + 1) + TILE * 4 + TILE * 4 * 9) * 4;
                hipLaunchKernelGGL((corr_iter2_kernel<TILE, ITERMVS_F32>), grid, dim3(kThreads), shm, (hipStream_t)stream, a, nmax);
                break;
            }
#else
            default: hipLaunchKernelGGL((corr_iter_kernel<TILE, ITERMVS_F32>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
#endif
        }
    }
    itermvs_profile_end(1, (hipStream_t)stream);
    return itermvs_launch_status();
}

extern "C" int itermvs_corr_init(const itermvs_corr_init_params* p, void* stream) {
    ITERMVS_RETURN_IF(!p, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1 || p->N < 2, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->ref.data || !p->proj || !p->inv_depth_min || !p->inv_depth_max || !p->out, ITERMVS_ERR_NULL);
    const int rc = itermvs_check_level(p->src, p->S);
    if (rc) return rc;
    ITERMVS_RETURN_IF(p->ref.C != p->src.C || p->ref.H != p->H || p->ref.W != p->W, ITERMVS_ERR_DIMS);
    InitArgs a;
    for (int v = 0; v < ITERMVS_MAX_SRC; ++v) a.src[v] = (const float*)p->src.view[v < p->S ? v : 0];
    a.sb = p->src.sb; a.sy = p->src.sy; a.sx = p->src.sx;
    a.ref = p->ref; a.proj = p->proj; a.depth = p->depth;
    a.inv_min = p->inv_depth_min; a.inv_max = p->inv_depth_max; a.out = p->out;
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.N = p->N;
    a.C = p->src.C; a.H1 = p->src.H; a.W1 = p->src.W; a.NB = kInitNB;
    ITERMVS_RETURN_IF(p->out_layout != 0 && p->out_layout != 1, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(p->out_layout == 1 && ((uintptr_t)p->out) % 16, ITERMVS_ERR_ALIGN);
    a.out_cl = p->out_layout;
    constexpr int TILE = 32;
    const int nblocks = (p->N + kInitNB - 1) / kInitNB;
    itermvs_profile_begin(2, (hipStream_t)stream);
    ITERMVS_RETURN_IF(p->ref.dtype != p->src.dtype, ITERMVS_ERR_DTYPE);
#ifdef ITERMVS_INIT_SWEEP_BUILD
    // the plane sweep through LDS (A/B builds only, see above): 48 fp32 channels; explicit per-pixel hypotheses are accepted (boxes
    // that do not fit take the memory taps), coordinates must fit the packed 16-bit box reduction
    static const bool sweep_on = [] { const char* e = getenv("ITERMVS_INIT_SWEEP"); return !e || e[0] != '0'; }();
    if (sweep_on && a.C == 48 && p->src.dtype == ITERMVS_F32 && p->src.H < 65535 && p->src.W < 65535) {
        const int tiles_x = (p->W + 7) / 8, tiles_y = (p->H + 7) / 8;
        // planes per wave: four waves of a workgroup take consecutive plane blocks of one (tile, view); as many workgroups per
        // (tile, view) as it takes to give every SIMD about two waves
        const int64_t tv = (int64_t)tiles_x * tiles_y * p->S * p->B;
        // (two waves per SIMD are resident: LDS.  More waves than that run in rounds -- give a wave twice the planes instead,
        //  as long as every SIMD still gets a wave)
        int ppw = 1;
        while (ppw < 8 && tv * ((p->N + 4 * ppw - 1) / (4 * ppw)) * 4 > 2 * 4 * (int64_t)itermvs_num_cus() &&
               tv * ((p->N + 8 * ppw - 1) / (8 * ppw)) * 4 >= 4 * (int64_t)itermvs_num_cus()) ppw *= 2;
        const int groups = (p->N + 4 * ppw - 1) / (4 * ppw);
        const dim3 grid(((tiles_x * tiles_y + 7) / 8) * 8, p->S * groups, p->B);
        hipLaunchKernelGGL(corr_init_sweep_kernel, grid, dim3(256), 4 * kSweepWaveLds, (hipStream_t)stream, a, ppw, tiles_x, tiles_y);
        itermvs_profile_end(2, (hipStream_t)stream);
        return itermvs_launch_status();
    }
#endif
    const int tiles = ((p->W + kIterTW - 1) / kIterTW) * ((p->H + TILE / kIterTW - 1) / (TILE / kIterTW));
    const dim3 grid(((tiles + 7) / 8) * 8, p->S * nblocks, p->B);
    switch (p->src.dtype) {
        case ITERMVS_F16: hipLaunchKernelGGL((corr_init_kernel<TILE, ITERMVS_F16>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
        case ITERMVS_BF16: hipLaunchKernelGGL((corr_init_kernel<TILE, ITERMVS_BF16>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL((corr_init_kernel<TILE, ITERMVS_F32>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
    }
    itermvs_profile_end(2, (hipStream_t)stream);
    return itermvs_launch_status();
}

extern "C" int itermvs_view_aggregate(const float* corr, const float* w, int32_t S, int32_t B, int32_t N, int32_t P,
                                      float* out, void* stream) {
    ITERMVS_RETURN_IF(!corr || !w || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(S < 1 || B < 1 || N < 1 || P < 1, ITERMVS_ERR_DIMS);
    const int vec4 = (P & 3) == 0 && ((((uintptr_t)corr) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0;
    const int64_t total = ((int64_t)B * N * ITERMVS_GROUPS * P) / (vec4 ? 4 : 1);
    const int na = (int)((total + 255) / 256);
    hipLaunchKernelGGL(view_aggregate_kernel, dim3((unsigned)na), dim3(256), 0, (hipStream_t)stream,
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out, na, 0, 0, (float*)nullptr, vec4, 0, 0);
    return itermvs_launch_status();
}

extern "C" int itermvs_view_aggregate_up(const float* corr, int32_t corr_layout, const float* w, int32_t S, int32_t B, int32_t N, int32_t H3,
                                         int32_t W3, float* out, float* w_up, int32_t w_up_interleaved, void* stream) {
    ITERMVS_RETURN_IF(!corr || !w || !out || !w_up, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(S < 1 || B < 1 || N < 1 || H3 < 1 || W3 < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(corr_layout != 0 && corr_layout != 1, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(corr_layout == 1 && ((uintptr_t)corr) % 16, ITERMVS_ERR_ALIGN);
    const int P = H3 * W3;
    const int vec4 = (P & 3) == 0 && ((((uintptr_t)corr) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0;
    const int na = corr_layout == 1 ? (int)(((int64_t)B * N * P + 255) / 256)
                                    : (int)(((int64_t)B * N * ITERMVS_GROUPS * P / (vec4 ? 4 : 1) + 255) / 256);
    const int nu = (int)(((int64_t)B * S * P * 4 + 255) / 256);
    hipLaunchKernelGGL(view_aggregate_kernel, dim3((unsigned)(na + nu)), dim3(256), 0, (hipStream_t)stream,
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out, na, H3, W3, w_up, vec4, w_up_interleaved ? 1 : 0, corr_layout);
    return itermvs_launch_status();
}

extern "C" int itermvs_softmax_max(const float* x, int32_t M, int32_t N, int32_t P, float* out, void* stream) {
    ITERMVS_RETURN_IF(!x || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(M < 1 || N < 1 || P < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)M * P;
    if (N <= 32)    // 64-thread blocks: the M * P pixels spread over as many CUs as possible
        hipLaunchKernelGGL(softmax_max_small_kernel<32>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x,
                           M, N, P, out);
    else
        hipLaunchKernelGGL(softmax_max_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M,
                           N, P, out);
    return itermvs_launch_status();
}
