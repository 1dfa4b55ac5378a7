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
    const int tile = xcd_tile(tiles_x * tiles_y);
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
#ifdef ITERMVS_ITER_ORDER       // A/B builds: dispatch order of the levels (heaviest first shortens the tail of the 1 920-workgroup launch?)
    const int lvl = (ITERMVS_ITER_ORDER >> (4 * blockIdx.y)) & 3;
#else
    const int lvl = blockIdx.y;
#endif
    const IterLevel& L = a.lv[lvl];
    switch (L.C) {
        case 16: corr_iter_level<2, TILE, FT>(a, L, lvl, lds); break;
        case 32: corr_iter_level<4, TILE, FT>(a, L, lvl, lds); break;
        default: corr_iter_level<6, TILE, FT>(a, L, lvl, lds); break;
    }
}

#ifdef ITERMVS_ITER_TWO_PHASE
// ---------------------------------------------------------------------------------------------
// MEASURED AND NOT SHIPPED (round 5, profiles/r05/r05t_corr_iter_two_phase.txt): 27.9 / 25.8 us against 24.4 / 21.9 us (noise / smooth
// depth) at cfg 1, 632 / 463 against 589 / 345 us at the cfg-5 shape.  The barrier-separated phases and the LDS round trip of the
// footprints cost more than the leaner gather loop wins (108 VGPRs: still 4 waves per SIMD).
// iteration branch, two-phase form (A/B builds only: -DITERMVS_ITER_TWO_PHASE).  Phase A: one thread per (pixel, hypothesis,
// view) of a batch of four views computes the projection, the footprint and the view weight and parks them in LDS (36 bytes);
// phase B: the quads gather -- footprints from LDS (quad-uniform addresses: broadcast reads), no projection code and no DPP
// traffic in the loop, fewer live registers.  Same arithmetic, same order of the view sum: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <int CPG, int TILE, int FT>
__device__ __forceinline__ void corr_iter2_level(const IterArgs& a, const IterLevel& L, int lvl, float* __restrict__ lds, uint32_t* __restrict__ fpl,
                                                 float* __restrict__ wl) {
    using K = Chunk<CPG>;
    constexpr int LS = TILE + 1;
    const int N = L.N;
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    constexpr int TW = kIterTW, TH = TILE / kIterTW;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tile = xcd_tile(tiles_x * tiles_y);
    if (tile >= tiles_x * tiles_y) return;
    const int tile_ty = tile / tiles_x, tile_tx = tile - tile_ty * tiles_x;
    const int x0 = tile_tx * TW, y0 = tile_ty * TH;
    constexpr int LPT = K::LPT, NGL = K::NG;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = a.proj + ((size_t)(lvl * a.B + b) * a.S) * 12;
    const uint32_t sy = (uint32_t)L.sy * feat_bytes<FT>(), sx = (uint32_t)L.sx * feat_bytes<FT>();
    const int per_px = N * LPT;                 // gather items per pixel
    const int items = TILE * per_px;
    for (int s0 = 0; s0 < a.S; s0 += 4) {
        const int nb = min(4, a.S - s0);
        // ---- phase A: footprints of (pixel, hypothesis, view s0 + v) -> LDS [(px * N + n) * 4 + v][9]
        const int fitems = TILE * N * 4;
#pragma unroll 1
        for (int it = threadIdx.x; it < fitems; it += kThreads) {
            const int v = it & 3, pn = it >> 2;
            const int n = pn % N, px = pn / N;
            const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
            Footprint f = {0u, 0u, 0u, 0u, 0.0f, 0.0f, 0.0f, 0.0f};
            float wv = 0.0f;
            if (v < nb && x < a.W && y < a.H) {
                const int p = y * a.W + x;
                const float d = L.depth ? L.depth[((size_t)b * N + n) * P + p] : iter_hypothesis(a.nd[b * a.nd_sb + p], L.offs[n], inv_min, inv_max);
                const float* m = proj + (s0 + v) * 12;
                float rx, ry, rz, ix, iy;
                ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
                project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
                f = make_footprint(ix, iy, L.W1, L.H1, sy, sx);
                wv = a.view_w[(int64_t)b * a.vw_sb + (int64_t)(s0 + v) * a.vw_ss + (int64_t)p * a.vw_sp];
            }
            uint32_t* o = fpl + it * 9;
            o[0] = f.r0; o[1] = f.r1; o[2] = f.c0; o[3] = f.c1;
            o[4] = __float_as_uint(f.nw); o[5] = __float_as_uint(f.ne); o[6] = __float_as_uint(f.sw); o[7] = __float_as_uint(f.se);
            o[8] = __float_as_uint(wv);
        }
        __syncthreads();
        // ---- phase B: gather; the running sums of an item live in LDS between the batches of views (same thread, same slots)
#pragma unroll 1
        for (int item = threadIdx.x; item < items; item += kThreads) {
            const int px = item / per_px;
            const int rem = item - px * per_px;
            const int n = rem / LPT, j = rem - n * LPT;
            const int x = x0 + (px & (TW - 1)), y = y0 + px / TW;
            if (x >= a.W || y >= a.H) continue;
            const int p = y * a.W + x;
            float refv[K::VEC];
            if constexpr (FT == ITERMVS_F32) load_vec<K::VEC>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff + j * 4, refv);
            else load_ref16<CPG>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff, j, refv);
            const uint32_t joff = (uint32_t)(j * 4) * feat_bytes<FT>();
            const uint32_t* fb = fpl + (size_t)(px * N + n) * 4 * 9;
            float acc[NGL], wsum = 1e-5f;
            int slot[NGL];
#pragma unroll
            for (int q = 0; q < NGL; ++q) {
                slot[q] = (n * ITERMVS_GROUPS + (FT == ITERMVS_F32 ? K::group(j, q) : group16<CPG>(j, q))) * LS + px;
                acc[q] = s0 == 0 ? 0.0f : lds[slot[q]];
            }
            if (s0 != 0) wsum = wl[item];
#pragma unroll 1
            for (int k = 0; k < nb; ++k) {
                const uint32_t* o = fb + k * 9;
                Footprint tp;
                tp.r0 = o[0]; tp.r1 = o[1]; tp.c0 = o[2]; tp.c1 = o[3];
                tp.nw = __uint_as_float(o[4]); tp.ne = __uint_as_float(o[5]); tp.sw = __uint_as_float(o[6]); tp.se = __uint_as_float(o[7]);
                const float wv = __uint_as_float(o[8]);
                float corr[NGL];
                if constexpr (FT == ITERMVS_F32) chunk_corr<CPG, FT>(feat_base<FT>(L.src[s0 + k], (int64_t)b * L.sb), joff, tp, refv, corr);
                else chunk_corr16<CPG, FT>(feat_base<FT>(L.src[s0 + k], (int64_t)b * L.sb), j, tp, refv, corr);
#pragma unroll
                for (int q = 0; q < NGL; ++q) acc[q] = acc[q] + corr[q] * wv;
                wsum = wsum + wv;
            }
            const bool last = s0 + 4 >= a.S;
#pragma unroll
            for (int q = 0; q < NGL; ++q) lds[slot[q]] = last ? acc[q] / wsum : acc[q];
            if (!last) wl[item] = wsum;
        }
        __syncthreads();
    }
    const int rows = N * ITERMVS_GROUPS;
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
__global__ void __launch_bounds__(kThreads) corr_iter2_kernel(const IterArgs a, int nmax) {
    extern __shared__ __attribute__((aligned(16))) float dyn2[];
    float* lds = dyn2;                                                            // [nmax * 8][TILE + 1]
    float* wl = lds + nmax * ITERMVS_GROUPS * (TILE + 1);                        // [TILE * nmax * 4]
    uint32_t* fpl = reinterpret_cast<uint32_t*>(wl + TILE * nmax * 4);            // [TILE * nmax * 4][9]
    const int lvl = blockIdx.y;
    const IterLevel& L = a.lv[lvl];
    switch (L.C) {
        case 16: corr_iter2_level<2, TILE, FT>(a, L, lvl, lds, fpl, wl); break;
        case 32: corr_iter2_level<4, TILE, FT>(a, L, lvl, lds, fpl, wl); break;
        default: corr_iter2_level<6, TILE, FT>(a, L, lvl, lds, fpl, wl); break;
    }
}
#endif  // ITERMVS_ITER_TWO_PHASE

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
};

template <int CPG, int TILE, int FT>
__device__ __forceinline__ void corr_init_body(const InitArgs& a, float* __restrict__ lds) {
    using K = Chunk<CPG>;
    constexpr int LS = TILE + 1;
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


#ifdef ITERMVS_INIT_SWEEP_BUILD
// ---------------------------------------------------------------------------------------------
// MEASURED AND NOT SHIPPED (round 5; compiled only with -DITERMVS_INIT_SWEEP_BUILD for A/B runs, tools/build_variants.sh
// CORR_VARIANTS="sweep:-DITERMVS_INIT_SWEEP_BUILD"; profiles/r05/r05l_corr_init_plane_sweep.txt): bit-identical results, but
// 43.8 us against the gather form's 23.5 us at cfg 1 and 454 against 347 us at the cfg-5 shape.  The patch per wave (19 KB) and
// 226 VGPRs allow two waves per SIMD, and a wave walks its planes as one dependent chain (project -> box reduction -> offsets ->
// loads -> LDS -> taps): knock-outs put 20 us of the 44 in that skeleton alone (no loads, no LDS traffic, no tap arithmetic), 14 us
// in the LDS taps + blend, 5 us in the LDS stores, 4 us in the box reduction, 0 in the staging loads (hidden).  The gather form
// hides its L1 latency with 8 waves per SIMD; this form cannot.
//
// initialisation branch as a PLANE SWEEP through LDS (C = 48, fp32 storage).
// The generated hypotheses are planes (itermvs.py:11-19): all pixels of a tile share the depth, so the bilinear footprints of
// an 8 x 8 pixel tile on one (view, plane) fall into ONE compact patch of the source map -- 81 pixels (median; 100 at the 99th
// percentile) against 256 taps, at every resolution (tools/plane_patch_count.py, profiles/r05/r05_plane_patch_count.txt).
// The gather form above fetches every tap through the vector L1 (503 MB per launch at cfg 1: the kernel sat at 60 % of the L1's
// rate and 0.13 of the HBM roof).  Here a WAVE owns (tile, view, a few planes), lane = pixel, and per plane
//   1. every lane projects its pixel with the same device functions as before (project_fast, make_taps: identical indices and
//      weights), the wave reduces the bounding box of the valid taps (packed 16-bit min / max, 6 butterfly steps);
//   2. the box is copied to the wave's own LDS region with coalesced 16-byte loads (lanes walk the 12 four-channel pieces of a
//      pixel, then the pixels: 64-byte runs), stored [piece][pixel] with an odd 16-byte plane stride so that the stores (8
//      lanes = 8 pieces of a pixel) and the reads (16 lanes = neighbouring pixels of one piece) hit distinct banks;
//      the NEXT plane's loads are issued before the current plane's arithmetic (registers), no workgroup barrier anywhere;
//   3. the four taps of the 12 pieces come from LDS (ds_read_b128 with the piece in the immediate offset); blend, products with the
//      lane's 48 reference values (registers, loaded once per wave) and the 8 group means follow the gather form's
//      arithmetic and association exactly (chunk_corr<6>) -- results are bit-identical;
//   4. a lane stores its 8 results: 32-byte runs per plane row, no transposing LDS pass.
// A box that exceeds the LDS patch (grazing planes, explicit per-pixel hypotheses) takes the same arithmetic with the taps
// loaded from memory.
// ---------------------------------------------------------------------------------------------
constexpr int kSweepCap = 100;                        // pixels of a staged box
constexpr int kSweepPl = kSweepCap * 16 + 16;         // bytes between the [piece] planes: an odd number of 16-byte slots
constexpr int kSweepWaveLds = 12 * kSweepPl;          // 19 392 B per wave: two workgroups of four waves per CU
using sweep_f4 = __attribute__((ext_vector_type(4))) float;      // (native vector: HIP's float4 class kept the staging array in scratch)
constexpr int kSweepLoads = (kSweepCap * 6 + 63) / 64;    // 16-byte pieces per lane of HALF a full box (6 of the 12 pieces per pixel)

struct SweepTaps {
    uint32_t o00, o01, o10, o11;      // byte offsets of the four taps (LDS: inside piece 0 of the box; memory: inside the map)
    float nw, ne, sw, se;
};

// reduction over the wave of two packed unsigned 16-bit values per lane (min of both halves)
__device__ __forceinline__ uint32_t wave_pk_min_u16(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        const uint32_t lo = min(v & 0xffffu, o & 0xffffu), hi = min(v >> 16, o >> 16);
        v = lo | (hi << 16);
    }
    return v;
}

// channel-pair products of pieces 6 * HALF .. 6 * HALF + 5 (24 channels): chunk_corr's per-channel fma chain over the four taps,
// then its pair products
template <bool FROM_LDS, int HALF>
__device__ __forceinline__ void sweep_pairs(const char* __restrict__ base, const SweepTaps& t, const float (&refv)[48], float (&pair)[24]) {
#pragma unroll
    for (int cb = 6 * HALF; cb < 6 * HALF + 6; ++cb) {
        constexpr int kStep = FROM_LDS ? kSweepPl : 16;
        const float4 v00 = *reinterpret_cast<const float4*>(base + t.o00 + cb * kStep);
        const float4 v01 = *reinterpret_cast<const float4*>(base + t.o01 + cb * kStep);
        const float4 v10 = *reinterpret_cast<const float4*>(base + t.o10 + cb * kStep);
        const float4 v11 = *reinterpret_cast<const float4*>(base + t.o11 + cb * kStep);
        const float w0 = fmaf(t.se, v11.x, fmaf(t.sw, v10.x, fmaf(t.ne, v01.x, t.nw * v00.x)));
        const float w1 = fmaf(t.se, v11.y, fmaf(t.sw, v10.y, fmaf(t.ne, v01.y, t.nw * v00.y)));
        const float w2 = fmaf(t.se, v11.z, fmaf(t.sw, v10.z, fmaf(t.ne, v01.z, t.nw * v00.z)));
        const float w3 = fmaf(t.se, v11.w, fmaf(t.sw, v10.w, fmaf(t.ne, v01.w, t.nw * v00.w)));
        const float* r = refv + 4 * cb;
        pair[2 * cb] = fmaf(w1, r[1], w0 * r[0]);
        pair[2 * cb + 1] = fmaf(w3, r[3], w2 * r[2]);
        // two pieces' taps in flight at most: hoisting all reads ahead of the arithmetic costs a hundred registers
        if (cb & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

struct SweepBox {
    int bx, by, bw, npx;      // origin, width, pixels (0: nothing to stage); wave-uniform
    bool fits;
};

// HALF a box -> registers -> LDS: piece i = lane + 64 k of a half is (pixel i / 6, four-channel piece 6 h + i % 6); the lanes walk
// the six pieces of a pixel, then the pixels: 96-byte runs in memory, distinct banks in LDS (odd plane stride).
// The split of i is the lane's own constant (SweepLane, once per wave); per plane only pixel -> (row, column) of the box remains,
// by a float reciprocal ((px + 0.5) / bw is at least 0.5 / bw away from an integer: exact for these sizes) and 24-bit
// multiplies -- 32-bit integer multiplies run at a quarter of the rate and, six per piece, cost this kernel more than its
// arithmetic.
struct SweepLane {
    uint32_t px[kSweepLoads];        // pixel index of piece k inside a box
    uint32_t lds[kSweepLoads];       // LDS byte offset of piece k of half 0 (+ 6 * kSweepPl for half 1)
    uint32_t cb16[kSweepLoads];      // 16 * (piece of the pixel, 0..5): its byte offset inside the pixel's vector
};
__device__ __forceinline__ void sweep_lane_init(SweepLane& L, int lane) {
#pragma unroll
    for (int k = 0; k < kSweepLoads; ++k) {
        const uint32_t i = (uint32_t)lane + 64u * k;
        const uint32_t px = __umulhi(i, 715827883u);              // i / 6 for i < 2^16  (2^32 / 6 + 1)
        L.px[k] = px;
        L.cb16[k] = (i - px * 6u) * 16u;
        L.lds[k] = (i - px * 6u) * (uint32_t)kSweepPl + px * 16u;
    }
}
// memory byte offsets of piece k of half 0 (half 1: + 96 bytes) for this plane's box
__device__ __forceinline__ void sweep_offsets(uint32_t (&goff)[kSweepLoads], const SweepLane& L, const SweepBox& box, uint32_t sy, uint32_t sx) {
    const float inv_bw = 1.0f / (float)box.bw;
    const uint32_t last = (uint32_t)max(box.npx, 1) - 1u;
    const uint32_t base = __umul24((uint32_t)box.by, sy) + __umul24((uint32_t)box.bx, sx);
#pragma unroll
    for (int k = 0; k < kSweepLoads; ++k) {
        // (lanes past the end re-read the last pixel: an unconditional load keeps the staging registers out of control flow)
        const uint32_t px = min(L.px[k], last);
        const uint32_t r = (uint32_t)(((float)px + 0.5f) * inv_bw);
        const uint32_t c = px - __umul24(r, (uint32_t)box.bw);
        goff[k] = base + __umul24(r, sy) + __umul24(c, sx) + L.cb16[k];
    }
}
#ifndef ITERMVS_SWEEP_KO          // knock-out builds (timing only): bit 0 no staging loads, 1 no LDS stores, 2 no tap arithmetic, 3 no box reduction
#define ITERMVS_SWEEP_KO 0
#endif
__device__ __forceinline__ void sweep_fetch_half(sweep_f4 (&stage)[kSweepLoads], const uint32_t (&goff)[kSweepLoads], const SweepBox& box, int h,
                                                 const char* __restrict__ fsrc) {
    if (!box.fits || box.npx == 0) return;
#if ITERMVS_SWEEP_KO & 1
#pragma unroll
    for (int k = 0; k < kSweepLoads; ++k) stage[k] = sweep_f4{(float)goff[k], 1.0f, 2.0f, (float)h};
    return;
#endif
#pragma unroll
    for (int k = 0; k < kSweepLoads; ++k) stage[k] = *reinterpret_cast<const sweep_f4*>(fsrc + goff[k] + 96u * (uint32_t)h);
}
__device__ __forceinline__ void sweep_store_half(const sweep_f4 (&stage)[kSweepLoads], const SweepLane& L, const SweepBox& box, int h,
                                                 char* __restrict__ patch) {
    if (!box.fits || box.npx == 0) return;
    const uint32_t npx = (uint32_t)box.npx;
#if ITERMVS_SWEEP_KO & 2
#pragma unroll
    for (int k = 0; k < kSweepLoads; ++k) asm volatile("" ::"v"(stage[k]));
    return;
#endif
#pragma unroll
    for (int k = 0; k < kSweepLoads; ++k)
        if (L.px[k] < npx) *reinterpret_cast<sweep_f4*>(patch + L.lds[k] + (uint32_t)(6 * kSweepPl) * (uint32_t)h) = stage[k];
}

__global__ void __launch_bounds__(256) corr_init_sweep_kernel(const InitArgs a, int planes_per_wave, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char sweep_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* __restrict__ patch = sweep_lds + wave * kSweepWaveLds;
    const int tile = xcd_tile(tiles_x * tiles_y);
    if (tile >= tiles_x * tiles_y) return;
    const int b = blockIdx.z;
    const int groups = (a.N + 4 * planes_per_wave - 1) / (4 * planes_per_wave);     // workgroups per (tile, view)
    const int s = blockIdx.y / groups;
    const int n_begin = ((blockIdx.y - s * groups) * 4 + wave) * planes_per_wave;
    const int n_end = min(a.N, n_begin + planes_per_wave);
    if (n_begin >= n_end) return;                       // (no barriers in this kernel: a wave may leave)
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    const bool live = x < a.W && y < a.H;
    const int P = a.H * a.W;
    const int p = live ? y * a.W + x : 0;
    const WarpGeom g = make_geom(a.W, a.H, a.W1, a.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* m = a.proj + ((size_t)b * a.S + s) * 12;
    const char* __restrict__ fsrc = reinterpret_cast<const char*>(a.src[s] + (int64_t)b * a.sb);
    const uint32_t sy = (uint32_t)a.sy * 4u, sx = (uint32_t)a.sx * 4u;      // byte strides of the source map

    float refv[48];
    {
        const float* rp = (const float*)a.ref.data + (int64_t)b * a.ref.sb + (live ? y * a.ref.sy + x * a.ref.sx : 0);
        if (a.ref.sc == 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(rp + 4 * i);
                refv[4 * i] = t.x; refv[4 * i + 1] = t.y; refv[4 * i + 2] = t.z; refv[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 48; ++c) refv[c] = rp[c * a.ref.sc];
        }
    }
    float rx, ry, rz;
    ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);

    // the sampling taps of this lane's pixel on plane n (LDS offsets once the box is known) + the wave's box of valid taps
    auto project_plane = [&](int n, SweepTaps& st, SweepBox& box) __attribute__((always_inline)) {
        const float d = a.depth ? a.depth[((size_t)b * a.N + n) * P + p] : init_hypothesis(n, a.N, inv_min, inv_max);
        float ix, iy;
        project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
        TapDiag dg;
        const Taps t = make_taps(ix, iy, a.W1, a.H1, &dg);
        const int vx = dg.bits & 3, vy = (dg.bits >> 2) & 3;
        const bool any = live && vx && vy;                    // no valid column or row: all four weights are zero
        // columns / rows this lane needs: the valid ones (make_taps clamps an invalid index to 0)
        const int lx0 = (vx & 1) ? t.x0 : t.x1, lx1 = (vx & 2) ? t.x1 : t.x0;
        const int ly0 = (vy & 1) ? t.y0 : t.y1, ly1 = (vy & 2) ? t.y1 : t.y0;
        // packed unsigned 16-bit: (min x | 0xffff - max x), (min y | 0xffff - max y); idle lanes are neutral
        uint32_t kx = 0xffffffffu, ky = 0xffffffffu;
        if (any) {
            kx = (uint32_t)lx0 | ((0xffffu - (uint32_t)lx1) << 16);
            ky = (uint32_t)ly0 | ((0xffffu - (uint32_t)ly1) << 16);
        }
#if ITERMVS_SWEEP_KO & 8
        kx = __builtin_amdgcn_readfirstlane(kx) & 0xfff0fff0u; ky = __builtin_amdgcn_readfirstlane(ky) & 0xfff0fff0u;      // (a box of 16 x 16: does not fit -> only with bit 0..2 builds)
        kx = (kx & 0xffffu) | ((0xffffu - ((kx & 0xffffu) + 9u)) << 16); ky = (ky & 0xffffu) | ((0xffffu - ((ky & 0xffffu) + 9u)) << 16);
#else
        kx = __builtin_amdgcn_readfirstlane(wave_pk_min_u16(kx));
        ky = __builtin_amdgcn_readfirstlane(wave_pk_min_u16(ky));
#endif
        st.nw = any ? t.nw : 0.0f; st.ne = any ? t.ne : 0.0f; st.sw = any ? t.sw : 0.0f; st.se = any ? t.se : 0.0f;
        if (kx == 0xffffffffu) {
            box.bx = box.by = 0; box.bw = 1; box.npx = 0; box.fits = true;
            st.o00 = st.o01 = st.o10 = st.o11 = 0u;
            return;
        }
        box.bx = (int)(kx & 0xffffu);
        box.by = (int)(ky & 0xffffu);
        box.bw = (int)(0xffffu - (kx >> 16)) - box.bx + 1;
        const int bh = (int)(0xffffu - (ky >> 16)) - box.by + 1;
        box.npx = box.bw * bh;
        box.fits = box.npx <= kSweepCap;
        if (box.fits) {
            // an invalid column / row reads its valid partner (weight 0); a lane without any valid tap reads the box origin
            const int cx0 = any ? lx0 : box.bx, cx1 = any ? lx1 : box.bx, cy0 = any ? ly0 : box.by, cy1 = any ? ly1 : box.by;
            const uint32_t r0 = __umul24((uint32_t)(cy0 - box.by), (uint32_t)box.bw), r1 = __umul24((uint32_t)(cy1 - box.by), (uint32_t)box.bw);
            const uint32_t c0 = (uint32_t)(cx0 - box.bx), c1 = (uint32_t)(cx1 - box.bx);
            st.o00 = (r0 + c0) * 16u; st.o01 = (r0 + c1) * 16u; st.o10 = (r1 + c0) * 16u; st.o11 = (r1 + c1) * 16u;
        } else {
            const uint32_t r0 = (uint32_t)t.y0 * sy, r1 = (uint32_t)t.y1 * sy, c0 = (uint32_t)t.x0 * sx, c1 = (uint32_t)t.x1 * sx;
            st.o00 = r0 + c0; st.o01 = r0 + c1; st.o10 = r1 + c0; st.o11 = r1 + c1;
        }
    };
    sweep_f4 stage[kSweepLoads];
    uint32_t goff[kSweepLoads];
    SweepLane L;
    sweep_lane_init(L, lane);
    // Software pipeline over the wave's planes, ONE LDS image: while plane n's pieces 0..5 are multiplied, the next plane's
    // pieces 0..5 are in flight to registers; they are stored once plane n has read its own (the LDS executes a wave's
    // operations in order), then the same registers take pieces 6..11 during the second half of the arithmetic.
    SweepTaps cur, nxt;
    SweepBox box, nbox;
    project_plane(n_begin, cur, box);
    sweep_offsets(goff, L, box, sy, sx);
    sweep_fetch_half(stage, goff, box, 0, fsrc);
    sweep_store_half(stage, L, box, 0, patch);
    sweep_fetch_half(stage, goff, box, 1, fsrc);
    sweep_store_half(stage, L, box, 1, patch);
    for (int n = n_begin; n < n_end; ++n) {
        const bool more = n + 1 < n_end;                // wave-uniform
        nbox.npx = 0; nbox.fits = true; nbox.bx = nbox.by = 0; nbox.bw = 1;
        if (more) {
            project_plane(n + 1, nxt, nbox);
            sweep_offsets(goff, L, nbox, sy, sx);
            sweep_fetch_half(stage, goff, nbox, 0, fsrc);
        }
        float pair[24];
#if ITERMVS_SWEEP_KO & 4
#pragma unroll
        for (int q = 0; q < 24; ++q) pair[q] = refv[q] * cur.nw + __uint_as_float(cur.o00);
#define SWEEP_PAIRS(H)
#else
#define SWEEP_PAIRS(H)                                                   \
        if (box.npx != 0) {                                              \
            if (box.fits) sweep_pairs<true, H>(patch, cur, refv, pair);  \
            else sweep_pairs<false, H>(fsrc, cur, refv, pair);           \
        }
#endif
        SWEEP_PAIRS(0)
        if (more) {
            sweep_store_half(stage, L, nbox, 0, patch);
            sweep_fetch_half(stage, goff, nbox, 1, fsrc);
        }
        SWEEP_PAIRS(1)
#undef SWEEP_PAIRS
        if (more) sweep_store_half(stage, L, nbox, 1, patch);
        if (live) {
            float* o = a.out + ((((size_t)b * a.S + s) * a.N + n) * ITERMVS_GROUPS) * P + p;
            // group q = channel pairs 3q, 3q+1, 3q+2; even groups (A + B) + C, odd groups A + (B + C)  (chunk_corr<6>'s association)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float A = pair[3 * q], B = pair[3 * q + 1], Cc = pair[3 * q + 2];
                const float sum = (q & 1) ? A + (B + Cc) : (A + B) + Cc;
                o[(size_t)q * P] = box.npx != 0 ? div_rcp(sum, 6.0f, 1.0f / 6.0f) : 0.0f;
            }
        }
        cur = nxt;
        box = nbox;
    }
}
#endif  // ITERMVS_INIT_SWEEP_BUILD

constexpr int kInitNB = 8;  // hypotheses per block

template <int TILE, int FT>
__global__ void __launch_bounds__(kThreads) corr_init_kernel(const InitArgs a) {
    __shared__ float lds[kInitNB * ITERMVS_GROUPS * (TILE + 1)];
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
                                      int up_interleaved) {
    if ((int)blockIdx.x >= n_agg) {
        bilinear_up_body(w, B * S, H3, W3, 2, 0, w_up, (int64_t)(blockIdx.x - n_agg) * blockDim.x + threadIdx.x, up_interleaved ? S : 0);
        return;
    }
    const int64_t per = (int64_t)NG * P;
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
        const int tiles = ((p->W + kIterTW - 1) / kIterTW) * ((p->H + TILE / kIterTW - 1) / (TILE / kIterTW));
        const dim3 grid(((tiles + 7) / 8) * 8, 3, p->B);
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
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out, na, 0, 0, (float*)nullptr, vec4, 0);
    return itermvs_launch_status();
}

extern "C" int itermvs_view_aggregate_up(const float* corr, const float* w, int32_t S, int32_t B, int32_t N, int32_t H3, int32_t W3,
                                         float* out, float* w_up, int32_t w_up_interleaved, void* stream) {
    ITERMVS_RETURN_IF(!corr || !w || !out || !w_up, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(S < 1 || B < 1 || N < 1 || H3 < 1 || W3 < 1, ITERMVS_ERR_DIMS);
    const int P = H3 * W3;
    const int vec4 = (P & 3) == 0 && ((((uintptr_t)corr) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0;
    const int na = (int)(((int64_t)B * N * ITERMVS_GROUPS * P / (vec4 ? 4 : 1) + 255) / 256);
    const int nu = (int)(((int64_t)B * S * P * 4 + 255) / 256);
    hipLaunchKernelGGL(view_aggregate_kernel, dim3((unsigned)(na + nu)), dim3(256), 0, (hipStream_t)stream,
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out, na, H3, W3, w_up, vec4, w_up_interleaved ? 1 : 0);
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
