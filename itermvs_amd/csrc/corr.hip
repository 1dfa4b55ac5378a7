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
    const int tile = xcd_tile((P + TILE - 1) / TILE);
    const int p0 = tile * TILE;
    if (p0 >= P) return;          // padding blocks of the XCD-aligned grid (uniform per block)
    const int per_px = N * K::LPT;
    const int items = TILE * per_px;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = a.proj + ((size_t)(lvl * a.B + b) * a.S) * 12;
    const uint32_t sy = (uint32_t)L.sy, sx = (uint32_t)L.sx;
    const int lane = threadIdx.x & 63;

    // index arithmetic without integer division on the common path (an unsigned division costs ~25 vector instructions,
    // and these gather kernels are bound by instruction issue): items per pixel is a power of two for the reference's
    // hypothesis counts, and a tile of 32 consecutive pixels wraps at most one image row when W >= 32
    const int px_shift = (per_px & (per_px - 1)) == 0 ? 31 - __clz(per_px) : -1;
    const int tile_y0 = p0 / a.W, tile_x0 = p0 - tile_y0 * a.W;
#pragma unroll 1
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int px = px_shift >= 0 ? item >> px_shift : item / per_px;
        const int rem = item - px * per_px;
        const int n = rem / K::LPT;
        const int j = rem - n * K::LPT;      // the LPT lanes of one (pixel, hypothesis) are adjacent lanes
        const int p = p0 + px;
        if (p >= P) continue;                // whole lane groups drop out together
        int y = tile_y0, x = tile_x0 + px;
        if (a.W >= TILE) {
            if (x >= a.W) { x -= a.W; ++y; }
        } else {
            y = p / a.W; x = p - y * a.W;
        }

        float d;
        if (L.depth) {
            d = L.depth[((size_t)b * N + n) * P + p];
        } else {  // itermvs.py:291-293
            float ns = a.nd[b * a.nd_sb + p] + L.offs[n];
            ns = fminf(fmaxf(ns, 0.0f), 1.0f);
            d = unnormalize_depth(ns, inv_min, inv_max);
        }
        float refv[K::VEC];
        load_vec<K::VEC>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff + j * 4, refv);

        const float xs = (float)x * g.xr, ys = (float)y * g.yr;
        float acc[K::NG];
#pragma unroll
        for (int q = 0; q < K::NG; ++q) acc[q] = 0.0f;
        float wsum = 1e-5f;  // itermvs.py:88
        const uint32_t joff = (uint32_t)(j * 4);
        const int gbase = lane - j;
        // The projection and bilinear footprint of (pixel, hypothesis) in view s are the same for
        // all LPT chunk lanes: lane j computes them for view s0 + j, then the group walks the batch of
        // views and every lane fetches the footprint of view s0 + k from lane k (wavefront shuffles).
        for (int s0 = 0; s0 < a.S; s0 += K::LPT) {
            Footprint mine = {0u, 0u, 0u, 0u, 0.0f, 0.0f, 0.0f, 0.0f};
            if (s0 + j < a.S) {
                const float* m = proj + (s0 + j) * 12;
                float rx, ry, rz, ix, iy;
                ray_dir(m, xs, ys, rx, ry, rz);
                project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
                mine = make_footprint(ix, iy, L.W1, L.H1, sy, sx);
            }
            const int nb = min(K::LPT, a.S - s0);
            for (int k = 0; k < nb; ++k) {
                const Footprint tp = shfl_footprint(mine, gbase + k);
                float corr[K::NG];
                chunk_corr<CPG, FT>(feat_base<FT>(L.src[s0 + k], (int64_t)b * L.sb), joff, tp, refv, corr);
                const float wv = a.view_w[((size_t)b * a.S + s0 + k) * P + p];
#pragma unroll
                for (int q = 0; q < K::NG; ++q) acc[q] = acc[q] + corr[q] * wv;  // itermvs.py:115
                wsum = wsum + wv;                                                // itermvs.py:116
            }
        }
#pragma unroll
        for (int q = 0; q < K::NG; ++q) lds[(n * ITERMVS_GROUPS + j * K::NG + q) * LS + px] = acc[q] / wsum;
    }
    __syncthreads();
    const int rows = N * ITERMVS_GROUPS;
    if constexpr (TILE % 4 == 0) {
        // 16-byte stores where the rows allow it (a wave-level store costs the CU about the same whatever its width:
        // tools/ubench/tile_read.hip): P a multiple of 4 keeps every quad of a tile row inside one plane row and aligned
        if ((P & 3) == 0 && ((uintptr_t)L.out & 15) == 0) {
            for (int idx = threadIdx.x; idx < rows * (TILE / 4); idx += kThreads) {
                const int row = idx / (TILE / 4), px = (idx - row * (TILE / 4)) * 4;
                if (p0 + px < P) {
                    const float* __restrict__ l = lds + row * LS + px;
                    *reinterpret_cast<float4*>(L.out + ((size_t)b * rows + row) * P + p0 + px) = make_float4(l[0], l[1], l[2], l[3]);
                }
            }
            return;
        }
    }
    for (int idx = threadIdx.x; idx < rows * TILE; idx += kThreads) {
        const int row = idx / TILE, px = idx - row * TILE;
        if (p0 + px < P) L.out[((size_t)b * rows + row) * P + p0 + px] = lds[row * LS + px];
    }
}

template <int TILE, int FT>
__global__ void __launch_bounds__(kThreads) corr_iter_kernel(const IterArgs a) {
    __shared__ float lds[ITERMVS_MAX_HYP * ITERMVS_GROUPS * (TILE + 1)];
    const int lvl = blockIdx.y;
    const IterLevel& L = a.lv[lvl];
    switch (L.C) {
        case 16: corr_iter_level<2, TILE, FT>(a, L, lvl, lds); break;
        case 32: corr_iter_level<4, TILE, FT>(a, L, lvl, lds); break;
        default: corr_iter_level<6, TILE, FT>(a, L, lvl, lds); break;
    }
}

// ---------------------------------------------------------------------------------------------
// iteration branch, "views across waves" form (the default)
//
// The form above walks the S source views inside the lane: S dependent rounds of {footprint shuffle, 4 tap loads,
// wait, blend} per item, 70 such rounds per SIMD at cfg 1 -- the launch is bound by memory latency x rounds.
// Here a lane group owns one (pixel, view) and keeps ALL hypotheses of the level in registers:
//   * wave-rounds are (view, block of 64/LPT pixels): the view is wave-uniform, so the source base stays in SGPRs;
//   * quad lane u projects hypothesis u and builds its footprint ONCE; the four lanes of the quad exchange the four
//     tap offsets and four weights with DPP quad_perm moves (no LDS shuffles, no redundant projections);
//   * the 4 x N tap loads of the item are all issued before the first blend: one latency round per item;
//   * per-view group correlations go to LDS [view][hypothesis*8 + group][pixel]; after a barrier each thread owns one
//     (group, pixel) column, walks the views IN ORDER and applies the view weights with the arithmetic of
//     itermvs.py:115-120 (acc = acc + corr * w; wsum = wsum + w; acc / wsum) -- the results equal the in-lane form's
//     bit for bit -- and writes the [B,N,8,H,W] planes.
// S > 4 runs in chunks of 4 views with the accumulators carried in registers.
// ---------------------------------------------------------------------------------------------
template <int CPG>
__device__ __forceinline__ void blend_corr_vw(const TapData<2 * CPG>& t, const Footprint& tp, const float (&refv)[2 * CPG],
                                              float (&corr)[2]) {
    constexpr int VEC = 2 * CPG;
    float w[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c)
        w[c] = fmaf(tp.se, t.v11[c], fmaf(tp.sw, t.v10[c], fmaf(tp.ne, t.v01[c], tp.nw * t.v00[c])));
    if constexpr (CPG == 2) {
        corr[0] = fmaf(w[1], refv[1], w[0] * refv[0]) * 0.5f;
        corr[1] = fmaf(w[3], refv[3], w[2] * refv[2]) * 0.5f;
    } else if constexpr (CPG == 4) {   // channels 4j..4j+3 = group j, 16+4j.. = group 4+j
        corr[0] = fmaf(w[3], refv[3], fmaf(w[2], refv[2], fmaf(w[1], refv[1], w[0] * refv[0]))) * 0.25f;
        corr[1] = fmaf(w[7], refv[7], fmaf(w[6], refv[6], fmaf(w[5], refv[5], w[4] * refv[4]))) * 0.25f;
    } else {   // see blend_corr<6>: partial sums re-grouped inside the quad
        float lo[3], hi[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lo[i] = fmaf(w[4 * i + 1], refv[4 * i + 1], w[4 * i] * refv[4 * i]);
            hi[i] = fmaf(w[4 * i + 3], refv[4 * i + 3], w[4 * i + 2] * refv[4 * i + 2]);
        }
        const float s0 = lo[0] + hi[0], s1 = lo[1] + hi[1], s2 = lo[2] + hi[2];
        const int j = threadIdx.x & 3;
        const float ta = (j == 0 || j == 3) ? s0 : (j == 2 ? s1 : s2);
        const float tb = (j == 1) ? lo[0] : (j == 2 ? lo[2] : lo[1]);
        const float tc = (j == 1) ? hi[0] : (j == 2 ? hi[2] : hi[1]);
        const float tdd = (j == 2) ? s0 : (j == 1 ? s1 : s2);
        const float g_first = quad_perm<ITERMVS_QP(0, 3, 2, 1)>(ta) + quad_perm<ITERMVS_QP(1, 0, 3, 2)>(tb);
        const float g_second = quad_perm<ITERMVS_QP(1, 0, 3, 2)>(tc) + quad_perm<ITERMVS_QP(2, 1, 0, 3)>(tdd);
        corr[0] = div_rcp(g_first, 6.0f, 1.0f / 6.0f);   // mean over the 6 channels of the group (itermvs.py:103-104)
        corr[1] = div_rcp(g_second, 6.0f, 1.0f / 6.0f);
    }
}

template <int CPG, int NB>
__device__ __forceinline__ void corr_iter_vw_level(const IterArgs& a, const IterLevel& L, int lvl, float* __restrict__ lds,
                                                   int tw_log2) {
    using K = VwChunk<CPG>;
    constexpr int TILE = kVwTile, LS = TILE + 1;
    constexpr int PPW = 16;            // pixels per wave-round (one quad each)
    constexpr int RPV = TILE / PPW;    // wave-rounds per view
    const int N = L.N;
    const int VROWS = N * ITERMVS_GROUPS;    // LDS: [kVwViews][VROWS][LS] per-view correlations, then (S > 4 only) the
    float* carry = lds + kVwViews * VROWS * LS;   // accumulators [VROWS][LS] and weight sums [8][LS] carried between chunks
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    const int TW = 1 << tw_log2, TH = TILE >> tw_log2;
    const int tiles_x = (a.W + TW - 1) >> tw_log2, tiles_y = (a.H + TH - 1) / TH;
    const int tile = xcd_tile(tiles_x * tiles_y);
    if (tile >= tiles_x * tiles_y) return;   // padding blocks of the XCD-aligned grid (uniform per block)
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx << tw_log2, y0 = ty * TH;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = a.proj + ((size_t)(lvl * a.B + b) * a.S) * 12;
    const uint32_t sy = (uint32_t)L.sy, sx = (uint32_t)L.sx;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 3;
    const uint32_t joff = (uint32_t)(j * 4);

    // second phase: this thread owns correlation group r2 of pixel px2, for every hypothesis
    const int px2 = threadIdx.x & (TILE - 1), r2 = threadIdx.x >> 5;
    const int xx2 = x0 + (px2 & (TW - 1)), yy2 = y0 + (px2 >> tw_log2);
    const bool live2 = xx2 < a.W && yy2 < a.H;
    const int p2 = yy2 * a.W + xx2;

    for (int s0 = 0; s0 < a.S; s0 += kVwViews) {
        const int sbc = min(kVwViews, a.S - s0);
#pragma unroll 1
        for (int r = wave; r < sbc * RPV; r += kThreads / 64) {
            const int v = r / RPV, pb = r - v * RPV;   // wave-uniform
            const float* fb = L.src[s0 + v] + (int64_t)b * L.sb;
            const float* m = proj + (s0 + v) * 12;
            const int px = pb * PPW + (lane >> 2);
            const int x = x0 + (px & (TW - 1)), y = y0 + (px >> tw_log2);
            if (x < a.W && y < a.H) {   // whole quads drop out together
                const int p = y * a.W + x;
                float refv[K::VEC];
                load_vec<K::VEC>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff + j * 4, refv);
                float rx, ry, rz;
                ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
                const float nd_p = L.depth ? 0.0f : a.nd[b * a.nd_sb + p];
#pragma unroll 1
                for (int hb = 0; hb < N; hb += 4) {
                    // quad lane u projects hypothesis hb + u (surplus lanes repeat the last one) and builds its footprint
                    const int n_own = min(hb + j, N - 1);
                    float d;
                    if (L.depth) {
                        d = L.depth[((size_t)b * N + n_own) * P + p];
                    } else {   // itermvs.py:291-293
                        float off = L.offs[0];
#pragma unroll
                        for (int k = 1; k < ITERMVS_MAX_HYP; ++k) off = (n_own == k) ? L.offs[k] : off;
                        float ns = nd_p + off;
                        ns = fminf(fmaxf(ns, 0.0f), 1.0f);
                        d = unnormalize_depth(ns, inv_min, inv_max);
                    }
                    float ix, iy;
                    project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
                    const Footprint f = make_footprint(ix, iy, L.W1, L.H1, sy, sx);
                    const uint32_t o00 = f.r0 + f.c0, o01 = f.r0 + f.c1, o10 = f.r1 + f.c0, o11 = f.r1 + f.c1;
#pragma unroll
                    for (int lb = 0; lb < 4; lb += NB) {   // NB hypotheses' taps in flight at a time
                        if (hb + lb < N) {                 // uniform
                            TapData<K::VEC> td[NB];
                            Footprint wt[NB];
#pragma unroll
                            for (int u = 0; u < NB; ++u) {
                                load_vec<K::VEC>(fb + (quad_bcast(o00, lb + u) + joff), td[u].v00);
                                load_vec<K::VEC>(fb + (quad_bcast(o01, lb + u) + joff), td[u].v01);
                                load_vec<K::VEC>(fb + (quad_bcast(o10, lb + u) + joff), td[u].v10);
                                load_vec<K::VEC>(fb + (quad_bcast(o11, lb + u) + joff), td[u].v11);
                                wt[u].nw = quad_bcast(f.nw, lb + u); wt[u].ne = quad_bcast(f.ne, lb + u);
                                wt[u].sw = quad_bcast(f.sw, lb + u); wt[u].se = quad_bcast(f.se, lb + u);
                            }
#pragma unroll
                            for (int u = 0; u < NB; ++u) {
                                float corr[2];
                                blend_corr_vw<CPG>(td[u], wt[u], refv, corr);
                                const int n = hb + lb + u;
                                if (n < N) {
                                    lds[(v * VROWS + n * ITERMVS_GROUPS + K::group(j, 0)) * LS + px] = corr[0];
                                    lds[(v * VROWS + n * ITERMVS_GROUPS + K::group(j, 1)) * LS + px] = corr[1];
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (live2) {
            // (views in order, the arithmetic of itermvs.py:115-120; between chunks the partial sums rest in LDS so that no
            // register stays live across the gather phase)
            float acc[ITERMVS_MAX_HYP];
            float wsum = 1e-5f;   // itermvs.py:88
#pragma unroll
            for (int i = 0; i < ITERMVS_MAX_HYP; ++i) acc[i] = 0.0f;
            if (s0 > 0) {
                wsum = carry[(VROWS + r2) * LS + px2];
#pragma unroll
                for (int i = 0; i < ITERMVS_MAX_HYP; ++i)
                    if (i < N) acc[i] = carry[(i * ITERMVS_GROUPS + r2) * LS + px2];
            }
            for (int v = 0; v < sbc; ++v) {
                const float w = a.view_w[((size_t)b * a.S + s0 + v) * P + p2];
#pragma unroll
                for (int i = 0; i < ITERMVS_MAX_HYP; ++i)
                    if (i < N) acc[i] = acc[i] + lds[(v * VROWS + i * ITERMVS_GROUPS + r2) * LS + px2] * w;   // itermvs.py:115
                wsum = wsum + w;                                                                              // itermvs.py:116
            }
            if (s0 + kVwViews < a.S) {
                carry[(VROWS + r2) * LS + px2] = wsum;
#pragma unroll
                for (int i = 0; i < ITERMVS_MAX_HYP; ++i)
                    if (i < N) carry[(i * ITERMVS_GROUPS + r2) * LS + px2] = acc[i];
            } else {
#pragma unroll
                for (int i = 0; i < ITERMVS_MAX_HYP; ++i)
                    if (i < N) L.out[((size_t)b * VROWS + i * ITERMVS_GROUPS + r2) * P + p2] = acc[i] / wsum;
            }
        }
        if (s0 + kVwViews < a.S) __syncthreads();
    }
}

template <int NBA, int NBB, int NBC, int WAVES>
__global__ void __launch_bounds__(kThreads, WAVES) corr_iter_vw_kernel(const IterArgs a, const int tw_log2) {
    extern __shared__ float lds[];   // see corr_iter_vw_level; sized by the launch for the level with most hypotheses
    const int lvl = blockIdx.y;
    const IterLevel& L = a.lv[lvl];
    switch (L.C) {
        case 16: corr_iter_vw_level<2, NBA>(a, L, lvl, lds, tw_log2); break;
        case 32: corr_iter_vw_level<4, NBB>(a, L, lvl, lds, tw_log2); break;
        default: corr_iter_vw_level<6, NBC>(a, L, lvl, lds, tw_log2); break;
    }
}

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
    const int tile = xcd_tile((P + TILE - 1) / TILE);
    const int p0 = tile * TILE;
    if (p0 >= P) return;
    const int ngrp = (nb + K::LPT - 1) / K::LPT;   // groups of LPT hypotheses
    const int per_px = ngrp * K::LPT;
    const int items = TILE * per_px;
    const WarpGeom g = make_geom(a.W, a.H, a.W1, a.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* m = a.proj + ((size_t)b * a.S + s) * 12;
    const float* fsrc = feat_base<FT>(a.src[s], (int64_t)b * a.sb);
    const uint32_t sy = (uint32_t)a.sy, sx = (uint32_t)a.sx;
    const int lane = threadIdx.x & 63;

    // index arithmetic without integer division on the common path (an unsigned division costs ~25 vector instructions,
    // and these gather kernels are bound by instruction issue): items per pixel is a power of two for the reference's
    // hypothesis counts, and a tile of 32 consecutive pixels wraps at most one image row when W >= 32
    const int px_shift = (per_px & (per_px - 1)) == 0 ? 31 - __clz(per_px) : -1;
    const int tile_y0 = p0 / a.W, tile_x0 = p0 - tile_y0 * a.W;
#pragma unroll 1
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int px = px_shift >= 0 ? item >> px_shift : item / per_px;
        const int rem = item - px * per_px;
        const int grp = rem / K::LPT;
        const int j = rem - grp * K::LPT;
        const int p = p0 + px;
        if (p >= P) continue;
        int y = tile_y0, x = tile_x0 + px;
        if (a.W >= TILE) {
            if (x >= a.W) { x -= a.W; ++y; }
        } else {
            y = p / a.W; x = p - y * a.W;
        }
        float refv[K::VEC];
        if (a.ref.sc == 1) {       // channels-last reference (the engine's layout): the chunk is VEC/4 vector loads off one address
            load_feat<K::VEC, FT>(feat_base<FT>((const float*)a.ref.data, (int64_t)b * a.ref.sb),
                                  (uint32_t)(y * (int)a.ref.sy + x * (int)a.ref.sx) + (uint32_t)(j * 4), refv);
        } else {
#pragma unroll
            for (int c = 0; c < K::VEC; ++c)
                refv[c] = ld_feat<FT>((const float*)a.ref.data, b * a.ref.sb + chunk_channel<K::VEC>(j, c) * a.ref.sc + y * a.ref.sy + x * a.ref.sx);
        }
        // lane j projects hypothesis grp*LPT + j once; the group then walks its LPT hypotheses and
        // every lane reads the footprint of hypothesis k from lane k (wavefront shuffles)
        const int nl_mine = grp * K::LPT + j;
        Footprint mine = {0u, 0u, 0u, 0u, 0.0f, 0.0f, 0.0f, 0.0f};
        if (nl_mine < nb) {
            const int n = n0 + nl_mine;
            float d;
            if (a.depth) {
                d = a.depth[((size_t)b * a.N + n) * P + p];
            } else {  // itermvs.py:13-17
                const float frac = (float)n / (float)(a.N - 1);
                d = 1.0f / (inv_max + frac * (inv_min - inv_max));
            }
            float rx, ry, rz, ix, iy;
            ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
            project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
            mine = make_footprint(ix, iy, a.W1, a.H1, sy, sx);
        }
        const uint32_t joff = (uint32_t)(j * 4);
        const int gbase = lane - j;
        const int cnt = min(K::LPT, nb - grp * K::LPT);
        for (int k = 0; k < cnt; ++k) {
            const Footprint tp = shfl_footprint(mine, gbase + k);
            float corr[K::NG];
            chunk_corr<CPG, FT>(fsrc, joff, tp, refv, corr);
            const int nl = grp * K::LPT + k;
#pragma unroll
            for (int q = 0; q < K::NG; ++q) lds[(nl * ITERMVS_GROUPS + j * K::NG + q) * LS + px] = corr[q];
        }
    }
    __syncthreads();
    const int rows = nb * ITERMVS_GROUPS;
    float* o = a.out + (((size_t)b * a.S + s) * a.N + n0) * ITERMVS_GROUPS * P;
    if constexpr (TILE % 4 == 0) {
        if ((P & 3) == 0 && ((uintptr_t)a.out & 15) == 0) {      // 16-byte stores (see corr_iter_level)
            for (int idx = threadIdx.x; idx < rows * (TILE / 4); idx += kThreads) {
                const int row = idx / (TILE / 4), px = (idx - row * (TILE / 4)) * 4;
                if (p0 + px < P) {
                    const float* __restrict__ l = lds + row * LS + px;
                    *reinterpret_cast<float4*>(o + (size_t)row * P + p0 + px) = make_float4(l[0], l[1], l[2], l[3]);
                }
            }
            return;
        }
    }
    for (int idx = threadIdx.x; idx < rows * TILE; idx += kThreads) {
        const int row = idx / TILE, px = idx - row * TILE;
        if (p0 + px < P) o[(size_t)row * P + p0 + px] = lds[row * LS + px];
    }
}

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
                                      int P, float* __restrict__ out, int n_agg, int H3, int W3, float* __restrict__ w_up, int vec4) {
    if ((int)blockIdx.x >= n_agg) {
        bilinear_up_body(w, B * S, H3, W3, 2, 0, w_up, (int64_t)(blockIdx.x - n_agg) * blockDim.x + threadIdx.x);
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

}  // namespace itermvs

using namespace itermvs;

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

// defaults of itermvs_corr_iter's kernel choice (see the launch below); the environment overrides are for experiments
static int default_impl() {
    static const int v = [] {
        const char* e = getenv("ITERMVS_CORR_ITER_IMPL");
        const int n = e ? atoi(e) : 0;
        return n > 0 ? n : 1;
    }();
    return v;
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
    // impl % 10: 0 = default (ITERMVS_CORR_ITER_IMPL=<number> overrides), 1 = views walked in the lane (default), 2 = views
    // across waves with (4, 2, 1) hypotheses' taps in flight on the (C=16, 32, 48) levels, 3 = 2 held to 128 registers
    // (4 waves per SIMD); impl / 10 = log2 narrowing of the 32-pixel tile of forms 2 / 3 (0: 32x1 strips, 1: 16x2, 2: 8x4).
    // Measured on MI355X (profiles/r02): both forms sit at the same 24-34 us per launch at cfg 1 -- the launch is bound by
    // the rate at which a CU's vector L1 gets its misses served (1.2-1.6 M 128-byte requests per launch, ~4800 per CU),
    // the second form issues 14 % fewer vector instructions but misses the L1 31 % more often.
    int variant = p->impl % 10, narrow = p->impl / 10;
    if (variant == 0) {
        variant = default_impl() % 10;
        narrow = default_impl() / 10;
    }
    ITERMVS_RETURN_IF(variant < 1 || variant > 3 || narrow < 0 || narrow > 2, ITERMVS_ERR_DIMS);
    const int dtype = p->src[0].dtype;
    ITERMVS_RETURN_IF(p->src[1].dtype != dtype || p->src[2].dtype != dtype, ITERMVS_ERR_DTYPE);
    if (dtype != ITERMVS_F32) variant = 1;   // 16-bit feature storage: the in-lane form
    itermvs_profile_begin(1, (hipStream_t)stream);
    if (variant == 1) {
        constexpr int TILE = 32;
        const int P = p->H * p->W;
        const dim3 grid((((P + TILE - 1) / TILE + 7) / 8) * 8, 3, p->B);
        switch (dtype) {
            case ITERMVS_F16: hipLaunchKernelGGL((corr_iter_kernel<TILE, ITERMVS_F16>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
            case ITERMVS_BF16: hipLaunchKernelGGL((corr_iter_kernel<TILE, ITERMVS_BF16>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
            default: hipLaunchKernelGGL((corr_iter_kernel<TILE, ITERMVS_F32>), grid, dim3(kThreads), 0, (hipStream_t)stream, a); break;
        }
    } else {
        const int tw_log2 = 5 - narrow, tw = 1 << tw_log2, th = kVwTile / tw;
        const int tiles = ((p->W + tw - 1) / tw) * ((p->H + th - 1) / th);
        const dim3 grid(((tiles + 7) / 8) * 8, 3, p->B);
        const int nmax = max(p->N[0], max(p->N[1], p->N[2]));
        const size_t shmem = (size_t)((kVwViews + (p->S > kVwViews ? 1 : 0)) * nmax * ITERMVS_GROUPS + (p->S > kVwViews ? ITERMVS_GROUPS : 0)) *
                             (kVwTile + 1) * sizeof(float);
        if (variant == 3)
            hipLaunchKernelGGL((corr_iter_vw_kernel<4, 2, 1, 4>), grid, dim3(kThreads), shmem, (hipStream_t)stream, a, tw_log2);
        else
            hipLaunchKernelGGL((corr_iter_vw_kernel<4, 2, 1, 3>), grid, dim3(kThreads), shmem, (hipStream_t)stream, a, tw_log2);
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
    const int P = p->H * p->W;
    const int nblocks = (p->N + kInitNB - 1) / kInitNB;
    itermvs_profile_begin(2, (hipStream_t)stream);
    const dim3 grid((((P + TILE - 1) / TILE + 7) / 8) * 8, p->S * nblocks, p->B);
    ITERMVS_RETURN_IF(p->ref.dtype != p->src.dtype, ITERMVS_ERR_DTYPE);
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
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out, na, 0, 0, (float*)nullptr, vec4);
    return itermvs_launch_status();
}

extern "C" int itermvs_view_aggregate_up(const float* corr, const float* w, int32_t S, int32_t B, int32_t N, int32_t H3, int32_t W3,
                                         float* out, float* w_up, void* stream) {
    ITERMVS_RETURN_IF(!corr || !w || !out || !w_up, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(S < 1 || B < 1 || N < 1 || H3 < 1 || W3 < 1, ITERMVS_ERR_DIMS);
    const int P = H3 * W3;
    const int vec4 = (P & 3) == 0 && ((((uintptr_t)corr) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0;
    const int na = (int)(((int64_t)B * N * ITERMVS_GROUPS * P / (vec4 ? 4 : 1) + 255) / 256);
    const int nu = (int)(((int64_t)B * S * P * 4 + 255) / 256);
    hipLaunchKernelGGL(view_aggregate_kernel, dim3((unsigned)(na + nu)), dim3(256), 0, (hipStream_t)stream,
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out, na, H3, W3, w_up, vec4);
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
