// itermvs_corr_iter, LDS-staged tile variant (params->impl = 2).
//
// Same arithmetic as corr.hip's direct-gather kernel (models/itermvs.py:84-120 fused with
// models/module.py:68-125), different data movement, designed around what the gather costs on
// CDNA4: one (pixel, hypothesis, view) needs 4 taps x C floats, i.e. 377 MB of L1/TA requests per
// launch at cfg 1 against 50 MB of algorithmic HBM traffic -- the texture path, not HBM, bounds
// the direct gather.  Here a workgroup owns a TILE of reference pixels of one pyramid level and,
// per source view:
//   A. every thread projects its (pixel, hypothesis) items ONCE (no redundancy across channel
//      chunks) and the block reduces the bounding box of all bilinear footprints;
//   B. the box -- a (rows x cols x C) patch whose rows are CONTIGUOUS in the channels-last source
//      map -- is copied global -> LDS with fully coalesced 16-byte loads (each source pixel read
//      once per tile instead of once per tap);
//   C. every thread blends its 4 taps for all C channels from LDS (ds_read_b128), multiplies with
//      the reference features it keeps in registers, reduces the 8 correlation groups in-thread
//      and accumulates the view-weighted sums in registers.
// Footprints that do not fit the LDS budget (degenerate cameras, behind-camera patches) fall back
// to direct global gathers for that (tile, view): same results, just slower.
// Lanes map to consecutive pixels of a tile row, so the [B,N,8,H,W] outputs are written in 64-byte
// runs; tiles are dealt to XCDs in contiguous bands so halo re-reads hit the same L2.
#include "common.hpp"

namespace itermvs {

constexpr int kLdsThreads = 256;
constexpr int kLdsFloats = 10240;   // 40 KiB patch budget per workgroup -> 3-4 workgroups per CU
constexpr int kPadFloats = 4;       // 16-byte pad per staged pixel: spreads pixels over LDS bank slots

struct LdsLevel {
    const float* src[ITERMVS_MAX_SRC];
    int64_t sb, sy, sx;
    const float* depth;
    float* out;
    float offs[ITERMVS_MAX_HYP];
    int C, H1, W1, N, coff;
    int tw, th, tiles_x, tiles, first_block;  // tile shape and this level's block range
};

struct LdsArgs {
    LdsLevel lv[3];
    const float* ref_q;
    const float* proj;
    const float* view_w;
    const float* nd;
    int64_t nd_sb;
    const float* inv_min;
    const float* inv_max;
    int B, S, H, W, CQ, total_blocks;
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// One level, C channels, NI items (hypotheses) per thread.
template <int C, int NI>
__device__ __forceinline__ void corr_tile(const LdsArgs& a, const LdsLevel& L, int lvl, int tile, int b,
                                          float* __restrict__ patch, int* __restrict__ box) {
    constexpr int CPG = C / ITERMVS_GROUPS;
    constexpr int CS = C + kPadFloats;            // staged pixel stride (floats)
    constexpr int CAP = kLdsFloats / CS;          // pixels that fit
    const int tid = threadIdx.x;
    const int TP = L.tw * L.th;                   // pixels per tile (divides 256)
    const int px = tid % TP;
    const int n_first = tid / TP;                 // first hypothesis of this thread
    const int n_step = kLdsThreads / TP;
    const int ty = px / L.tw, tx = px - ty * L.tw;
    const int tile_y = tile / L.tiles_x, tile_x = tile - tile_y * L.tiles_x;
    const int x = tile_x * L.tw + tx, y = tile_y * L.th + ty;
    const bool in_img = x < a.W && y < a.H;
    const int P = a.H * a.W;
    const int p = in_img ? y * a.W + x : 0;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = a.proj + ((size_t)(lvl * a.B + b) * a.S) * 12;

    // hypotheses of this thread (itermvs.py:291-293 when generated) and reference features
    float d[NI];
    bool live[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n_first + i * n_step;
        live[i] = in_img && n < L.N;
        if (!live[i]) {
            d[i] = 1.0f;
        } else if (L.depth) {
            d[i] = L.depth[((size_t)b * L.N + n) * P + p];
        } else {
            float ns = a.nd[b * a.nd_sb + p] + L.offs[n];
            ns = fminf(fmaxf(ns, 0.0f), 1.0f);
            d[i] = unnormalize_depth(ns, inv_min, inv_max);
        }
    }
    float refv[C];
    {
        const float4* rp = reinterpret_cast<const float4*>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff);
#pragma unroll
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 t = rp[c4];
            refv[4 * c4] = t.x; refv[4 * c4 + 1] = t.y; refv[4 * c4 + 2] = t.z; refv[4 * c4 + 3] = t.w;
        }
    }
    const float xs = (float)x * g.xr, ys = (float)y * g.yr;
    float acc[NI][ITERMVS_GROUPS];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < ITERMVS_GROUPS; ++q) acc[i][q] = 0.0f;
    float wsum = 1e-5f;  // itermvs.py:88

    for (int s = 0; s < a.S; ++s) {
        const float* m = proj + s * 12;
        float rx, ry, rz;
        ray_dir(m, xs, ys, rx, ry, rz);
        // ---- A: project, bounding box of the footprints that touch the map ----------------
        Taps tp[NI];
        int bx0 = 1 << 30, by0 = 1 << 30, bx1 = -1, by1 = -1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float ix, iy;
            project(g, m, rx, ry, rz, d[i], ix, iy, nullptr);
            tp[i] = make_taps(ix, iy, L.W1, L.H1);
            const bool any = live[i] && (tp[i].nw != 0.0f || tp[i].ne != 0.0f || tp[i].sw != 0.0f || tp[i].se != 0.0f);
            if (any) {  // clamped tap coordinates are always inside the map
                const bool ux0 = tp[i].nw != 0.0f || tp[i].sw != 0.0f, ux1 = tp[i].ne != 0.0f || tp[i].se != 0.0f;
                const bool uy0 = tp[i].nw != 0.0f || tp[i].ne != 0.0f, uy1 = tp[i].sw != 0.0f || tp[i].se != 0.0f;
                if (ux0) { bx0 = min(bx0, tp[i].x0); bx1 = max(bx1, tp[i].x0); }
                if (ux1) { bx0 = min(bx0, tp[i].x1); bx1 = max(bx1, tp[i].x1); }
                if (uy0) { by0 = min(by0, tp[i].y0); by1 = max(by1, tp[i].y0); }
                if (uy1) { by0 = min(by0, tp[i].y1); by1 = max(by1, tp[i].y1); }
            }
        }
        bx0 = wave_min(bx0); by0 = wave_min(by0); bx1 = wave_max(bx1); by1 = wave_max(by1);
        const int wave = tid >> 6;
        if ((tid & 63) == 0) {
            box[wave * 4 + 0] = bx0; box[wave * 4 + 1] = by0; box[wave * 4 + 2] = bx1; box[wave * 4 + 3] = by1;
        }
        __syncthreads();  // also: everybody is done reading the previous view's patch
        int X0 = box[0], Y0 = box[1], X1 = box[2], Y1 = box[3];
#pragma unroll
        for (int wv = 1; wv < kLdsThreads / 64; ++wv) {
            X0 = min(X0, box[wv * 4]); Y0 = min(Y0, box[wv * 4 + 1]);
            X1 = max(X1, box[wv * 4 + 2]); Y1 = max(Y1, box[wv * 4 + 3]);
        }
        const int pw = X1 - X0 + 1, ph = Y1 - Y0 + 1;
        const bool empty = X1 < 0;                       // no footprint touches the map: all zeros
        const bool staged = !empty && pw * ph <= CAP;
        const float* fsrc = L.src[s] + (int64_t)b * L.sb;
        // ---- B: stage the patch (rows are contiguous in the channels-last map) ---------------
        if (staged) {
            const int row_chunks = pw * (C / 4);
            const int lane = tid & 63;
            for (int r = wave; r < ph; r += kLdsThreads / 64) {
                const float4* grow = reinterpret_cast<const float4*>(fsrc + (int64_t)(Y0 + r) * L.sy + (int64_t)X0 * L.sx);
                float* lrow = patch + (size_t)r * pw * CS;
                for (int q = lane; q < row_chunks; q += 64) {
                    const int col = q / (C / 4), c4 = q - col * (C / 4);
                    *reinterpret_cast<float4*>(lrow + col * CS + c4 * 4) = grow[q];
                }
            }
        }
        __syncthreads();
        // ---- C: blend taps, correlate with the reference, accumulate -------------------------
        const float w = in_img ? a.view_w[((size_t)b * a.S + s) * P + p] : 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float gs[ITERMVS_GROUPS];
#pragma unroll
            for (int q = 0; q < ITERMVS_GROUPS; ++q) gs[q] = 0.0f;
            const Taps& t = tp[i];
            if (staged) {
                const float* p00 = patch + ((t.y0 - Y0) * pw + (t.x0 - X0)) * CS;
                const float* p01 = patch + ((t.y0 - Y0) * pw + (t.x1 - X0)) * CS;
                const float* p10 = patch + ((t.y1 - Y0) * pw + (t.x0 - X0)) * CS;
                const float* p11 = patch + ((t.y1 - Y0) * pw + (t.x1 - X0)) * CS;
                // taps with zero weight may point outside the box: redirect them to a safe slot
                const float* z = patch;
                const float* q00 = t.nw != 0.0f ? p00 : z;
                const float* q01 = t.ne != 0.0f ? p01 : z;
                const float* q10 = t.sw != 0.0f ? p10 : z;
                const float* q11 = t.se != 0.0f ? p11 : z;
#pragma unroll
                for (int c4 = 0; c4 < C / 4; ++c4) {
                    const float4 v00 = *reinterpret_cast<const float4*>(q00 + 4 * c4);
                    const float4 v01 = *reinterpret_cast<const float4*>(q01 + 4 * c4);
                    const float4 v10 = *reinterpret_cast<const float4*>(q10 + 4 * c4);
                    const float4 v11 = *reinterpret_cast<const float4*>(q11 + 4 * c4);
                    const float wv[4] = {
                        fmaf(t.se, v11.x, fmaf(t.sw, v10.x, fmaf(t.ne, v01.x, t.nw * v00.x))),
                        fmaf(t.se, v11.y, fmaf(t.sw, v10.y, fmaf(t.ne, v01.y, t.nw * v00.y))),
                        fmaf(t.se, v11.z, fmaf(t.sw, v10.z, fmaf(t.ne, v01.z, t.nw * v00.z))),
                        fmaf(t.se, v11.w, fmaf(t.sw, v10.w, fmaf(t.ne, v01.w, t.nw * v00.w)))};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 4 * c4 + k;
                        gs[c / CPG] = (c % CPG == 0) ? wv[k] * refv[c] : fmaf(wv[k], refv[c], gs[c / CPG]);
                    }
                    if (c4 % 2 == 1) __builtin_amdgcn_sched_barrier(0);   // bound the LDS reads in flight (VGPRs)
                }
            } else if (!empty && live[i]) {
                // slow path (footprint larger than the LDS budget): direct gathers, one 16-byte chunk at
                // a time; dynamic channel index, so the reference chunk is re-read from memory
                const float4* g00 = reinterpret_cast<const float4*>(fsrc + t.y0 * L.sy + t.x0 * L.sx);
                const float4* g01 = reinterpret_cast<const float4*>(fsrc + t.y0 * L.sy + t.x1 * L.sx);
                const float4* g10 = reinterpret_cast<const float4*>(fsrc + t.y1 * L.sy + t.x0 * L.sx);
                const float4* g11 = reinterpret_cast<const float4*>(fsrc + t.y1 * L.sy + t.x1 * L.sx);
                const float4* rq = reinterpret_cast<const float4*>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff);
#pragma unroll 1
                for (int c4 = 0; c4 < C / 4; ++c4) {
                    const float4 v00 = g00[c4], v01 = g01[c4], v10 = g10[c4], v11 = g11[c4], r4 = rq[c4];
                    const float wv[4] = {
                        fmaf(t.se, v11.x, fmaf(t.sw, v10.x, fmaf(t.ne, v01.x, t.nw * v00.x))),
                        fmaf(t.se, v11.y, fmaf(t.sw, v10.y, fmaf(t.ne, v01.y, t.nw * v00.y))),
                        fmaf(t.se, v11.z, fmaf(t.sw, v10.z, fmaf(t.ne, v01.z, t.nw * v00.z))),
                        fmaf(t.se, v11.w, fmaf(t.sw, v10.w, fmaf(t.ne, v01.w, t.nw * v00.w)))};
                    const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int grp = (4 * c4 + k) / CPG;
#pragma unroll
                        for (int q = 0; q < ITERMVS_GROUPS; ++q)
                            gs[q] = (grp == q) ? fmaf(wv[k], rr[k], gs[q]) : gs[q];   // fma(w, r, +0) == w*r
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < ITERMVS_GROUPS; ++q) {
                float corr;
                if constexpr (CPG == 2) corr = gs[q] * 0.5f;
                else if constexpr (CPG == 4) corr = gs[q] * 0.25f;
                else corr = gs[q] / (float)CPG;
                acc[i][q] = acc[i][q] + corr * w;  // itermvs.py:115
            }
        }
        wsum = wsum + w;  // itermvs.py:116
    }
    // ---- output: [B, N, 8, H, W] ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (!live[i]) continue;
        const int n = n_first + i * n_step;
        float* o = L.out + ((size_t)b * L.N + n) * ITERMVS_GROUPS * P + p;
#pragma unroll
        for (int q = 0; q < ITERMVS_GROUPS; ++q) o[(size_t)q * P] = acc[i][q] / wsum;
    }
}

template <int C>
__device__ __forceinline__ void corr_tile_ni(const LdsArgs& a, const LdsLevel& L, int lvl, int tile, int b, float* patch,
                                             int* box) {
    const int per_thread = (L.N * L.tw * L.th + kLdsThreads - 1) / kLdsThreads;
    if constexpr (C <= 32) {
        if (per_thread <= 1) corr_tile<C, 1>(a, L, lvl, tile, b, patch, box);
        else corr_tile<C, 2>(a, L, lvl, tile, b, patch, box);   // host guarantees per_thread <= 2
    } else {
        corr_tile<C, 1>(a, L, lvl, tile, b, patch, box);        // host guarantees per_thread == 1 for C = 48
    }
}

__global__ void __launch_bounds__(kLdsThreads, 3) corr_iter_lds_kernel(const LdsArgs a) {
    __shared__ __attribute__((aligned(16))) float patch[kLdsFloats];
    __shared__ int box[16];
    // XCD-aware order: block k runs on XCD k % 8 (observed, used for speed only); give each XCD a
    // contiguous band of tiles so neighbouring tiles' halos are served by the same L2.
    const int nb = a.total_blocks;
    const int per_xcd = (nb + 7) / 8;
    const int blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;   // bijection onto [0, 8*per_xcd)
    if (blk >= nb) return;
    const int b = blockIdx.y;
    int lvl = 0;
    if (blk >= a.lv[1].first_block) lvl = 1;
    if (blk >= a.lv[2].first_block) lvl = 2;
    const LdsLevel& L = a.lv[lvl];
    const int tile = blk - L.first_block;
    switch (L.C) {
        case 16: corr_tile_ni<16>(a, L, lvl, tile, b, patch, box); break;
        case 32: corr_tile_ni<32>(a, L, lvl, tile, b, patch, box); break;
        default: corr_tile_ni<48>(a, L, lvl, tile, b, patch, box); break;
    }
}

}  // namespace itermvs

using namespace itermvs;

// called from itermvs_corr_iter (corr.hip) after argument validation; returns 1 if this variant
// cannot take the problem (source rows not densely packed, too many hypotheses per tile)
int itermvs_corr_iter_lds(const itermvs_corr_iter_params* p, hipStream_t stream) {
    LdsArgs a;
    int coff = 0, first = 0;
    for (int l = 0; l < 3; ++l) {
        LdsLevel& L = a.lv[l];
        if (p->src[l].sx != p->src[l].C) return 1;  // a patch row must be one contiguous run
        // 8x8-pixel tiles where the source map is finer than the sample grid (large footprints),
        // 16x8 otherwise; a tile's pixel count must divide the 256 threads
        L.tw = (p->src[l].W > p->W) ? 8 : 16;
        L.th = 8;
        L.tiles_x = (p->W + L.tw - 1) / L.tw;
        L.tiles = L.tiles_x * ((p->H + L.th - 1) / L.th);
        const int per_thread = (p->N[l] * L.tw * L.th + kLdsThreads - 1) / kLdsThreads;
        if (per_thread > (p->src[l].C <= 32 ? 2 : 1)) return 1;   // register budget: keeps 3-4 workgroups per CU
        L.first_block = first;
        first += L.tiles;
        for (int v = 0; v < ITERMVS_MAX_SRC; ++v) L.src[v] = p->src[l].view[v < p->S ? v : 0];
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
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.CQ = coff; a.total_blocks = first;
    const int grid_x = ((first + 7) / 8) * 8;
    hipLaunchKernelGGL(corr_iter_lds_kernel, dim3(grid_x, p->B), dim3(kLdsThreads), 0, stream, a);
    return itermvs_launch_status();
}
