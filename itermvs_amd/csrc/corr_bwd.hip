// Gradient of the fused warp + group-correlation kernels (itermvs_corr_iter / itermvs_corr_init) for training
// (train.py:194-243; the training branches of models/itermvs.py:59-61, 111-113).
//
// Forward, per 1/4-res (iteration) or 1/8-res (initialisation) pixel p, view s, hypothesis n, group g:
//     W_s,n[c]   = sum_t w_t * src_s[tap_t][c]                      bilinear gather (module.py:117-119)
//     corr_s,n,g = (1/cpg) sum_{c in g} W_s,n[c] * ref[c]           itermvs.py:50-51 / 103-104
//     iteration:  out_n,g = sum_s w_s corr_s,n,g / (1e-5 + sum_s w_s)   (view weights detached, itermvs.py:295)
// The sampling grid carries no gradient (module.py:77 torch.no_grad), so with E_s,n,g = dL/dcorr_s,n,g
// (= dL/dout_n,g * w_s / wsum in the iteration branch):
//     dL/dref[c]            = sum_{s,n} E_s,n,g(c) / cpg * W_s,n[c]          gather, recomputed -- never stored
//     dL/dsrc_s[tap_t][c]  += E_s,n,g(c) / cpg * ref[c] * w_t                scatter-add, fp32 hardware atomics
// No [B,C,N,H,W] warped volume exists in training either.
//
// Decomposition.  The scatter is bound by the rate of atomic REQUESTS, not of lanes: MI355X retires ~21 G 64-byte atomic
// segments per second whatever the collisions (tools/ubench/atomic_rate.hip: 84 G lane-atomics/s when a wave-instruction
// touches 16 segments, 335 G/s when it touches 4 full ones; LDS float atomics ~200 G/s, so an LDS window in front of the
// scatter loses).  Hence a ROW of 16 lanes owns one pixel and lane t of the row owns channel 16*blk + t of every
// 16-channel block: each atomic instruction of a wave covers four complete 64-byte segments (one per row) -- the round-2
// form (a quad per (pixel, view), a float4 of channels per lane, one channel per instruction) touched 16 segments of four
// dwords and ran at exactly the 84 G/s figure (1.1-1.5 ms per launch at the cfg-4 shape).  All 16 lanes of a row project
// the same (pixel, hypothesis) -- redundant arithmetic that is free under the atomic bound -- and walk the row's
// (view, hypothesis) steps with the NEXT step's tap loads issued before the current step's atomics, so a wave never waits
// for its own atomics (memory operations of a wave retire in order).  E is staged in LDS per chunk of views (iteration
// branch) or per (view, 8 hypotheses) (initialisation branch: those units are spread over blockIdx.y, the machine needs
// >= 8k waves in flight and 1/8-res maps have few pixels); dL/dref stays in registers (iteration: stored once) or is
// added to the zero-filled gradient of the reference view (initialisation: several blocks per pixel).
#include "corr_common.hpp"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <type_traits>

namespace itermvs {

struct BwdLevel {
    const float* src[ITERMVS_MAX_SRC];
    float* gsrc[ITERMVS_MAX_SRC];   // same strides as src, zero-filled by the caller
    int64_t sb, sy, sx;
    const float* depth;             // explicit hypotheses [B,N,P] or nullptr
    const float* gout;              // iteration: [B,N,8,P]; initialisation: [B,S,N,8,P]
    const float* ref;               // reference features of this level, element (b, c, y, x) at b*rsb + c + y*rsy + x*rsx
    float* gref;                    // same addressing as ref; iteration: written, initialisation: accumulated (zero-filled by the caller)
    int64_t rsb, rsy, rsx;
    const float* proj;              // [B,S,12] of this level
    float offs[ITERMVS_MAX_HYP];
    int C, H1, W1, N;
};

struct BwdArgs {
    BwdLevel lv[3];
    const float* view_w;            // [B,S,P] (iteration) or nullptr (initialisation: E = gout)
    const float* nd;                // normalised depth for generated hypotheses (iteration)
    int64_t nd_sb;
    const float* inv_min;
    const float* inv_max;
    int B, S, H, W;
    int init;                       // 1: initialisation branch (per-view gout, hypotheses uniform in inverse depth)
    int vch;                        // iteration: views per staged chunk (vch * N <= 32)
    int hyp_chunks;                 // initialisation: chunks of 8 hypotheses per view (grid.y = S * hyp_chunks)
    int scatter;                    // 1: dL/dsrc of every step; 2 (initialisation, generated planes): only of the (view, plane)
                                    // pairs plane_inverse routes here, init_gather_kernel serves the others; 0: dL/dref only
};

// Inverse of the (view, plane) homography of the initialisation branch (see init_gather_level) and the decision which
// kernel produces that pair's dL/dsrc.  G_n^-1 (row-major, formed in fp64) goes to o[0..8]; the return value is true when
// the atomic-free gather can serve the WHOLE source image for this pair: G_n is invertible and the denominator of the
// inverse mapping keeps one sign, with a margin, over the source image grown by a pixel square (it is affine in the pixel
// coordinates, so its extremes sit at the four corners) -- i.e. the plane's vanishing line stays clear of the image.
// Otherwise the pair is ROUTED to the atomic scatter of corr_bwd_kernel (the same decision is taken there, by the same
// arithmetic): a degenerate camera costs the scatter's time for its planes, not a scan of the reference grid per source
// pixel (O(P1 * N * P) projections: seconds at 1/8 of 1920x1280).
__device__ __forceinline__ bool plane_inverse(const float* __restrict__ m, float d_f, const WarpGeom& g, int W1, int H1,
                                              float* __restrict__ o) {
    const double d = (double)d_f;
    const double kx = (double)g.w1m1 * 0.5 / (double)g.half_w, ky = (double)g.h1m1 * 0.5 / (double)g.half_h;
    double G[9];
    for (int r = 0; r < 3; ++r) {
        const double k = r == 0 ? kx : (r == 1 ? ky : 1.0);
        G[3 * r + 0] = k * d * (double)m[4 * r + 0] * (double)g.xr;
        G[3 * r + 1] = k * d * (double)m[4 * r + 1] * (double)g.yr;
        G[3 * r + 2] = k * (d * (double)m[4 * r + 2] + (double)m[4 * r + 3]);
    }
    const double c0 = G[4] * G[8] - G[5] * G[7], c1 = G[5] * G[6] - G[3] * G[8], c2 = G[3] * G[7] - G[4] * G[6];
    const double det = G[0] * c0 + G[1] * c1 + G[2] * c2;
    double nrm = 0.0;
    for (int i = 0; i < 9; ++i) nrm += G[i] * G[i];
    const bool usable = det == det && fabs(det) > 1e-13 * nrm * sqrt(nrm);
    const double id = usable ? 1.0 / det : 0.0;
    float inv[9];
    inv[0] = (float)(c0 * id); inv[1] = (float)((G[2] * G[7] - G[1] * G[8]) * id); inv[2] = (float)((G[1] * G[5] - G[2] * G[4]) * id);
    inv[3] = (float)(c1 * id); inv[4] = (float)((G[0] * G[8] - G[2] * G[6]) * id); inv[5] = (float)((G[2] * G[3] - G[0] * G[5]) * id);
    inv[6] = (float)(c2 * id); inv[7] = (float)((G[1] * G[6] - G[0] * G[7]) * id); inv[8] = (float)((G[0] * G[4] - G[1] * G[3]) * id);
    bool whole = usable;
    float w2_first = 0.0f;
    for (int c = 0; c < 4; ++c) {
        const float cx = (c & 1) ? (float)(W1 - 1) + 1.02f : -1.02f, cy = (c & 2) ? (float)(H1 - 1) + 1.02f : -1.02f;
        const float w0 = fmaf(inv[0], cx, fmaf(inv[1], cy, inv[2]));
        const float w1 = fmaf(inv[3], cx, fmaf(inv[4], cy, inv[5]));
        const float w2 = fmaf(inv[6], cx, fmaf(inv[7], cy, inv[8]));
        if (c == 0) w2_first = w2;
        // ten times the margin the per-pixel test of the gather asks for: a pair that passes here passes there everywhere
        whole = whole && (w2 * w2_first > 0.0f) && fabsf(w2) > 1e-5f * (fabsf(w0) + fabsf(w1) + fabsf(w2));
    }
    if (o)
        for (int i = 0; i < 9; ++i) o[i] = inv[i];
    return whole;
}
// depth of plane n of the initialisation branch (itermvs.py:13-17, the forward kernel's expression)
__device__ __forceinline__ float plane_depth(int n, int N, float inv_min, float inv_max) {
    return init_hypothesis(n, N, inv_min, inv_max);
}

constexpr int kBwdTile = 16;                                           // pixels per block = rows of 16 lanes
constexpr int kBwdLS = kBwdTile + 1;
constexpr int kBwdRows = 4 * ITERMVS_MAX_HYP * ITERMVS_GROUPS;         // staged E rows: (views) x (hypotheses) x 8 <= 256
constexpr int kBwdLdsFloats = kBwdRows * kBwdLS + kBwdTile + ITERMVS_MAX_SRC * 12 + 8;   // + wsum [px] + the views' 3x4 matrices + routed[8]

// one (view, hypothesis) of a row's pixel: tap offsets (elements, incl. the lane's channel), weights, the loaded taps
template <int NB>
struct BwdStep {
    uint32_t o[4];
    float w[4];
    float tap[4 * NB];
};

template <int CPG, int FT>
__device__ __forceinline__ void corr_bwd_level(const BwdArgs& a, const BwdLevel& L, float* __restrict__ lds) {
    constexpr int TILE = kBwdTile, LS = kBwdLS, C = 8 * CPG, NB = C / 16;
    const int N = L.N;
    float* __restrict__ e_lds = lds;                            // [rows][LS]
    float* __restrict__ wsum_lds = lds + kBwdRows * LS;         // [TILE]
    float* __restrict__ m_lds = wsum_lds + TILE;                // [S][12]: read back with LDS broadcasts -- as vector loads
                                                                // they would put a vmcnt(0) (= wait for every atomic in
                                                                // flight) in front of each step
    float* __restrict__ routed_lds = m_lds + ITERMVS_MAX_SRC * 12;   // [8]: planes of this block whose dL/dsrc the gather kernel
                                                                     // does NOT produce (a.scatter == 2, see plane_inverse)
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    const int tile = xcd_tile((P + TILE - 1) / TILE);
    const int p0 = tile * TILE;
    if (p0 >= P) return;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const uint32_t sy = (uint32_t)L.sy, sx = (uint32_t)L.sx;
    const int px = threadIdx.x >> 4, t = threadIdx.x & 15;     // row = pixel of the tile, lane of the row = channel of a block
    const float inv_cpg = CPG == 2 ? 0.5f : (CPG == 4 ? 0.25f : 1.0f / 6.0f);

    // the views and hypotheses of this block
    int s_begin = 0, s_end = a.S, vch = a.vch, n0 = 0, nh = N;
    if (a.init) {
        s_begin = blockIdx.y / a.hyp_chunks; s_end = s_begin + 1; vch = 1;
        n0 = (blockIdx.y % a.hyp_chunks) * 8; nh = min(8, N - n0);
    }

    if (threadIdx.x < TILE) {
        float ws = 1e-5f;   // itermvs.py:88
        const int p = p0 + threadIdx.x;
        if (a.view_w && p < P)
            for (int s = 0; s < a.S; ++s) ws = ws + a.view_w[((size_t)b * a.S + s) * P + p];
        wsum_lds[threadIdx.x] = ws;
    }
    if (threadIdx.x < a.S * 12) m_lds[threadIdx.x] = L.proj[(size_t)b * a.S * 12 + threadIdx.x];
    __syncthreads();
    if (a.scatter == 2) {          // initialisation branch, generated planes: which (view, plane) pairs of this block are routed here
        if ((int)threadIdx.x < nh)
            routed_lds[threadIdx.x] = plane_inverse(m_lds + s_begin * 12, plane_depth(n0 + (int)threadIdx.x, N, inv_min, inv_max), g,
                                                    L.W1, L.H1, nullptr) ? 0.0f : 1.0f;
        __syncthreads();
    }

    const int p = p0 + px;
    const bool ok = p < P;
    const int pc = ok ? p : P - 1;                              // rows past the end compute on the last pixel and add nothing
    const int y = pc / a.W, x = pc - y * a.W;
    const float xs = (float)x * g.xr, ys = (float)y * g.yr;
    // the reference features: the fp32 ref_q pack (iteration) or the level-3 features themselves (initialisation)
    const int64_t roff = (int64_t)b * L.rsb + (int64_t)y * L.rsy + (int64_t)x * L.rsx + t;
    float refv[NB], gacc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        refv[blk] = a.init ? ld_feat<FT>(L.ref, roff + 16 * blk) : L.ref[roff + 16 * blk];
        gacc[blk] = 0.0f;
    }
    const float ndv = (!L.depth && !a.init) ? a.nd[b * a.nd_sb + pc] : 0.0f;

    for (int s0 = s_begin; s0 < s_end; s0 += vch) {
        const int sbc = min(vch, s_end - s0);
        const int rows = sbc * nh * ITERMVS_GROUPS;
        // E of this chunk -> LDS (already divided by the channels per group); row r = ((view, hypothesis), group)
        for (int i = threadIdx.x; i < rows * TILE; i += kThreads) {
            const int ipx = i % TILE, r = i / TILE;
            const int gi = r % ITERMVS_GROUPS, hn = (r / ITERMVS_GROUPS) % nh, v = r / (ITERMVS_GROUPS * nh);
            const int ip = p0 + ipx;
            float e = 0.0f;
            if (ip < P) {
                const int n = n0 + hn;
                if (a.init) {
                    e = L.gout[((((size_t)b * a.S + s0 + v) * N + n) * ITERMVS_GROUPS + gi) * P + ip] * inv_cpg;
                } else {
                    const float w = a.view_w[((size_t)b * a.S + s0 + v) * P + ip];
                    e = L.gout[(((size_t)b * N + n) * ITERMVS_GROUPS + gi) * P + ip] * (w / wsum_lds[ipx]) * inv_cpg;
                }
            }
            e_lds[r * LS + ipx] = e;
        }
        __syncthreads();

        // step k = (view k / nh, hypothesis n0 + k % nh): project, footprint, issue the tap loads
        // (`explicit_depth`: hypotheses read from L.depth -- a compile-time flag of the step loop, because a vector load on one
        // branch of the step makes the compiler drain vmcnt, i.e. every atomic in flight, where the branches join)
        auto prepare = [&](int k, BwdStep<NB>& st, auto explicit_depth) {
            const int v = k / nh, n = n0 + (k - v * nh);                       // block-uniform
            const float* m = m_lds + (s0 + v) * 12;
            float rx, ry, rz;
            ray_dir(m, xs, ys, rx, ry, rz);
            float d;
            if constexpr (decltype(explicit_depth)::value) {
                d = L.depth[((size_t)b * N + n) * P + pc];
            } else if (a.init) {   // itermvs.py:13-17
                d = init_hypothesis(n, N, inv_min, inv_max);
            } else {               // itermvs.py:291-293
                d = iter_hypothesis(ndv, L.offs[n], inv_min, inv_max);
            }
            float ix, iy;
            project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
            const Footprint f = make_footprint(ix, iy, L.W1, L.H1, sy, sx);
            st.o[0] = f.r0 + f.c0 + (uint32_t)t; st.o[1] = f.r0 + f.c1 + (uint32_t)t;
            st.o[2] = f.r1 + f.c0 + (uint32_t)t; st.o[3] = f.r1 + f.c1 + (uint32_t)t;
            st.w[0] = f.nw; st.w[1] = f.ne; st.w[2] = f.sw; st.w[3] = f.se;
            const float* fb = feat_base<FT>(L.src[s0 + v], (int64_t)b * L.sb);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) st.tap[4 * blk + tp] = ld_feat<FT>(fb, st.o[tp] + 16u * blk);
        };
        auto process = [&](int k, const BwdStep<NB>& st, auto scatter) {
            const int v = k / nh;
            float* gb = L.gsrc[s0 + v] + (int64_t)b * L.sb;
            const float* __restrict__ er = e_lds + k * ITERMVS_GROUPS * LS + px;   // row block of (view, hypothesis) k
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int ch = 16 * blk + t;
                const float e = er[(ch / CPG) * LS];
                const float wv = fmaf(st.w[3], st.tap[4 * blk + 3], fmaf(st.w[2], st.tap[4 * blk + 2],
                                      fmaf(st.w[1], st.tap[4 * blk + 1], st.w[0] * st.tap[4 * blk])));
                gacc[blk] = fmaf(e, wv, gacc[blk]);
                if constexpr (!decltype(scatter)::value) continue;      // dL/dref only (the gather kernel below does dL/dsrc)
                const float gs = e * refv[blk];
                // Unconditional: a tap outside the map (weight 0, offset clamped to pixel 0) adds +0 there.  A branch around
                // the atomic would cost more than the rare wasted request: with a data-dependent number of atomics the
                // compiler can only wait for the next step's taps with vmcnt(0), i.e. for every atomic in flight.
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
                    unsafeAtomicAdd(gb + (st.o[tp] + 16u * blk), st.w[tp] != 0.0f ? gs * st.w[tp] : 0.0f);
            }
        };
        auto walk = [&](auto explicit_depth, auto scatter) {
            const int steps = sbc * nh;
            BwdStep<NB> cur, nxt;
            prepare(0, cur, explicit_depth);
#pragma unroll 1
            for (int k = 0; k < steps; ++k) {
                if (k + 1 < steps) prepare(k + 1, nxt, explicit_depth);   // the next step's loads are in flight before this step's atomics
                if constexpr (decltype(scatter)::value == 2) {            // block-uniform: only the routed planes scatter
                    if (routed_lds[k] != 0.0f) process(k, cur, std::true_type{});
                    else process(k, cur, std::false_type{});
                } else {
                    process(k, cur, std::integral_constant<bool, decltype(scatter)::value != 0>{});
                }
                cur = nxt;
            }
        };
        if (ok) {                                        // rows past the last pixel add nothing
            if (L.depth) walk(std::true_type{}, std::integral_constant<int, 1>{});
            else if (a.scatter == 1) walk(std::false_type{}, std::integral_constant<int, 1>{});
            else if (a.scatter == 2) walk(std::false_type{}, std::integral_constant<int, 2>{});
            else walk(std::false_type{}, std::integral_constant<int, 0>{});
        }
        __syncthreads();
    }
    if (ok) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            if (a.init) unsafeAtomicAdd(L.gref + roff + 16 * blk, gacc[blk]);
            else L.gref[roff + 16 * blk] = gacc[blk];
        }
    }
}

template <int FT>
__global__ void __launch_bounds__(kThreads) corr_bwd_kernel(const BwdArgs a) {
    __shared__ float lds[kBwdLdsFloats];
    const BwdLevel& L = a.lv[a.init ? 0 : blockIdx.y];
    switch (L.C) {
        case 16: corr_bwd_level<2, FT>(a, L, lds); break;
        case 32: corr_bwd_level<4, FT>(a, L, lds); break;
        default: corr_bwd_level<6, FT>(a, L, lds); break;
    }
}

// ---- initialisation branch, generated hypotheses: dL/dsrc WITHOUT atomics -------------------------------------------------
// The 32 hypotheses of the initialisation branch are fronto-parallel planes (itermvs.py:11-19: one depth per index, every
// pixel), so for a (view, plane) the warp ref -> src is a homography G_n = K (d_n R D + t e3^T) (R | t = the 3x4 matrix of
// the view, D = diag(W1/W, H1/H, 1), K = the grid_sample un-normalisation) -- and it can be inverted: a source pixel q
// receives bilinear weight only from reference pixels whose image lies in the open square q +- 1, i.e. inside the
// quadrilateral G_n^-1(square).  A ROW of 16 lanes owns one source pixel of one view: per plane it maps the corners of
// the square grown to +-1.02 back to the reference grid, takes the integer bounding box (typically 3x3..4x4 pixels; the
// growth and a +-0.02 pad absorb the rounding of the inverse -- ~1e-4 pixels, G_n^-1 is formed in fp64 --, the projective pre-image of a convex set that does not
// cross the line at infinity is convex), lets lane t FORWARD-project candidate t with exactly the arithmetic of the scatter
// (ray_dir / project_fast / make_taps) and keeps the candidates whose footprint really contains q; the survivors' E * ref
// * weight are summed over channels (lane t = channel 16 blk + t) and planes in registers and stored once: no atomics,
// no collisions, a deterministic result.  Pairs whose vanishing line comes within a pixel of the source image, or whose G_n
// is singular, are not served here at all: plane_inverse routes them to the atomic scatter above (round 3 scanned the whole
// reference grid per source pixel for them -- exact, but O(P1 * N * P) projections for a degenerate camera).  Inside a
// served pair a square whose own corner test still fails (not observed; the pair test carries a 10x margin) falls back to
// that scan for the one pixel.
// The scatter form above (1.07 ms at the cfg-4 shape: 126 M lane-atomics on few, heavily shared addresses) remains for
// explicit per-pixel hypotheses; for generated ones it only computes dL/dref.
template <int CPG, int FT>
__device__ __forceinline__ void init_gather_level(const BwdArgs& a, const BwdLevel& L, float* __restrict__ lds) {
    constexpr int C = 8 * CPG, NB = C / 16;
    const int N = L.N, b = blockIdx.z, s = blockIdx.y;
    const int P = a.H * a.W, P1 = L.H1 * L.W1;
    const int q0 = xcd_tile((P1 + 15) / 16) * 16;
    if (q0 >= P1) return;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float inv_cpg = CPG == 2 ? 0.5f : (CPG == 4 ? 0.25f : 1.0f / 6.0f);
    float* __restrict__ ginv = lds;                        // [N][12]: G_n^-1 row-major, [9] = usable flag
    float* __restrict__ m = lds + 32 * 12;                 // the view's 3x4 matrix
    if (threadIdx.x < 12) m[threadIdx.x] = L.proj[((size_t)b * a.S + s) * 12 + threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < N) {
        float* o = ginv + threadIdx.x * 12;
        o[9] = plane_inverse(m, plane_depth((int)threadIdx.x, N, inv_min, inv_max), g, L.W1, L.H1, o) ? 1.0f : 0.0f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, t = lane & 15, rowbase = lane & 48;
    const int q = q0 + (threadIdx.x >> 4);
    const bool ok = q < P1;
    const int qc = ok ? q : P1 - 1;
    const int v = qc / L.W1, u = qc - v * L.W1;            // the source pixel of this row
    const float* __restrict__ gout = L.gout + ((size_t)b * a.S + s) * N * ITERMVS_GROUPS * P;
    float acc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) acc[blk] = 0.0f;

#pragma unroll 1
    for (int n = 0; n < N; ++n) {
        const float* gi = ginv + n * 12;
        if (gi[9] == 0.0f) continue;                        // routed to the scatter of corr_bwd_kernel (plane_inverse)
        const float d = plane_depth(n, N, inv_min, inv_max);
        // the square's pre-image on the reference grid
        bool reg = true;
        float minx = 3.0e38f, maxx = -3.0e38f, miny = 3.0e38f, maxy = -3.0e38f, w2_first = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float cx = (float)u + ((c & 1) ? 1.02f : -1.02f), cy = (float)v + ((c & 2) ? 1.02f : -1.02f);
            const float w0 = fmaf(gi[0], cx, fmaf(gi[1], cy, gi[2]));
            const float w1 = fmaf(gi[3], cx, fmaf(gi[4], cy, gi[5]));
            const float w2 = fmaf(gi[6], cx, fmaf(gi[7], cy, gi[8]));
            if (c == 0) w2_first = w2;
            reg = reg && (w2 * w2_first > 0.0f) && fabsf(w2) > 1e-6f * (fabsf(w0) + fabsf(w1) + fabsf(w2));
            const float x = w0 / w2, y = w1 / w2;
            minx = fminf(minx, x); maxx = fmaxf(maxx, x); miny = fminf(miny, y); maxy = fmaxf(maxy, y);
        }
        reg = reg && minx > -1.0e6f && maxx < 1.0e6f && miny > -1.0e6f && maxy < 1.0e6f;     // also false for NaN
        int x_lo = 0, x_hi = a.W - 1, y_lo = 0, y_hi = a.H - 1;
        if (reg) {
            x_lo = max(0, (int)floorf(minx - 0.02f)); x_hi = min(a.W - 1, (int)ceilf(maxx + 0.02f));
            y_lo = max(0, (int)floorf(miny - 0.02f)); y_hi = min(a.H - 1, (int)ceilf(maxy + 0.02f));
        }
        const int nx = max(x_hi - x_lo + 1, 0), ny = max(y_hi - y_lo + 1, 0);
        const int nc = ok ? nx * ny : 0;
        for (int k0 = 0; __any(k0 < nc); k0 += 16) {        // candidates, 16 per round and row
            const int idx = k0 + t;
            const int iy_ = idx / max(nx, 1), ix_ = idx - iy_ * max(nx, 1);
            const int cx = x_lo + ix_, cy = min(y_lo + iy_, a.H - 1);
            float rx, ry, rz, sx_, sy_;
            ray_dir(m, (float)cx * g.xr, (float)cy * g.yr, rx, ry, rz);
            project_fast(g, rc, m, rx, ry, rz, d, sx_, sy_);
            const Taps tp = make_taps(sx_, sy_, L.W1, L.H1);
            float wq = 0.0f;                                 // the weight this candidate gives to (u, v); zero-weight taps are clamped to 0
            wq += (tp.x0 == u && tp.y0 == v) ? tp.nw : 0.0f;
            wq += (tp.x1 == u && tp.y0 == v) ? tp.ne : 0.0f;
            wq += (tp.x0 == u && tp.y1 == v) ? tp.sw : 0.0f;
            wq += (tp.x1 == u && tp.y1 == v) ? tp.se : 0.0f;
            if (idx >= nc) wq = 0.0f;
            unsigned rowmask = (unsigned)(__ballot(wq != 0.0f) >> rowbase) & 0xffffu;
            while (__any(rowmask != 0u)) {                   // contributing candidates of the row, one per round
                const bool act = rowmask != 0u;
                const int src_lane = rowbase + (act ? __ffs(rowmask) - 1 : 0);
                rowmask &= rowmask - 1u;
                const float wv = __shfl(wq, src_lane, 64);
                const int px_ = __shfl(cx, src_lane, 64), py_ = __shfl(cy, src_lane, 64);
                if (act) {
                    const int pc = py_ * a.W + px_;
                    const int64_t roff = (int64_t)b * L.rsb + (int64_t)py_ * L.rsy + (int64_t)px_ * L.rsx + t;
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk) {
                        const int ch = 16 * blk + t;
                        const float e = gout[((size_t)n * ITERMVS_GROUPS + ch / CPG) * P + pc] * inv_cpg;
                        const float r = ld_feat<FT>(L.ref, roff + 16 * blk);
                        acc[blk] = fmaf(e * r, wv, acc[blk]);
                    }
                }
            }
        }
    }
    if (ok) {
        float* __restrict__ gb = L.gsrc[s] + (int64_t)b * L.sb + (int64_t)v * L.sy + (int64_t)u * L.sx + t;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) gb[16 * blk] += acc[blk];      // this row is the only writer of the pixel
    }
}

template <int FT>
__global__ void __launch_bounds__(kThreads) init_gather_kernel(const BwdArgs a) {
    __shared__ float lds[33 * 12];
    const BwdLevel& L = a.lv[0];
    switch (L.C) {
        case 16: init_gather_level<2, FT>(a, L, lds); break;
        case 32: init_gather_level<4, FT>(a, L, lds); break;
        default: init_gather_level<6, FT>(a, L, lds); break;
    }
}

}  // namespace itermvs

using namespace itermvs;

static int launch_bwd(const BwdArgs& a, int dtype, dim3 grid, hipStream_t stream) {
    switch (dtype) {
        case ITERMVS_F32: hipLaunchKernelGGL(corr_bwd_kernel<ITERMVS_F32>, grid, dim3(kThreads), 0, stream, a); break;
        case ITERMVS_F16: hipLaunchKernelGGL(corr_bwd_kernel<ITERMVS_F16>, grid, dim3(kThreads), 0, stream, a); break;
        case ITERMVS_BF16: hipLaunchKernelGGL(corr_bwd_kernel<ITERMVS_BF16>, grid, dim3(kThreads), 0, stream, a); break;
        default: return ITERMVS_ERR_DTYPE;
    }
    return itermvs_launch_status();
}

static int launch_gather(const BwdArgs& a, int dtype, dim3 grid, hipStream_t stream) {
    switch (dtype) {
        case ITERMVS_F32: hipLaunchKernelGGL(init_gather_kernel<ITERMVS_F32>, grid, dim3(kThreads), 0, stream, a); break;
        case ITERMVS_F16: hipLaunchKernelGGL(init_gather_kernel<ITERMVS_F16>, grid, dim3(kThreads), 0, stream, a); break;
        case ITERMVS_BF16: hipLaunchKernelGGL(init_gather_kernel<ITERMVS_BF16>, grid, dim3(kThreads), 0, stream, a); break;
        default: return ITERMVS_ERR_DTYPE;
    }
    return itermvs_launch_status();
}

static int fill_level(BwdLevel& L, const itermvs_level_src& s, float* const* gsrc, int S) {
    const int rc = itermvs_check_level(s, S);
    if (rc) return rc;
    for (int v = 0; v < ITERMVS_MAX_SRC; ++v) {
        L.src[v] = (const float*)s.view[v < S ? v : 0];
        L.gsrc[v] = gsrc[v < S ? v : 0];
        ITERMVS_RETURN_IF(!L.gsrc[v], ITERMVS_ERR_NULL);
    }
    L.sb = s.sb; L.sy = s.sy; L.sx = s.sx;
    L.C = s.C; L.H1 = s.H; L.W1 = s.W;
    return ITERMVS_OK;
}

extern "C" int itermvs_corr_iter_backward(const itermvs_corr_iter_params* p, const float* const grad_out[3],
                                          float* const* const grad_src[3], float* grad_ref_q, void* stream) {
    ITERMVS_RETURN_IF(!p || !grad_out || !grad_src || !grad_ref_q, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->ref_q || !p->proj || !p->view_w || !p->inv_depth_min || !p->inv_depth_max, ITERMVS_ERR_NULL);
    // the gradient kernels read the view weights in the contiguous [B,S,H,W] form only
    ITERMVS_RETURN_IF((p->view_w_sb || p->view_w_ss || p->view_w_sp) &&
                      !(p->view_w_sp == 1 && p->view_w_ss == (int64_t)p->H * p->W && p->view_w_sb == (int64_t)p->S * p->H * p->W),
                      ITERMVS_ERR_LAYOUT);
    BwdArgs a;
    const int cq = p->src[0].C + p->src[1].C + p->src[2].C;
    int coff = 0;
    for (int l = 0; l < 3; ++l) {
        BwdLevel& L = a.lv[l];
        ITERMVS_RETURN_IF(!grad_out[l] || !grad_src[l], ITERMVS_ERR_NULL);
        const int rc = fill_level(L, p->src[l], grad_src[l], p->S);
        if (rc) return rc;
        ITERMVS_RETURN_IF(p->N[l] < 1 || p->N[l] > ITERMVS_MAX_HYP, ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(!p->depth[l] && !p->norm_depth, ITERMVS_ERR_NULL);
        L.depth = p->depth[l];
        L.gout = grad_out[l];
        L.ref = p->ref_q + coff;
        L.gref = grad_ref_q + coff;
        L.rsb = (int64_t)p->H * p->W * cq; L.rsy = (int64_t)p->W * cq; L.rsx = cq;
        L.proj = p->proj + (size_t)l * p->B * p->S * 12;
        for (int n = 0; n < ITERMVS_MAX_HYP; ++n) L.offs[n] = p->offsets[l][n];
        L.N = p->N[l];
        coff += L.C;
    }
    a.view_w = p->view_w; a.nd = p->norm_depth; a.nd_sb = p->norm_depth_sb;
    a.inv_min = p->inv_depth_min; a.inv_max = p->inv_depth_max;
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.init = 0; a.vch = 4; a.hyp_chunks = 1; a.scatter = 1;
    const int P = p->H * p->W;
    ITERMVS_RETURN_IF(p->src[1].dtype != p->src[0].dtype || p->src[2].dtype != p->src[0].dtype, ITERMVS_ERR_DTYPE);
    return launch_bwd(a, p->src[0].dtype, dim3((((P + kBwdTile - 1) / kBwdTile + 7) / 8) * 8, 3, p->B), (hipStream_t)stream);
}

extern "C" int itermvs_corr_init_backward(const itermvs_corr_init_params* p, const float* grad_out, float* const* grad_src,
                                          float* grad_ref, void* stream) {
    ITERMVS_RETURN_IF(p && p->out_layout != 0, ITERMVS_ERR_LAYOUT);      // grad_out is [B,S,N,8,H,W]
    ITERMVS_RETURN_IF(!p || !grad_out || !grad_src || !grad_ref, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1 || p->N < 2 || p->N > 32, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->ref.data || !p->proj || !p->inv_depth_min || !p->inv_depth_max, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->ref.sc != 1 || p->ref.C != p->src.C || p->ref.H != p->H || p->ref.W != p->W, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF((p->ref.sx % 4) || (p->ref.sy % 4) || (p->ref.sb % 4) || ((uintptr_t)p->ref.data % 16), ITERMVS_ERR_ALIGN);
    BwdArgs a;
    BwdLevel& L = a.lv[0];
    const int rc = fill_level(L, p->src, grad_src, p->S);
    if (rc) return rc;
    L.depth = p->depth;
    L.gout = grad_out;
    L.ref = (const float*)p->ref.data;
    L.gref = grad_ref;                      // addressed with the strides of `ref`
    L.rsb = p->ref.sb; L.rsy = p->ref.sy; L.rsx = p->ref.sx;
    L.proj = p->proj;
    for (int n = 0; n < ITERMVS_MAX_HYP; ++n) L.offs[n] = 0.0f;
    L.N = p->N;
    a.lv[1] = a.lv[2] = a.lv[0];
    a.view_w = nullptr; a.nd = nullptr; a.nd_sb = 0;
    a.inv_min = p->inv_depth_min; a.inv_max = p->inv_depth_max;
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.init = 1;
    a.vch = 1; a.hyp_chunks = (p->N + 7) / 8;
    // generated hypotheses are planes: dL/dsrc by the atomic-free gather, except the (view, plane) pairs whose vanishing
    // line comes near the source image or whose homography is singular: those are scattered here (mode 2, plane_inverse)
    a.scatter = p->depth ? 1 : 2;
    const int P = p->H * p->W;
    ITERMVS_RETURN_IF(p->ref.dtype != p->src.dtype, ITERMVS_ERR_DTYPE);
    const int rc2 = launch_bwd(a, p->src.dtype, dim3((((P + kBwdTile - 1) / kBwdTile + 7) / 8) * 8, p->S * a.hyp_chunks, p->B), (hipStream_t)stream);
    if (rc2 || a.scatter == 1) return rc2;
    const int P1 = p->src.H * p->src.W;
    return launch_gather(a, p->src.dtype, dim3((((P1 + 15) / 16 + 7) / 8) * 8, p->S, p->B), (hipStream_t)stream);
}
