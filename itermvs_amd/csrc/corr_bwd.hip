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
};

constexpr int kBwdTile = 16;                                           // pixels per block = rows of 16 lanes
constexpr int kBwdLS = kBwdTile + 1;
constexpr int kBwdRows = 4 * ITERMVS_MAX_HYP * ITERMVS_GROUPS;         // staged E rows: (views) x (hypotheses) x 8 <= 256
constexpr int kBwdLdsFloats = kBwdRows * kBwdLS + kBwdTile + ITERMVS_MAX_SRC * 12;   // + wsum [px] + the views' 3x4 matrices

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
                const float frac = (float)n / (float)(N - 1);
                d = 1.0f / (inv_max + frac * (inv_min - inv_max));
            } else {               // itermvs.py:291-293
                float ns = ndv + L.offs[n];
                ns = fminf(fmaxf(ns, 0.0f), 1.0f);
                d = unnormalize_depth(ns, inv_min, inv_max);
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
        auto process = [&](int k, const BwdStep<NB>& st) {
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
                const float gs = e * refv[blk];
                // Unconditional: a tap outside the map (weight 0, offset clamped to pixel 0) adds +0 there.  A branch around
                // the atomic would cost more than the rare wasted request: with a data-dependent number of atomics the
                // compiler can only wait for the next step's taps with vmcnt(0), i.e. for every atomic in flight.
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
                    unsafeAtomicAdd(gb + (st.o[tp] + 16u * blk), st.w[tp] != 0.0f ? gs * st.w[tp] : 0.0f);
            }
        };
        auto walk = [&](auto explicit_depth) {
            const int steps = sbc * nh;
            BwdStep<NB> cur, nxt;
            prepare(0, cur, explicit_depth);
#pragma unroll 1
            for (int k = 0; k < steps; ++k) {
                if (k + 1 < steps) prepare(k + 1, nxt, explicit_depth);   // the next step's loads are in flight before this step's atomics
                process(k, cur);
                cur = nxt;
            }
        };
        if (ok) {                                        // rows past the last pixel add nothing
            if (L.depth) walk(std::true_type{});
            else walk(std::false_type{});
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
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.init = 0; a.vch = 4; a.hyp_chunks = 1;
    const int P = p->H * p->W;
    ITERMVS_RETURN_IF(p->src[1].dtype != p->src[0].dtype || p->src[2].dtype != p->src[0].dtype, ITERMVS_ERR_DTYPE);
    return launch_bwd(a, p->src[0].dtype, dim3((((P + kBwdTile - 1) / kBwdTile + 7) / 8) * 8, 3, p->B), (hipStream_t)stream);
}

extern "C" int itermvs_corr_init_backward(const itermvs_corr_init_params* p, const float* grad_out, float* const* grad_src,
                                          float* grad_ref, void* stream) {
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
    const int P = p->H * p->W;
    ITERMVS_RETURN_IF(p->ref.dtype != p->src.dtype, ITERMVS_ERR_DTYPE);
    return launch_bwd(a, p->src.dtype, dim3((((P + kBwdTile - 1) / kBwdTile + 7) / 8) * 8, p->S * a.hyp_chunks, p->B), (hipStream_t)stream);
}
