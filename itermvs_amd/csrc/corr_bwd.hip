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
// Decomposition = the forward's "views across waves" form (corr.hip): a quad owns one (pixel, view); quad lane u projects
// hypothesis hb+u once and the quad shares the footprint with DPP broadcasts; E is staged in LDS per chunk of views so every
// lane can pick the group of each of its channels; dL/dref is accumulated in LDS (ds_add_f32) and stored once per block.
#include "corr_common.hpp"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace itermvs {

struct BwdLevel {
    const float* src[ITERMVS_MAX_SRC];
    float* gsrc[ITERMVS_MAX_SRC];   // same strides as src, zero-filled by the caller
    int64_t sb, sy, sx;
    const float* depth;             // explicit hypotheses [B,N,P] or nullptr
    const float* gout;              // iteration: [B,N,8,P]; initialisation: [B,S,N,8,P]
    const float* ref;               // reference features of this level, element (b, c, y, x) at b*rsb + c + y*rsy + x*rsx
    float* gref;                    // same addressing as ref, written (not accumulated)
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
    int vch;                        // views per chunk: 4 (N <= 8), 2 (N <= 16), 1 (N <= 32)
};

constexpr int kBwdLdsE = 4 * ITERMVS_MAX_HYP * ITERMVS_GROUPS * (kVwTile + 1);   // staged E: vch * N * 8 rows
constexpr int kBwdLdsFloats = kBwdLdsE + kVwTile * 49 + kVwTile;                   // + dL/dref [px][C+1] + wsum [px]

template <int CPG, int FT>
__device__ __forceinline__ void corr_bwd_level(const BwdArgs& a, const BwdLevel& L, float* __restrict__ lds) {
    using K = VwChunk<CPG>;
    constexpr int TILE = kVwTile, LS = TILE + 1, C = 8 * CPG, CS = C + 1;
    const int N = L.N, rows = N * ITERMVS_GROUPS;
    float* __restrict__ e_lds = lds;                       // [vch][rows][LS]
    float* __restrict__ gref_lds = lds + kBwdLdsE;         // [TILE][CS]
    float* __restrict__ wsum_lds = gref_lds + TILE * 49;   // [TILE]
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    const int tile = xcd_tile((P + TILE - 1) / TILE);
    const int p0 = tile * TILE;
    if (p0 >= P) return;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const WarpRcp rc = make_rcp(g);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = L.proj + (size_t)b * a.S * 12;
    const uint32_t sy = (uint32_t)L.sy, sx = (uint32_t)L.sx;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 3;
    const uint32_t joff = (uint32_t)(j * 4);
    const float inv_cpg = CPG == 2 ? 0.5f : (CPG == 4 ? 0.25f : 1.0f / 6.0f);

    for (int i = threadIdx.x; i < TILE * CS; i += kThreads) gref_lds[i] = 0.0f;
    if (threadIdx.x < TILE) {
        float ws = 1e-5f;   // itermvs.py:88
        const int p = p0 + threadIdx.x;
        if (a.view_w && p < P)
            for (int s = 0; s < a.S; ++s) ws = ws + a.view_w[((size_t)b * a.S + s) * P + p];
        wsum_lds[threadIdx.x] = ws;
    }
    __syncthreads();

    const int hbn = (N + 3) / 4;   // hypothesis batches of 4
    for (int s0 = 0; s0 < a.S; s0 += a.vch) {
        const int sbc = min(a.vch, a.S - s0);
        // E of this chunk -> LDS (already divided by the channels per group)
        for (int i = threadIdx.x; i < sbc * rows * TILE; i += kThreads) {
            const int px = i % TILE, r = (i / TILE) % rows, v = i / (TILE * rows);
            const int p = p0 + px;
            float e = 0.0f;
            if (p < P) {
                if (a.init) {
                    e = L.gout[(((size_t)b * a.S + s0 + v) * rows + r) * P + p] * inv_cpg;
                } else {
                    const float w = a.view_w[((size_t)b * a.S + s0 + v) * P + p];
                    e = L.gout[((size_t)b * rows + r) * P + p] * (w / wsum_lds[px]) * inv_cpg;
                }
            }
            e_lds[(v * rows + r) * LS + px] = e;
        }
        __syncthreads();
#pragma unroll 1
        for (int r = wave; r < sbc * hbn * 2; r += kThreads / 64) {
            const int pb = r & 1, hbi = (r >> 1) % hbn, v = (r >> 1) / hbn;   // wave-uniform
            const float* fb = feat_base<FT>(L.src[s0 + v], (int64_t)b * L.sb);
            float* gb = L.gsrc[s0 + v] + (int64_t)b * L.sb;
            const float* m = proj + (s0 + v) * 12;
            const int px = pb * 16 + (lane >> 2);
            const int p = p0 + px;
            if (p < P) {   // whole quads drop out together
                const int y = p / a.W, x = p - y * a.W;
                float refv[K::VEC];
                // the reference features: the fp32 ref_q pack (iteration) or the level-3 features themselves (initialisation)
                const int64_t roff = (int64_t)b * L.rsb + (int64_t)y * L.rsy + (int64_t)x * L.rsx + j * 4;
                if (a.init) load_feat<K::VEC, FT>(feat_base<FT>(L.ref, roff), 0u, refv);
                else load_vec<K::VEC>(L.ref + roff, refv);
                float rx, ry, rz;
                ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
                const int hb = hbi * 4;
                const int n_own = min(hb + j, N - 1);
                float d;
                if (L.depth) {
                    d = L.depth[((size_t)b * N + n_own) * P + p];
                } else if (a.init) {   // itermvs.py:13-17
                    const float frac = (float)n_own / (float)(N - 1);
                    d = 1.0f / (inv_max + frac * (inv_min - inv_max));
                } else {               // itermvs.py:291-293
                    float off = L.offs[0];
#pragma unroll
                    for (int k = 1; k < ITERMVS_MAX_HYP; ++k) off = (n_own == k) ? L.offs[k] : off;
                    float ns = a.nd[b * a.nd_sb + p] + off;
                    ns = fminf(fmaxf(ns, 0.0f), 1.0f);
                    d = unnormalize_depth(ns, inv_min, inv_max);
                }
                float ix, iy;
                project_fast(g, rc, m, rx, ry, rz, d, ix, iy);
                const Footprint f = make_footprint(ix, iy, L.W1, L.H1, sy, sx);
                const uint32_t o00 = f.r0 + f.c0, o01 = f.r0 + f.c1, o10 = f.r1 + f.c0, o11 = f.r1 + f.c1;
                float gacc[K::VEC];
#pragma unroll
                for (int c = 0; c < K::VEC; ++c) gacc[c] = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (hb + u < N) {   // uniform
                        const uint32_t q00 = quad_bcast(o00, u) + joff, q01 = quad_bcast(o01, u) + joff;
                        const uint32_t q10 = quad_bcast(o10, u) + joff, q11 = quad_bcast(o11, u) + joff;
                        const float nw = quad_bcast(f.nw, u), ne = quad_bcast(f.ne, u), sw = quad_bcast(f.sw, u), se = quad_bcast(f.se, u);
                        TapData<K::VEC> t;
                        load_feat<K::VEC, FT>(fb, q00, t.v00);
                        load_feat<K::VEC, FT>(fb, q01, t.v01);
                        load_feat<K::VEC, FT>(fb, q10, t.v10);
                        load_feat<K::VEC, FT>(fb, q11, t.v11);
                        const float* __restrict__ er = e_lds + (v * rows + (hb + u) * ITERMVS_GROUPS) * LS + px;
#pragma unroll
                        for (int c = 0; c < K::VEC; ++c) {
                            const int ch = 16 * (c / 4) + 4 * j + (c % 4);      // channel of element c (chunk layout of corr.hip)
                            const float e = er[(ch / CPG) * LS];
                            const float wv = fmaf(se, t.v11[c], fmaf(sw, t.v10[c], fmaf(ne, t.v01[c], nw * t.v00[c])));
                            gacc[c] = fmaf(e, wv, gacc[c]);
                            const float gs = e * refv[c];
                            const uint32_t co = (uint32_t)(16 * (c / 4) + (c % 4));
                            // taps outside the map carry weight 0 (their offsets were clamped to 0): nothing to add
                            if (nw != 0.0f) unsafeAtomicAdd(gb + (q00 + co), gs * nw);
                            if (ne != 0.0f) unsafeAtomicAdd(gb + (q01 + co), gs * ne);
                            if (sw != 0.0f) unsafeAtomicAdd(gb + (q10 + co), gs * sw);
                            if (se != 0.0f) unsafeAtomicAdd(gb + (q11 + co), gs * se);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < K::VEC; ++c)
                    atomicAdd(&gref_lds[px * CS + 16 * (c / 4) + 4 * j + (c % 4)], gacc[c]);
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < TILE * C; i += kThreads) {
        const int px = i / C, c = i - px * C;
        const int p = p0 + px;
        if (p < P) {
            const int y = p / a.W, x = p - y * a.W;
            L.gref[(int64_t)b * L.rsb + (int64_t)y * L.rsy + (int64_t)x * L.rsx + c] = gref_lds[px * CS + c];
        }
    }
}

template <int FT>
__global__ void __launch_bounds__(kThreads) corr_bwd_kernel(const BwdArgs a) {
    __shared__ float lds[kBwdLdsFloats];
    const BwdLevel& L = a.lv[blockIdx.y];
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
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.init = 0; a.vch = 4;
    const int P = p->H * p->W;
    ITERMVS_RETURN_IF(p->src[1].dtype != p->src[0].dtype || p->src[2].dtype != p->src[0].dtype, ITERMVS_ERR_DTYPE);
    return launch_bwd(a, p->src[0].dtype, dim3((((P + kVwTile - 1) / kVwTile + 7) / 8) * 8, 3, p->B), (hipStream_t)stream);
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
    a.vch = p->N <= 8 ? 4 : (p->N <= 16 ? 2 : 1);
    const int P = p->H * p->W;
    ITERMVS_RETURN_IF(p->ref.dtype != p->src.dtype, ITERMVS_ERR_DTYPE);
    return launch_bwd(a, p->src.dtype, dim3((((P + kVwTile - 1) / kVwTile + 7) / 8) * 8, 1, p->B), (hipStream_t)stream);
}
