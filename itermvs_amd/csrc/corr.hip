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
#include "common.hpp"

namespace itermvs {

constexpr int kThreads = 256;

template <int CPG>
struct Chunk {
    static constexpr int VEC = (CPG == 6) ? 6 : 4;  // floats per lane
    static constexpr int LPT = (CPG == 2) ? 4 : 8;  // lanes per tap == C / VEC
    static constexpr int NG = (CPG == 2) ? 2 : 1;   // correlation groups per lane
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const float2 a = reinterpret_cast<const float2*>(p)[0];
        const float2 b = reinterpret_cast<const float2*>(p)[1];
        const float2 c = reinterpret_cast<const float2*>(p)[2];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    }
}

// group correlation of one lane's chunk for one view: bilinear blend of the four taps, product
// with the reference chunk, mean over the channels of each group (itermvs.py:50-51).
template <int CPG>
__device__ __forceinline__ void chunk_corr(const float* __restrict__ fb, int64_t sy, int64_t sx, const Taps& tp,
                                           const float (&refv)[Chunk<CPG>::VEC], float (&corr)[Chunk<CPG>::NG]) {
    constexpr int VEC = Chunk<CPG>::VEC;
    float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
    const float* r0 = fb + tp.y0 * sy;
    const float* r1 = fb + tp.y1 * sy;
    load_vec<VEC>(r0 + tp.x0 * sx, v00);
    load_vec<VEC>(r0 + tp.x1 * sx, v01);
    load_vec<VEC>(r1 + tp.x0 * sx, v10);
    load_vec<VEC>(r1 + tp.x1 * sx, v11);
    float w[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c)
        w[c] = fmaf(tp.se, v11[c], fmaf(tp.sw, v10[c], fmaf(tp.ne, v01[c], tp.nw * v00[c])));
    if constexpr (CPG == 2) {
        corr[0] = fmaf(w[1], refv[1], w[0] * refv[0]) * 0.5f;
        corr[1] = fmaf(w[3], refv[3], w[2] * refv[2]) * 0.5f;
    } else if constexpr (CPG == 4) {
        corr[0] = fmaf(w[3], refv[3], fmaf(w[2], refv[2], fmaf(w[1], refv[1], w[0] * refv[0]))) * 0.25f;
    } else {
        float s = w[0] * refv[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) s = fmaf(w[c], refv[c], s);
        corr[0] = s / 6.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// iteration branch
// ---------------------------------------------------------------------------------------------
struct IterLevel {
    const float* src[ITERMVS_MAX_SRC];
    int64_t sb, sy, sx;
    const float* depth;  // explicit hypotheses or nullptr
    float* out;
    float offs[ITERMVS_MAX_HYP];
    int C, H1, W1, N, coff;
};

struct IterArgs {
    IterLevel lv[3];
    const float* ref_q;
    const float* proj;
    const float* view_w;
    const float* nd;
    int64_t nd_sb;
    const float* inv_min;
    const float* inv_max;
    int B, S, H, W, CQ;
};

template <int CPG, int TILE>
__device__ __forceinline__ void corr_iter_level(const IterArgs& a, const IterLevel& L, int lvl, float* __restrict__ lds) {
    using K = Chunk<CPG>;
    const int N = L.N;
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    const int p0 = blockIdx.x * TILE;
    const int per_px = N * K::LPT;
    const int items = TILE * per_px;
    const WarpGeom g = make_geom(a.W, a.H, L.W1, L.H1);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* proj = a.proj + ((size_t)(lvl * a.B + b) * a.S) * 12;

#pragma unroll 1
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int px = item / per_px;
        const int rem = item - px * per_px;
        const int n = rem / K::LPT;
        const int j = rem - n * K::LPT;
        const int p = p0 + px;
        if (p >= P) continue;
        const int y = p / a.W, x = p - y * a.W;

        float d;
        if (L.depth) {
            d = L.depth[((size_t)b * N + n) * P + p];
        } else {  // itermvs.py:291-293
            float ns = a.nd[b * a.nd_sb + p] + L.offs[n];
            ns = fminf(fmaxf(ns, 0.0f), 1.0f);
            d = unnormalize_depth(ns, inv_min, inv_max);
        }
        float refv[K::VEC];
        load_vec<K::VEC>(a.ref_q + ((size_t)b * P + p) * a.CQ + L.coff + j * K::VEC, refv);

        const float xs = (float)x * g.xr, ys = (float)y * g.yr;
        float acc[K::NG];
#pragma unroll
        for (int q = 0; q < K::NG; ++q) acc[q] = 0.0f;
        float wsum = 1e-5f;  // itermvs.py:88
        const int64_t boff = (int64_t)b * L.sb + j * K::VEC;
        for (int s = 0; s < a.S; ++s) {
            const float* m = proj + s * 12;
            float rx, ry, rz, ix, iy;
            ray_dir(m, xs, ys, rx, ry, rz);
            project(g, m, rx, ry, rz, d, ix, iy, nullptr);
            const Taps tp = make_taps(ix, iy, L.W1, L.H1);
            float corr[K::NG];
            chunk_corr<CPG>(L.src[s] + boff, L.sy, L.sx, tp, refv, corr);
            const float w = a.view_w[((size_t)b * a.S + s) * P + p];
#pragma unroll
            for (int q = 0; q < K::NG; ++q) acc[q] = acc[q] + corr[q] * w;  // itermvs.py:115
            wsum = wsum + w;                                                // itermvs.py:116
        }
#pragma unroll
        for (int q = 0; q < K::NG; ++q) lds[(n * ITERMVS_GROUPS + j * K::NG + q) * TILE + px] = acc[q] / wsum;
    }
    __syncthreads();
    const int rows = N * ITERMVS_GROUPS;
    for (int idx = threadIdx.x; idx < rows * TILE; idx += kThreads) {
        const int row = idx / TILE, px = idx - row * TILE;
        if (p0 + px < P) L.out[((size_t)b * rows + row) * P + p0 + px] = lds[row * TILE + px];
    }
}

template <int TILE>
__global__ void __launch_bounds__(kThreads) corr_iter_kernel(const IterArgs a) {
    __shared__ float lds[ITERMVS_MAX_HYP * ITERMVS_GROUPS * TILE];
    const int lvl = blockIdx.y;
    const IterLevel& L = a.lv[lvl];
    switch (L.C) {
        case 16: corr_iter_level<2, TILE>(a, L, lvl, lds); break;
        case 32: corr_iter_level<4, TILE>(a, L, lvl, lds); break;
        default: corr_iter_level<6, TILE>(a, L, lvl, lds); break;
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

template <int CPG, int TILE>
__device__ __forceinline__ void corr_init_body(const InitArgs& a, float* __restrict__ lds) {
    using K = Chunk<CPG>;
    const int nblocks = (a.N + a.NB - 1) / a.NB;
    const int s = blockIdx.y / nblocks;
    const int n0 = (blockIdx.y - s * nblocks) * a.NB;
    const int nb = min(a.NB, a.N - n0);
    const int b = blockIdx.z;
    const int P = a.H * a.W;
    const int p0 = blockIdx.x * TILE;
    const int per_px = nb * K::LPT;
    const int items = TILE * per_px;
    const WarpGeom g = make_geom(a.W, a.H, a.W1, a.H1);
    const float inv_min = a.inv_min[b], inv_max = a.inv_max[b];
    const float* m = a.proj + ((size_t)b * a.S + s) * 12;
    const float* fsrc = a.src[s] + (int64_t)b * a.sb;

#pragma unroll 1
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int px = item / per_px;
        const int rem = item - px * per_px;
        const int nl = rem / K::LPT;
        const int j = rem - nl * K::LPT;
        const int n = n0 + nl;
        const int p = p0 + px;
        if (p >= P) continue;
        const int y = p / a.W, x = p - y * a.W;
        float d;
        if (a.depth) {
            d = a.depth[((size_t)b * a.N + n) * P + p];
        } else {  // itermvs.py:13-17
            const float frac = (float)n / (float)(a.N - 1);
            d = 1.0f / (inv_max + frac * (inv_min - inv_max));
        }
        float refv[K::VEC];
#pragma unroll
        for (int c = 0; c < K::VEC; ++c)
            refv[c] = a.ref.data[b * a.ref.sb + (j * K::VEC + c) * a.ref.sc + y * a.ref.sy + x * a.ref.sx];
        float rx, ry, rz, ix, iy;
        ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
        project(g, m, rx, ry, rz, d, ix, iy, nullptr);
        const Taps tp = make_taps(ix, iy, a.W1, a.H1);
        float corr[K::NG];
        chunk_corr<CPG>(fsrc + j * K::VEC, a.sy, a.sx, tp, refv, corr);
#pragma unroll
        for (int q = 0; q < K::NG; ++q) lds[(nl * ITERMVS_GROUPS + j * K::NG + q) * TILE + px] = corr[q];
    }
    __syncthreads();
    const int rows = nb * ITERMVS_GROUPS;
    float* o = a.out + (((size_t)b * a.S + s) * a.N + n0) * ITERMVS_GROUPS * P;
    for (int idx = threadIdx.x; idx < rows * TILE; idx += kThreads) {
        const int row = idx / TILE, px = idx - row * TILE;
        if (p0 + px < P) o[(size_t)row * P + p0 + px] = lds[row * TILE + px];
    }
}

constexpr int kInitNB = 8;  // hypotheses per block

template <int TILE>
__global__ void __launch_bounds__(kThreads) corr_init_kernel(const InitArgs a) {
    __shared__ float lds[kInitNB * ITERMVS_GROUPS * TILE];
    switch (a.C) {
        case 16: corr_init_body<2, TILE>(a, lds); break;
        case 32: corr_init_body<4, TILE>(a, lds); break;
        default: corr_init_body<6, TILE>(a, lds); break;
    }
}

// out[b,n,g,p] = sum_s corr[b,s,n,g,p]*w[b,s,p] / (1e-5 + sum_s w[b,s,p])     (itermvs.py:59-69)
__global__ void view_aggregate_kernel(const float* __restrict__ corr, const float* __restrict__ w, int S, int B, int NG,
                                      int P, float* __restrict__ out) {
    const int64_t per = (int64_t)NG * P;
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

}  // namespace itermvs

using namespace itermvs;

static int check_level(const itermvs_level_src& s, int S) {
    ITERMVS_RETURN_IF(s.C != 16 && s.C != 32 && s.C != 48, ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(s.H < 1 || s.W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(s.sc != 1, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF((s.sx % 4) || (s.sy % 4) || (s.sb % 4), ITERMVS_ERR_ALIGN);
    for (int v = 0; v < S; ++v) {
        ITERMVS_RETURN_IF(!s.view[v], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(((uintptr_t)s.view[v]) % 16, ITERMVS_ERR_ALIGN);
    }
    return ITERMVS_OK;
}

extern "C" int itermvs_corr_iter(const itermvs_corr_iter_params* p, void* stream) {
    ITERMVS_RETURN_IF(!p, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->ref_q || !p->proj || !p->view_w || !p->inv_depth_min || !p->inv_depth_max, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(((uintptr_t)p->ref_q) % 16, ITERMVS_ERR_ALIGN);
    IterArgs a;
    int coff = 0;
    for (int l = 0; l < 3; ++l) {
        const int rc = check_level(p->src[l], p->S);
        if (rc) return rc;
        ITERMVS_RETURN_IF(p->N[l] < 1 || p->N[l] > ITERMVS_MAX_HYP, ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(!p->out[l], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(!p->depth[l] && !p->norm_depth, ITERMVS_ERR_NULL);
        IterLevel& L = a.lv[l];
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
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.CQ = coff;
    constexpr int TILE = 32;
    const int P = p->H * p->W;
    itermvs_profile_begin(1, (hipStream_t)stream);
    hipLaunchKernelGGL(corr_iter_kernel<TILE>, dim3((P + TILE - 1) / TILE, 3, p->B), dim3(kThreads), 0,
                       (hipStream_t)stream, a);
    itermvs_profile_end(1, (hipStream_t)stream);
    return itermvs_launch_status();
}

extern "C" int itermvs_corr_init(const itermvs_corr_init_params* p, void* stream) {
    ITERMVS_RETURN_IF(!p, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->B < 1 || p->H < 1 || p->W < 1 || p->N < 2, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->S < 1 || p->S > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    ITERMVS_RETURN_IF(!p->ref.data || !p->proj || !p->inv_depth_min || !p->inv_depth_max || !p->out, ITERMVS_ERR_NULL);
    const int rc = check_level(p->src, p->S);
    if (rc) return rc;
    ITERMVS_RETURN_IF(p->ref.C != p->src.C || p->ref.H != p->H || p->ref.W != p->W, ITERMVS_ERR_DIMS);
    InitArgs a;
    for (int v = 0; v < ITERMVS_MAX_SRC; ++v) a.src[v] = p->src.view[v < p->S ? v : 0];
    a.sb = p->src.sb; a.sy = p->src.sy; a.sx = p->src.sx;
    a.ref = p->ref; a.proj = p->proj; a.depth = p->depth;
    a.inv_min = p->inv_depth_min; a.inv_max = p->inv_depth_max; a.out = p->out;
    a.B = p->B; a.S = p->S; a.H = p->H; a.W = p->W; a.N = p->N;
    a.C = p->src.C; a.H1 = p->src.H; a.W1 = p->src.W; a.NB = kInitNB;
    constexpr int TILE = 32;
    const int P = p->H * p->W;
    const int nblocks = (p->N + kInitNB - 1) / kInitNB;
    itermvs_profile_begin(2, (hipStream_t)stream);
    hipLaunchKernelGGL(corr_init_kernel<TILE>, dim3((P + TILE - 1) / TILE, p->S * nblocks, p->B), dim3(kThreads), 0,
                       (hipStream_t)stream, a);
    itermvs_profile_end(2, (hipStream_t)stream);
    return itermvs_launch_status();
}

extern "C" int itermvs_view_aggregate(const float* corr, const float* w, int32_t S, int32_t B, int32_t N, int32_t P,
                                      float* out, void* stream) {
    ITERMVS_RETURN_IF(!corr || !w || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(S < 1 || B < 1 || N < 1 || P < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * N * ITERMVS_GROUPS * P;
    hipLaunchKernelGGL(view_aggregate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       corr, w, S, B, N * ITERMVS_GROUPS, P, out);
    return itermvs_launch_status();
}

extern "C" int itermvs_softmax_max(const float* x, int32_t M, int32_t N, int32_t P, float* out, void* stream) {
    ITERMVS_RETURN_IF(!x || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(M < 1 || N < 1 || P < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)M * P;
    hipLaunchKernelGGL(softmax_max_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M,
                       N, P, out);
    return itermvs_launch_status();
}
