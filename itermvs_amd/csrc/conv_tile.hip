// itermvs_conv2d, matrix-core path for the 3x3 convolutions: LDS-tiled implicit GEMM on
// v_mfma_f32_16x16x4_f32 (weight_format 2).
//
// The direct-gather kernel (conv_mfma.hip) re-fetches every input element once per tap: nine times
// through L2 -> L1 in 64-byte pieces of 128-byte lines, which -- not the matrix cores, not HBM -- bounds
// it (rocprofv3: 35-50 TFLOP/s on the FeatureNet layers against a ~130 TFLOP/s practical fp32 MFMA
// peak).  Here a workgroup (4 waves) owns a TH x 16*TWT tile of output pixels and, per chunk of 4*S
// input channels, copies the input tile + halo into LDS once (zero padding and channel padding come from
// the buffer descriptor's bounds check); all nine taps then read their B operands from LDS.
//
// K ordering inside a chunk: k-slot q (= lane >> 4) of MFMA step s holds channel  c0 + q*S + s, so the
// S values a lane needs across the S steps of a chunk are CONTIGUOUS:
//   * LDS tile layout  [q][pixel][S]            -> one ds_read_b32/b64/b128 per (tap, 16-pixel segment);
//   * weight layout    [tap][chunk][s][q][cout] -> S coalesced buffer_load_dword per (tap, 16 channels);
// and every LDS address is `lane base + compile-time offset` (tile geometry, stride and dilation are
// template parameters), every weight address `lane base + SGPR offset`: no vector address math in the
// k-loop.  S = 1 / 2 / 4 for Cin <= 4 / <= 8 / larger (3, 8 and 16..64 input channels on the path).
// The q-plane stride is padded so that a wave's read hits each bank once (MI355X_MICROARCH.md, LDS:
// b128 -> 64 banks, 16-lane groups mixing two q's; b64 -> 64 banks / 32 lanes; b32 -> 32 banks / 32 lanes).
#include <stdlib.h>

#include "common.hpp"
#include "conv_epilogue.hpp"

namespace itermvs {

using f32x2 = __attribute__((ext_vector_type(2))) float;

struct TileArgs {
    const float* in;
    float* out;
    float* out2;
    const float* add;
    const float* aux1;
    const float* aux2;
    int64_t in_sn, out_sn, add_sn, aux1_sn, aux2_sn;
    const float* weight[3];   // packed [9][nchunk][S][4][CoutPad]
    const float* bias[3];
    int seg_end[3];
    int N, Cin, Hin, Win, Cout, CoutPad, Hout, Wout;
    int pad, act, nchunk, tiles_x, tiles_y, ncb, total;
    uint32_t rcp_tiles_x, rcp_tiles_y;   // floor(2^32 / d) + 1
};

constexpr uint32_t kTileOob = 0x7fffffffu;

// -DITERMVS_TILE_TRACE (tools/ubench/conv_tile_trace.hip): wave 0 of workgroup 0 stamps the phases of its tiles
#ifdef ITERMVS_TILE_TRACE
__device__ unsigned long long g_tile_trace[8 * 64];
#define TILE_STAMP(slot)                                                                        \
    do {                                                                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0 && trace_tile < 64)                              \
            g_tile_trace[trace_tile * 8 + (slot)] = __builtin_readcyclecounter();                \
    } while (0)
#else
#define TILE_STAMP(slot)
#endif

template <int S> struct Vec;
template <> struct Vec<1> { float v[1]; };
template <> struct Vec<2> { float v[2]; };
template <> struct Vec<4> { float v[4]; };

// weights of one (tap, chunk, 16 output channels): S dword loads, each a fully coalesced 256-byte row
// [q][cout] (this compiler lowers the b64/b128 raw-buffer builtins to a single dword load, so the wide
// forms are not used)
template <int S>
__device__ __forceinline__ Vec<S> wload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint32_t sstep_b) {
    Vec<S> o;
#pragma unroll
    for (int s = 0; s < S; ++s)
        o.v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff + s * sstep_b, 0));
    return o;
}

template <int S>
__device__ __forceinline__ Vec<S> lds_read(const float* p) {
    Vec<S> o;
    if constexpr (S == 1) {
        o.v[0] = *p;
    } else if constexpr (S == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p);
        o.v[0] = t[0];
        o.v[1] = t[1];
    } else {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[i] = t[i];
    }
    return o;
}

template <int S>
__device__ __forceinline__ void lds_write(float* p, const Vec<S>& x) {
    if constexpr (S == 1) {
        *p = x.v[0];
    } else if constexpr (S == 2) {
        *reinterpret_cast<f32x2*>(p) = f32x2{x.v[0], x.v[1]};
    } else {
        *reinterpret_cast<f32x4*>(p) = f32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
    }
}

template <int S, int STRIDE, int DIL, int TH, int TWT>
struct TileGeom {
    static constexpr int TW = 16 * TWT;
    static constexpr int NB = TH * TWT / 4;                          // 16-pixel segments per wave
    static constexpr int IN_H = (TH - 1) * STRIDE + 2 * DIL + 1;
    static constexpr int IN_W = (TW - 1) * STRIDE + 2 * DIL + 1;
    static constexpr int IN_PX = IN_H * IN_W;
    static constexpr int PL = (IN_PX * S + 63) / 64 * 64 + (S == 4 ? 0 : S == 2 ? 32 : 16);   // floats per q-plane
    static constexpr int ITEMS = (4 * IN_PX + 255) / 256;            // (q, pixel) staging items per thread
    static_assert(TH * TWT % 4 == 0, "segments must divide over 4 waves");
    static_assert(NB % TWT == 0 || TWT % NB == 0, "a wave's segments must form whole rows or a row part");
};

template <int MB, int S, int STRIDE, int DIL, int TH, int TWT>
__global__ void __launch_bounds__(256) conv_tile_kernel(const TileArgs a) {
    using G = TileGeom<S, STRIDE, DIL, TH, TWT>;
    constexpr int NB = G::NB;
    __shared__ __attribute__((aligned(16))) float tile[4 * G::PL];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
    const int m0 = blockIdx.y * (MB * 16);          // the workgroup's channel block is fixed: weights and bias
                                                    // are fetched once and stay in registers across its tiles
    // work item w -> (tile column, tile row, batch item)
    struct Work { int n, oy0, ox0; };
    auto decode = [&](int w) {
        // exact floor divisions by scalar multiply-high with host-side reciprocals (w * divisor < 2^32)
        Work k;
        const int t2 = a.tiles_x == 1 ? w : (int)__umulhi((uint32_t)w, a.rcp_tiles_x);
        const int tx = w - t2 * a.tiles_x;
        k.n = a.tiles_y == 1 ? t2 : (int)__umulhi((uint32_t)t2, a.rcp_tiles_y);
        const int ty = t2 - k.n * a.tiles_y;
        k.oy0 = ty * TH;
        k.ox0 = tx * G::TW;
        return k;
    };

    // staging items of this thread: item = (q', pixel of the input tile).  Tile-relative values are fixed:
    // the byte offset of channel q'*S at that pixel relative to the tile origin, and the LDS float index.
    int rel[G::ITEMS], loff[G::ITEMS];
    uint32_t reloff[G::ITEMS];
#pragma unroll
    for (int j = 0; j < G::ITEMS; ++j) {
        const int item = tid + j * 256;
        const int iq = item / G::IN_PX;
        const int px = item - iq * G::IN_PX;
        const int y = px / G::IN_W, x = px - y * G::IN_W;
        const bool live = item < 4 * G::IN_PX;
        rel[j] = live ? (y << 12) | x : -1;
        reloff[j] = live ? ((uint32_t)(iq * S) * plane + (uint32_t)(y * a.Win + x)) * 4u : kTileOob;
        loff[j] = iq * G::PL + px * S;
    }
    uint32_t goff[G::ITEMS];
    __amdgpu_buffer_rsrc_t ir;
    auto setup = [&](const Work& k) {
        ir = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (int64_t)k.n * a.in_sn), 0, (int)(a.Cin * plane * 4u), 0x00020000);
        const int iy0 = k.oy0 * STRIDE - a.pad, ix0 = k.ox0 * STRIDE - a.pad;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + G::IN_H <= a.Hin && ix0 + G::IN_W <= a.Win) {
            // interior tile (uniform): one add per item
            const uint32_t base = (uint32_t)(iy0 * a.Win + ix0) * 4u;
#pragma unroll
            for (int j = 0; j < G::ITEMS; ++j) goff[j] = reloff[j] + base;
        } else {
            const int base = (iy0 * a.Win + ix0) * 4;
#pragma unroll
            for (int j = 0; j < G::ITEMS; ++j) {
                const int gy = iy0 + (rel[j] >> 12), gx = ix0 + (rel[j] & 0xfff);
                const bool ok = rel[j] >= 0 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
                goff[j] = ok ? reloff[j] + (uint32_t)base : kTileOob;
            }
        }
    };
    Vec<S> stage[G::ITEMS];
    const uint32_t chunk_b = 4u * S * plane * 4u;   // bytes between chunks in the input planes
    // the ITEMS*S dword loads of one stage, split into nine parts: part t is issued inside tap t of the
    // previous stage's MFMA loop.  (Issued in one burst the loads fill the CU's vector-memory queue and the
    // wave sits in the issue of the 24 loads for 2-5k cycles -- as long as the whole MFMA phase -- before
    // its first MFMA.)
    constexpr int kLoads = G::ITEMS * S;
    auto fetch_part = [&](uint32_t soff, int part) {
#pragma unroll
        for (int e = 0; e < kLoads; ++e)
            if (e * 9 / kLoads == part) {
                const int j = e / S, s2 = e % S;
                stage[j].v[s2] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(ir, goff[j], soff + s2 * plane * 4u, 0));
            }
    };
    auto fetch = [&](int ch) {
#pragma unroll
        for (int part = 0; part < 9; ++part) fetch_part(ch * chunk_b, part);
    };

    // this wave's segments: id = wave * NB + nb -> (row, column block) of the output tile
    const int seg0 = wave * NB;
    const int row0 = seg0 / TWT, col0 = seg0 - row0 * TWT;          // NB >= TWT: col0 == 0
    const float* __restrict__ bbase = tile + q * G::PL + ((row0 * STRIDE) * G::IN_W + (col0 * 16 + l16) * STRIDE) * S;
    const uint32_t wv = (uint32_t)(q * a.CoutPad + l16) * 4u;
    const uint32_t wstep_b = 4u * a.CoutPad * 4u;                  // bytes per MFMA step ([q][cout] row block)
    const uint32_t wchunk_b = S * wstep_b;                          // bytes per (tap, chunk) of the packed weights
    const int P = a.Hout * a.Wout;

    // Persistent workgroups: the grid is about two workgroups per CU and each walks the tile list with
    // stride gridDim.x.  The next (tile, chunk) is fetched into registers while the matrix cores work on
    // the current one.  vmcnt retires in order, so this chunk's weights are requested BEFORE the prefetch
    // (the MFMAs then wait only for the older weight loads); with a single chunk (Cin <= 16) the weights
    // are loaded once per workgroup.
    Vec<S> av[9][MB];
    int w = blockIdx.x;
#ifdef ITERMVS_TILE_TRACE
    int trace_tile = 0;
#endif
    int wseg = -1;
    Work cur = decode(w);
    setup(cur);
    fetch(0);
    while (true) {
        const int seg = (cur.n >= a.seg_end[0]) + (cur.n >= a.seg_end[1]);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.weight[seg] + m0), 0, (int)(9u * a.nchunk * S * wstep_b), 0x00020000);
        f32x4 acc[MB][NB];
        conv_bias_init<MB, NB>(acc, a.bias[seg], a.Cout, m0, q);
        const int wn = w + gridDim.x;
        Work nxt = cur;
        for (int ch = 0; ch < a.nchunk; ++ch) {
            TILE_STAMP(0);
            __syncthreads();                        // the previous stage's LDS reads are done
#pragma unroll
            for (int j = 0; j < G::ITEMS; ++j)
                if (j < G::ITEMS - 1 || tid + j * 256 < 4 * G::IN_PX) lds_write<S>(tile + loff[j], stage[j]);
            __syncthreads();
            TILE_STAMP(1);
            if (a.nchunk > 1 || seg != wseg) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        av[tap][mb] = wload<S>(wr, wv + mb * 64, (uint32_t)(tap * a.nchunk + ch) * wchunk_b, wstep_b);
                wseg = seg;
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next stage: the following chunk of this tile, or chunk 0 of the workgroup's next tile
            bool prefetch = true;
            uint32_t pf_soff = 0;
            if (ch + 1 < a.nchunk) {
                pf_soff = (uint32_t)(ch + 1) * chunk_b;
            } else if (wn < a.total) {
                nxt = decode(wn);
                setup(nxt);
            } else {
                prefetch = false;
            }
            TILE_STAMP(2);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                Vec<S> bv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int r = (nb / TWT), c = nb % TWT;     // relative to (row0, col0)
                    bv[nb] = lds_read<S>(bbase + ((r * STRIDE + ky * DIL) * G::IN_W + c * 16 * STRIDE + kx * DIL) * S);
                }
                if (prefetch) fetch_part(pf_soff, tap);
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap][mb].v[s], bv[nb].v[s], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        TILE_STAMP(3);
        // D: col (pixel) = lane & 15, row (cout) = (lane >> 4) * 4 + r
        uint32_t pix_off[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int oy = cur.oy0 + row0 + nb / TWT, ox = cur.ox0 + (col0 + nb % TWT) * 16 + l16;
            pix_off[nb] = oy < a.Hout && ox < a.Wout ? (uint32_t)(oy * a.Wout + ox) * 4u : kEpiOob;
        }
        EpilogueArgs e;
        e.out = a.out + (int64_t)cur.n * a.out_sn;
        e.out2 = a.out2 ? a.out2 + (int64_t)cur.n * a.Cout * P : nullptr;
        e.add = a.add ? a.add + (int64_t)cur.n * a.add_sn : nullptr;
        e.aux1 = a.aux1 ? a.aux1 + (int64_t)cur.n * a.aux1_sn : nullptr;
        e.aux2 = a.aux2 ? a.aux2 + (int64_t)cur.n * a.aux2_sn : nullptr;
        e.Cout = a.Cout; e.P = P; e.act = a.act;
        conv_epilogue<MB, NB>(e, acc, m0, q, pix_off);
        TILE_STAMP(4);
#ifdef ITERMVS_TILE_TRACE
        ++trace_tile;
#endif
        if (wn >= a.total) break;
        w = wn;
        cur = nxt;
    }
}

template <int MB, int S, int STRIDE, int DIL, int TH, int TWT>
static void launch_tile(TileArgs& a, int mt, hipStream_t stream) {
    constexpr int TW = 16 * TWT;
    a.tiles_x = (a.Wout + TW - 1) / TW;
    a.tiles_y = (a.Hout + TH - 1) / TH;
    a.ncb = mt / MB;
    a.total = a.N * a.tiles_y * a.tiles_x;
    a.rcp_tiles_x = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_x + 1);   // (unused when the divisor is 1)
    a.rcp_tiles_y = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_y + 1);
    // persistent grid: about ITERMVS_TILE_PERSIST (default 2) workgroups per CU in total, each walking the
    // tile list of its channel block (one tile each when there are fewer tiles than that)
    static const int per_cu = [] {
        const char* e = getenv("ITERMVS_TILE_PERSIST");
        int want = e ? atoi(e) : 2, fit = 1;
        // never more workgroups than are resident at once: a persistent workgroup queued behind another
        // one would serialise its whole tile list
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, conv_tile_kernel<MB, S, STRIDE, DIL, TH, TWT>, 256, 0) != hipSuccess)
            fit = 1;
        if (want < 1) want = 2;
        return want < fit ? want : (fit < 1 ? 1 : fit);
    }();
    int gx = 256 * per_cu / a.ncb;
    if (gx > a.total) gx = a.total;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, a.ncb);
    hipLaunchKernelGGL((conv_tile_kernel<MB, S, STRIDE, DIL, TH, TWT>), grid, dim3(256), 0, stream, a);
}

// tile shapes: big = 8 x 32 pixels (4 segments per wave), mid = 4 x 32 (2), small = 4 x 16 (1)
template <int MB, int S, int STRIDE, int DIL>
static void launch_shape(TileArgs& a, int mt, int shape, hipStream_t stream) {
    if (shape == 2) {
        if constexpr (STRIDE == 1) launch_tile<MB, S, STRIDE, DIL, 8, 2>(a, mt, stream);
        else launch_tile<MB, S, STRIDE, DIL, 4, 2>(a, mt, stream);     // stride 2: the 8x32 halo tile would not fit
    } else if (shape == 1) {
        launch_tile<MB, S, STRIDE, DIL, 4, 2>(a, mt, stream);
    } else {
        launch_tile<MB, S, STRIDE, DIL, 4, 1>(a, mt, stream);
    }
}

template <int S, int STRIDE, int DIL>
static void launch_mb(TileArgs& a, int mt, int mb, int shape, hipStream_t stream) {
    if (mb == 2) launch_shape<2, S, STRIDE, DIL>(a, mt, shape, stream);
    else launch_shape<1, S, STRIDE, DIL>(a, mt, shape, stream);
}

}  // namespace itermvs

using namespace itermvs;

// called from itermvs_conv2d (conv.hip) when weight_format == 2; returns 1 when the shape is not covered
// (the caller reports ITERMVS_ERR_DIMS: format-2 weights cannot feed another kernel)
int itermvs_conv2d_tile(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream) {
    if (p->ksize != 3) return 1;
    const bool s1d1 = p->stride == 1 && p->dilation == 1, s2d1 = p->stride == 2 && p->dilation == 1;
    const bool s1d2 = p->stride == 1 && p->dilation == 2;
    if (!s1d1 && !s2d1 && !s1d2) return 1;
    TileArgs a;
    a.in = p->in; a.out = p->out; a.out2 = p->out2; a.add = p->add; a.aux1 = p->aux1; a.aux2 = p->aux2;
    a.in_sn = p->in_sn; a.out_sn = p->out_sn; a.add_sn = p->add_sn; a.aux1_sn = p->aux1_sn; a.aux2_sn = p->aux2_sn;
    for (int i = 0; i < 3; ++i) {
        const int k = i < p->n_seg ? i : p->n_seg - 1;
        a.weight[i] = p->weight[k];
        a.bias[i] = p->bias[k];
        a.seg_end[i] = i < p->n_seg - 1 ? p->seg_end[i] : p->N;
    }
    a.N = p->N; a.Cin = p->Cin; a.Hin = p->Hin; a.Win = p->Win;
    a.Cout = p->Cout; a.CoutPad = (p->Cout + 15) / 16 * 16; a.Hout = hout; a.Wout = wout;
    a.pad = p->pad; a.act = p->act;
    const int S = p->Cin <= 4 ? 1 : p->Cin <= 8 ? 2 : 4;
    a.nchunk = (p->Cin + 4 * S - 1) / (4 * S);
    const int mt = a.CoutPad / 16;
    // largest tile / channel blocking that still gives every CU >= 2 workgroups; otherwise the most workgroups
    auto blocks = [&](int shape, int mb) -> int64_t {
        const int th = shape == 2 && p->stride == 1 ? 8 : 4, tw = shape == 0 ? 16 : 32;
        return (int64_t)((hout + th - 1) / th) * ((wout + tw - 1) / tw) * (mt / mb) * p->N;   // work items
    };
    int shape = 0, mb = 1;
    int64_t best = -1;
    bool found = false;
    for (int sh = 2; sh >= 0 && !found; --sh)
        for (int m : {2, 1}) {   // MB 3 would need 108 weight registers per chunk
            if (mt % m != 0) continue;
            const int64_t b = blocks(sh, m);
            if (b >= 512) { shape = sh; mb = m; found = true; break; }
            if (b > best) { best = b; shape = sh; mb = m; }
        }
    if (S == 1) {
        if (!s1d1) return 1;
        launch_mb<1, 1, 1>(a, mt, mb, shape, stream);
    } else if (S == 2) {
        if (s1d1) launch_mb<2, 1, 1>(a, mt, mb, shape, stream);
        else if (s2d1) launch_mb<2, 2, 1>(a, mt, mb, shape, stream);
        else return 1;
    } else {
        if (s1d1) launch_mb<4, 1, 1>(a, mt, mb, shape, stream);
        else if (s2d1) launch_mb<4, 2, 1>(a, mt, mb, shape, stream);
        else launch_mb<4, 1, 2>(a, mt, mb, shape, stream);
    }
    return itermvs_launch_status();
}
