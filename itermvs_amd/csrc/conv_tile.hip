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
//   * LDS tile layout    [q][pixel][S]          -> one ds_read_b32/b64/b128 per (tap, 16-pixel segment);
//   * LDS weight layout  [chunk][tap][q][co][S] -> one ds_read per (tap, 16 output channels);
// and every LDS address is `lane base + compile-time offset` (tile geometry, stride and dilation are
// template parameters): no vector address math in the k-loop.  S = 1 / 2 / 4 for Cin <= 4 / <= 8 / larger
// (3, 8 and 16..64 input channels on the path).  The q-plane stride is padded so that a wave's read hits
// each bank once (MI355X_MICROARCH.md, LDS: b128 -> 64 banks, 16-lane groups mixing two q's; b64 -> 64
// banks / 32 lanes; b32 -> 32 banks / 32 lanes).
//
// The weights of the workgroup's channel block (all chunks, all taps: Cin_pad * 576 * MB bytes) are copied
// to LDS ONCE per workgroup and shared by its four waves.  (Held per wave in registers they cost
// 36 * MB * chunks VGPRs and, worse, 27 KB of L1 -> register traffic per wave: on the ConvGRU gate conv the
// 108 weight loads of a wave took 9.4k cycles to issue, twice its MFMA work.)
#include <stdlib.h>

#include "common.hpp"
#include "conv_epilogue.hpp"
#include "conv_tile.hpp"

namespace itermvs {

using f32x2 = __attribute__((ext_vector_type(2))) float;



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

template <int MB, int S, int STRIDE, int DIL, int TH, int TWT, int CPS>
__global__ void __launch_bounds__(256) conv_tile_kernel(const TileArgs a) {
    using G = TileGeom<S, STRIDE, DIL, TH, TWT>;
    constexpr int NB = G::NB;
    constexpr int CH_FLOATS = 4 * G::PL;             // one staged chunk (4*S input channels of the tile + halo)
    constexpr int WROW = 16 * MB * S;                // floats of one weight row (tap, chunk, q): [16*MB co][S]
    constexpr int WBLK = 4 * WROW;                   // floats per (chunk, tap)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ tile = smem;                                 // [CPS][4][PL]
    float* __restrict__ wlds = smem + CPS * CH_FLOATS;               // [nchunk][9][4][16*MB][S]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
    const int m0 = blockIdx.y * (MB * 16);          // the workgroup's channel block is fixed
    // copy of the channel block's weights for weight set `seg`: global rows [tap][chunk][q] of CoutPad*S
    // floats -> LDS rows [chunk][tap][q] of 16*MB*S floats (contiguous 16-byte pieces)
    auto fill_weights = [&](int seg) {
        constexpr int VPR = WROW / 4;                // float4 per row
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.weight[seg] + m0 * S);
        const int rows = a.nchunk * 36;
        const int col = tid % VPR;
        constexpr int RPP = 256 / VPR;               // rows per pass of the workgroup
        // kWPasses passes of loads in flight before their LDS stores (one L2 round trip per batch of passes; up to 48 input
        // channels at MB = 1 are one batch)
        constexpr int kWPasses = 6;
        for (int r0 = tid / VPR; r0 < rows; r0 += kWPasses * RPP) {
            f32x4 t[kWPasses];
#pragma unroll
            for (int i = 0; i < kWPasses; ++i) {
                const int r = r0 + i * RPP;
                const int cidx = r / 36, rem = r - cidx * 36;
                const int tap = rem >> 2, qq = rem & 3;
                const int64_t g = (int64_t)((tap * a.nchunk + cidx) * 4 + qq) * (a.CoutPad * S / 4) + col;
                if (r < rows) t[i] = src[g];
            }
#pragma unroll
            for (int i = 0; i < kWPasses; ++i) {
                const int r = r0 + i * RPP;
                if (r < rows) reinterpret_cast<f32x4*>(wlds)[r * VPR + col] = t[i];
            }
        }
    };

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
    // A STAGE is CPS consecutive chunks (CPS*4*S input channels) of one tile: staged, synchronised and
    // multiplied together.  CPS > 1 shortens the per-tile chain of barriers and weight fetches for the
    // layers with few, deep tiles (ConvGRU: 43 channels = 3 chunks, 320 tiles of 64 pixels).
    Vec<S> stage[CPS][G::ITEMS];
    const uint32_t chunk_b = 4u * S * plane * 4u;   // bytes between chunks in the input planes
    // the CPS*ITEMS*S dword loads of one stage, split into 9*CPS parts: part t is issued inside tap t of
    // the previous stage's MFMA loop.  (Issued in one burst the loads fill the CU's vector-memory queue and
    // the wave sits in the issue of its 24 loads for 2-5k cycles -- as long as the whole MFMA phase.)
    constexpr int kLoads = CPS * G::ITEMS * S;
    constexpr int kParts = 9 * CPS;
    auto fetch_part = [&](uint32_t soff, int part) {
#pragma unroll
        for (int e = 0; e < kLoads; ++e)
            if (e * kParts / kLoads == part) {
                const int c = e / (G::ITEMS * S), j = (e / S) % G::ITEMS, s2 = e % S;
                stage[c][j].v[s2] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(ir, goff[j], soff + c * chunk_b + s2 * plane * 4u, 0));
            }
    };
    auto fetch = [&](uint32_t soff) {
#pragma unroll
        for (int part = 0; part < kParts; ++part) fetch_part(soff, part);
    };

    // this wave's segments: id = wave * NB + nb -> (row, column block) of the output tile
    const int seg0 = wave * NB;
    const int row0 = seg0 / TWT, col0 = seg0 - row0 * TWT;          // NB >= TWT: col0 == 0
    const float* __restrict__ bbase = tile + q * G::PL + ((row0 * STRIDE) * G::IN_W + (col0 * 16 + l16) * STRIDE) * S;
    const float* __restrict__ abase = wlds + (q * 16 * MB + l16) * S;
    const int P = a.Hout * a.Wout;

    // Persistent workgroups: the grid is about two workgroups per CU and each walks the tile list with
    // stride gridDim.x.  The next stage (chunks of this tile, or the first ones of the next tile) is fetched
    // into registers while the matrix cores work on the current one.
    // Tile order (a.banded): workgroup column x runs on XCD x % 8 (round-robin dispatch, gridDim.x a multiple of 8); the tiles of
    // one XCD are a contiguous run of the tile list (whole image bands), so that tiles sharing halo rows meet in the same 4 MB L2
    int w = blockIdx.x, wstep = gridDim.x, wend = a.total;
    if (a.banded) {
        const int xcd = blockIdx.x & 7;
        w = (int)((int64_t)a.total * xcd / 8) + (blockIdx.x >> 3);
        wend = (int)((int64_t)a.total * (xcd + 1) / 8);
        wstep = gridDim.x >> 3;
    }
#ifdef ITERMVS_TILE_TRACE
    int trace_tile = 0;
#endif
    int wseg = -1;
    Work cur = decode(w);
    setup(cur);
    fetch(0);
    while (true) {
        const int seg = (cur.n >= a.seg_end[0]) + (cur.n >= a.seg_end[1]);
        f32x4 acc[MB][NB];
        conv_bias_init<MB, NB>(acc, a.bias[seg], a.Cout, m0, q);
        const int wn = w + wstep;
        Work nxt = cur;
        for (int st = 0; st < a.nstage; ++st) {
            TILE_STAMP(0);
            __syncthreads();                        // the previous stage's LDS reads are done
            if (seg != wseg) {                      // first tile, or a batch item of another weight set
                fill_weights(seg);
                wseg = seg;
            }
#pragma unroll
            for (int c = 0; c < CPS; ++c)
#pragma unroll
                for (int j = 0; j < G::ITEMS; ++j)
                    if (j < G::ITEMS - 1 || tid + j * 256 < 4 * G::IN_PX) lds_write<S>(tile + c * CH_FLOATS + loff[j], stage[c][j]);
            __syncthreads();
            TILE_STAMP(1);
            const float* __restrict__ ast = abase + st * (CPS * 9 * WBLK);
            __builtin_amdgcn_sched_barrier(0);
            // the next stage: the following chunks of this tile, or the first ones of the workgroup's next tile
            bool prefetch = true;
            uint32_t pf_soff = 0;
            if (st + 1 < a.nstage) {
                pf_soff = (uint32_t)((st + 1) * CPS) * chunk_b;
            } else if (wn < wend) {
                nxt = decode(wn);
                setup(nxt);
            } else {
                prefetch = false;
            }
            TILE_STAMP(2);
            // operands of step u+1 are read from LDS before the MFMAs of step u (two register sets): the
            // LDS latency hides behind MB*NB*S MFMAs instead of stalling every tap
            Vec<S> av[2][MB], bv[2][NB];
            auto read_operands = [&](int u, int set) {
                const int c = u / 9, tap = u % 9;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[set][mb] = lds_read<S>(ast + (c * 9 + tap) * WBLK + mb * 16 * S);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int r = (nb / TWT), cc = nb % TWT;     // relative to (row0, col0)
                    bv[set][nb] = lds_read<S>(bbase + c * CH_FLOATS +
                                              ((r * STRIDE + ky * DIL) * G::IN_W + cc * 16 * STRIDE + kx * DIL) * S);
                }
            };
            read_operands(0, 0);
#pragma unroll
            for (int u = 0; u < kParts; ++u) {
                if (u + 1 < kParts) read_operands(u + 1, (u + 1) & 1);
                if (prefetch) fetch_part(pf_soff, u);
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u & 1][mb].v[s2], bv[u & 1][nb].v[s2], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        TILE_STAMP(3);
        // D: col (pixel) = lane & 15, row (cout) = (lane >> 4) * 4 + r
        uint32_t pix_off[NB];
        int py[NB], px[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int oy = cur.oy0 + row0 + nb / TWT, ox = cur.ox0 + (col0 + nb % TWT) * 16 + l16;
            pix_off[nb] = oy < a.Hout && ox < a.Wout ? (uint32_t)(oy * a.Wout + ox) * 4u : kEpiOob;
            py[nb] = oy;
            px[nb] = ox;
        }
        EpilogueArgs e;
        e.out = epi_out_base(a.out, (int64_t)cur.n * a.out_sn, a.out_nhwc);
        e.out2 = a.out2 ? a.out2 + (int64_t)cur.n * a.Cout * P : nullptr;
        e.add = a.add ? a.add + (int64_t)cur.n * a.add_sn : nullptr;
        e.aux1 = a.aux1 ? a.aux1 + (int64_t)cur.n * a.aux1_sn : nullptr;
        e.aux2 = a.aux2 ? a.aux2 + (int64_t)cur.n * a.aux2_sn : nullptr;
        e.Cout = a.Cout; e.P = P; e.act = a.act;
        e.add_mode = a.add_mode; e.Hout = a.Hout; e.Wout = a.Wout; e.out_nhwc = a.out_nhwc;
        int me = m0;
        if (a.split) {          // two results from one launch: channels below / from `split` (uniform per workgroup)
            if (m0 >= a.split) {
                e.out = a.out_b + (int64_t)cur.n * a.out_b_sn;
                e.act = a.act_b;
                e.Cout = a.Cout - a.split;
                me = m0 - a.split;
            } else {
                e.Cout = a.split;
            }
        }
        conv_epilogue<MB, NB>(e, acc, me, q, pix_off, py, px);
        TILE_STAMP(4);
#ifdef ITERMVS_TILE_TRACE
        ++trace_tile;
#endif
        if (wn >= wend) break;
        w = wn;
        cur = nxt;
    }
}

constexpr int kLdsBudget = 64 * 1024;    // per workgroup (default dynamic-LDS limit; two workgroups fit a CU)

template <int MB, int S, int STRIDE, int DIL, int TH, int TWT, int CPS>
static constexpr int tile_lds_bytes(int nchunk) {
    return (CPS * 4 * TileGeom<S, STRIDE, DIL, TH, TWT>::PL + nchunk * 36 * 16 * MB * S) * 4;
}

template <int MB, int S, int STRIDE, int DIL, int TH, int TWT, int CPS>
static int launch_tile(TileArgs& a, int mt, hipStream_t stream) {
    constexpr int TW = 16 * TWT;
    const int lds = tile_lds_bytes<MB, S, STRIDE, DIL, TH, TWT, CPS>(a.nchunk);
    if (lds > kLdsBudget) return 1;
    a.tiles_x = (a.Wout + TW - 1) / TW;
    a.tiles_y = (a.Hout + TH - 1) / TH;
    a.ncb = mt / MB;
    a.nstage = (a.nchunk + CPS - 1) / CPS;
    a.total = a.N * a.tiles_y * a.tiles_x;
    a.rcp_tiles_x = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_x + 1);   // (unused when the divisor is 1)
    a.rcp_tiles_y = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_y + 1);
    // persistent grid: about ITERMVS_TILE_PERSIST (default 4; measured 630 / 645 / 650 / 648 depth-maps/s at
    // 2 / 3 / 4 / 8, bounded by what fits a CU) workgroups per CU in total, each walking the
    // tile list of its channel block (one tile each when there are fewer tiles than that); never more than
    // are resident at once -- a persistent workgroup queued behind another would serialise its tile list
    static const int want = [] { const char* e = itermvs_tuning_env("ITERMVS_TILE_PERSIST"); const int v = e ? atoi(e) : 4; return v < 1 ? 4 : v; }();
    int fit = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, conv_tile_kernel<MB, S, STRIDE, DIL, TH, TWT, CPS>, 256, lds) != hipSuccess || fit < 1)
        fit = 1;
    int gx = 256 * (want < fit ? want : fit) / a.ncb;
    if (gx > a.total) gx = a.total;
    if (gx < 1) gx = 1;
    if (gx >= 16) gx &= ~7;              // a multiple of 8 workgroup columns: the XCD-banded tile order needs it (85 -> 80 costs nothing)
    a.banded = gx % 8 == 0 && a.total >= gx ? 1 : 0;
    const dim3 grid(gx, a.ncb);
    hipLaunchKernelGGL((conv_tile_kernel<MB, S, STRIDE, DIL, TH, TWT, CPS>), grid, dim3(256), lds, stream, a);
    return 0;
}

// chunks per stage: all chunks of a 2..4-chunk layer at once when input stages + weights fit the LDS budget
template <int MB, int S, int STRIDE, int DIL, int TH, int TWT>
static int launch_cps(TileArgs& a, int mt, hipStream_t stream) {
    if constexpr (S == 4) {
        if (a.nchunk == 3 && tile_lds_bytes<MB, S, STRIDE, DIL, TH, TWT, 3>(3) <= kLdsBudget)
            return launch_tile<MB, S, STRIDE, DIL, TH, TWT, 3>(a, mt, stream);
        if ((a.nchunk == 2 || a.nchunk == 4) && tile_lds_bytes<MB, S, STRIDE, DIL, TH, TWT, 2>(a.nchunk) <= kLdsBudget)
            return launch_tile<MB, S, STRIDE, DIL, TH, TWT, 2>(a, mt, stream);
    }
    return launch_tile<MB, S, STRIDE, DIL, TH, TWT, 1>(a, mt, stream);
}

// tile shapes: big = 8 x 32 pixels (4 segments per wave), mid = 4 x 32 (2), small = 4 x 16 (1)
template <int MB, int S, int STRIDE, int DIL>
static int launch_shape(TileArgs& a, int mt, int shape, hipStream_t stream) {
    if (shape == 2) {
        if constexpr (STRIDE == 1) return launch_cps<MB, S, STRIDE, DIL, 8, 2>(a, mt, stream);
        else return launch_cps<MB, S, STRIDE, DIL, 4, 2>(a, mt, stream);   // stride 2: the 8x32 halo tile is too big
    } else if (shape == 1) {
        return launch_cps<MB, S, STRIDE, DIL, 4, 2>(a, mt, stream);
    }
    return launch_cps<MB, S, STRIDE, DIL, 4, 1>(a, mt, stream);
}

template <int S, int STRIDE, int DIL>
static int launch_mb(TileArgs& a, int mt, int mb, int shape, hipStream_t stream) {
    if (mb == 3) return launch_shape<3, S, STRIDE, DIL>(a, mt, shape, stream);
    if (mb == 2) return launch_shape<2, S, STRIDE, DIL>(a, mt, shape, stream);
    return launch_shape<1, S, STRIDE, DIL>(a, mt, shape, stream);
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose2d(3, stride 2, pad 1, output_padding 1) on the matrix cores (CorrNet, itermvs.py:359-363).
// out[2y+py][2x+px] only receives the taps of matching parity:
//   py = 0: ky = 1 from in[y];   py = 1: ky = 0 from in[y+1], ky = 2 from in[y]      (same for px / kx)
// so a wave takes 16 INPUT positions of one row, reads the four neighbours in[y+dy][x+dx] (dy, dx in {0,1})
// from the LDS tile once per chunk and accumulates the four output parities separately: 9 weight reads,
// 4 input reads and 9*S MFMAs per chunk -- the MFMA count of a 3x3 convolution on the input grid.  The
// four parities are the four pixel slots of the shared epilogue (skip connection `add`, bias).
// One tile (TH x 16 input positions, all chunks staged at once) per workgroup: these layers are a few
// hundred tiles.
// ---------------------------------------------------------------------------------------------
template <int MB, int S, int TH, int NCH>
__global__ void __launch_bounds__(256) deconv_tile_kernel(const TileArgs a) {
    constexpr int IN_H = TH + 1, IN_W = 17, IN_PX = IN_H * IN_W;
    constexpr int PL = (IN_PX * S + 63) / 64 * 64 + (S == 4 ? 0 : S == 2 ? 32 : 16);
    constexpr int CH_FLOATS = 4 * PL;
    constexpr int ITEMS = (4 * IN_PX + 255) / 256;
    constexpr int WROW = 16 * MB * S, WBLK = 4 * WROW, VPR = WROW / 4;
    static_assert(TH == 4, "one input row per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ tile = smem;                        // [NCH][4][PL]
    float* __restrict__ wlds = smem + NCH * CH_FLOATS;      // [NCH][9][4][16*MB][S]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
    const int m0 = blockIdx.y * (MB * 16);
    const int t2 = blockIdx.x / a.tiles_x, tx = blockIdx.x - t2 * a.tiles_x;
    const int n = t2 / a.tiles_y, ty = t2 - n * a.tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;
    const int seg = (n >= a.seg_end[0]) + (n >= a.seg_end[1]);

    // input tile (rows y0 .. y0+TH, columns x0 .. x0+16; beyond the image -> 0), all chunks
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (int64_t)n * a.in_sn), 0, (int)(a.Cin * plane * 4u), 0x00020000);
    Vec<S> stage[NCH][ITEMS];
    int loff[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int item = tid + j * 256;
        const int iq = item / IN_PX, px = item - iq * IN_PX;
        const int y = px / IN_W, x = px - y * IN_W;
        const int gy = y0 + y, gx = x0 + x;
        const bool ok = item < 4 * IN_PX && gy < a.Hin && gx < a.Win;
        const uint32_t goff = ok ? ((uint32_t)(iq * S) * plane + (uint32_t)(gy * a.Win + gx)) * 4u : kTileOob;
        loff[j] = iq * PL + px * S;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2)
                stage[c][j].v[s2] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(ir, goff, (uint32_t)(c * 4 * S + s2) * plane * 4u, 0));
    }
    // weights of this channel block: global rows [tap][chunk][q] -> LDS rows [chunk][tap][q]
    {
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.weight[seg] + m0 * S);
        const int col = tid % VPR;
        for (int r = tid / VPR; r < NCH * 36; r += 256 / VPR) {
            const int cidx = r / 36, rem = r - cidx * 36;
            const int64_t g = (int64_t)(((rem >> 2) * NCH + cidx) * 4 + (rem & 3)) * (a.CoutPad * S / 4) + col;
            reinterpret_cast<f32x4*>(wlds)[r * VPR + col] = src[g];
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (j < ITEMS - 1 || tid + j * 256 < 4 * IN_PX) lds_write<S>(tile + c * CH_FLOATS + loff[j], stage[c][j]);
    __syncthreads();

    f32x4 acc[MB][4];     // [.][py * 2 + px]
    conv_bias_init<MB, 4>(acc, a.bias[seg], a.Cout, m0, q);
    const float* __restrict__ bbase = tile + q * PL + (wave * IN_W + l16) * S;
    const float* __restrict__ abase = wlds + (q * 16 * MB + l16) * S;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        Vec<S> bv[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) bv[dy][dx] = lds_read<S>(bbase + c * CH_FLOATS + (dy * IN_W + dx) * S);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int py = ky == 1 ? 0 : 1, dy = ky == 0 ? 1 : 0;
            const int px = kx == 1 ? 0 : 1, dx = kx == 0 ? 1 : 0;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const Vec<S> av = lds_read<S>(abase + (c * 9 + tap) * WBLK + mb * 16 * S);
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2)
                    acc[mb][py * 2 + px] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.v[s2], bv[dy][dx].v[s2], acc[mb][py * 2 + px], 0, 0, 0);
            }
        }
    }

    const int P = a.Hout * a.Wout;
    const int iy = y0 + wave, ix = x0 + l16;
    uint32_t pix_off[4];
    int oyv[4], oxv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        oyv[k] = 2 * iy + (k >> 1);
        oxv[k] = 2 * ix + (k & 1);
        pix_off[k] = iy < a.Hin && ix < a.Win ? (uint32_t)(oyv[k] * a.Wout + oxv[k]) * 4u : kEpiOob;
    }
    EpilogueArgs e;
    e.out = a.out + (int64_t)n * a.out_sn;
    e.out2 = a.out2 ? a.out2 + (int64_t)n * a.Cout * P : nullptr;
    e.add = a.add ? a.add + (int64_t)n * a.add_sn : nullptr;
    e.aux1 = nullptr; e.aux2 = nullptr;
    e.Cout = a.Cout; e.P = P; e.act = a.act;
    e.add_mode = 0; e.Hout = a.Hout; e.Wout = a.Wout; e.out_nhwc = 0;
    conv_epilogue<MB, 4>(e, acc, m0, q, pix_off, oyv, oxv);
}

template <int MB, int S, int NCH>
static int launch_deconv(TileArgs& a, int mt, hipStream_t stream) {
    constexpr int TH = 4, IN_PX = (TH + 1) * 17;
    constexpr int PL = (IN_PX * S + 63) / 64 * 64 + (S == 4 ? 0 : S == 2 ? 32 : 16);
    constexpr int lds = (NCH * 4 * PL + NCH * 36 * 16 * MB * S) * 4;
    static_assert(lds <= kLdsBudget, "deconv tile does not fit LDS");
    a.tiles_x = (a.Win + 15) / 16;
    a.tiles_y = (a.Hin + TH - 1) / TH;
    const dim3 grid(a.N * a.tiles_y * a.tiles_x, mt / MB);
    hipLaunchKernelGGL((deconv_tile_kernel<MB, S, TH, NCH>), grid, dim3(256), lds, stream, a);
    return 0;
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
    a.banded = 0;
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
    a.pad = p->pad; a.act = p->act; a.add_mode = p->add_mode; a.out_nhwc = p->out_layout;
    a.split = p->split_cout; a.act_b = p->act_b; a.out_b = p->out_b; a.out_b_sn = p->out_b_sn;
    const int S = p->Cin <= 4 ? 1 : p->Cin <= 8 ? 2 : 4;
    a.nchunk = (p->Cin + 4 * S - 1) / (4 * S);
    const int mt = a.CoutPad / 16;
    // largest tile / channel blocking that still gives every CU >= 4 workgroups; otherwise the most workgroups
    auto blocks = [&](int shape, int mb) -> int64_t {
        const int th = shape == 2 && p->stride == 1 ? 8 : 4, tw = shape == 0 ? 16 : 32;
        return (int64_t)((hout + th - 1) / th) * ((wout + tw - 1) / tw) * (mt / mb) * p->N;   // work items
    };
    // LDS bytes of a candidate when ALL chunks of a tile are staged together (launch_cps picks that when it fits)
    auto full_stage_fits = [&](int sh, int m) {
        const int th = sh == 2 && p->stride == 1 ? 8 : 4, tw = sh == 0 ? 16 : 32;
        const int in_px = ((th - 1) * p->stride + 2 * p->dilation + 1) * ((tw - 1) * p->stride + 2 * p->dilation + 1);
        const int pl = (in_px * S + 63) / 64 * 64 + (S == 4 ? 0 : S == 2 ? 32 : 16);
        const int cps = a.nchunk == 4 ? 2 : a.nchunk;
        return (cps * 4 * pl + a.nchunk * 36 * 16 * m * S) * 4 <= kLdsBudget;
    };
    // (only for layers of a few hundred tiles, where the per-tile chain of barriers and weight copies is what
    //  bounds the launch; large layers amortise it over big tiles)
    const bool deep = a.nchunk >= 2 && a.nchunk <= 4 && blocks(2, 1) < 1024;
    const char* force = itermvs_tuning_env("ITERMVS_TILE_FORCE");             // "shape,mb" (tools/conv_bench.py --sweep)
    static const bool tuned = [] { const char* e = itermvs_tuning_env("ITERMVS_TILE_TUNED"); return !e || e[0] != '0'; }();
    static const int min_work = [] { const char* e = itermvs_tuning_env("ITERMVS_TILE_MINWORK"); return e ? atoi(e) : 1024; }();   // work items wanted: 4 workgroups per CU
    int shape = 0, mb = 1;
    int64_t best = -1;
    bool found = false;
    // multi-chunk layers: prefer candidates that stage a whole tile at once (one barrier pair and one weight
    // copy per tile instead of one per chunk)
    for (int pass = deep ? 0 : 1; pass < 2 && !found; ++pass) {
        best = -1;
        for (int sh = 2; sh >= 0 && !found; --sh)
            for (int m : {3, 2, 1}) {
                if (mt % m != 0 || (p->split_cout && (p->split_cout / 16) % m != 0)) continue;
                if (pass == 0 && !full_stage_fits(sh, m)) continue;
                const int64_t b = blocks(sh, m);
                if (b >= min_work) { shape = sh; mb = m; found = true; break; }
                if (b > best) { best = b; shape = sh; mb = m; }
            }
        if (best >= 0) found = true;       // pass 0 found a full-stage candidate (the one with the most work items)
    }
    // 16+ input channels (S = 4): measured sweep over all (tile shape, channel blocking) pairs on the layers of
    // the path (tools/conv_bench.py --sweep): the 8x32 tile never wins there.  An even number of channel blocks
    // runs best as 4x16 tiles with two blocks per wave (the stride-2 stages, 32->32, 48->32, 32->64: -10..-25 %),
    // an odd one as 4x32 tiles with one block (16->16, 48->48, 48->16); the dilated ConvGRU convolutions as 4x32.
    if (tuned && S == 4 && !p->split_cout) {
        if (p->dilation == 2) {
            if (a.nchunk == 3) { shape = 1; mb = 1; }
        } else if (mt % 2 == 0) {
            shape = 0; mb = 2;
        } else {
            shape = 1; mb = 1;
        }
    } else if (tuned && S == 2 && p->stride == 2 && mt % 2 == 0 && (!p->split_cout || (p->split_cout / 16) % 2 == 0)) {
        shape = 0; mb = 2;                                                            // 8 -> 16+16, stride 2
    } else if (tuned && S == 4 && p->split_cout) {
        if (p->dilation == 2) { shape = 1; mb = 1; }                                  // z / r gates
        else if ((p->split_cout / 16) % 2 == 0 && mt % 2 == 0) { shape = 0; mb = 2; }  // stride-2 conv + shortcut
    }
    if (force) {
        shape = force[0] - '0';
        mb = force[2] - '0';
        if (shape < 0 || shape > 2 || mb < 1 || mb > 3 || mt % mb != 0) return 1;
    }
    const bool dot = p->act == 6 || p->act == 7;     // the epilogue contracts over ALL output channels: one block per wave
    if (dot) {
        mb = mt;
        if (mt == 2) shape = 0;                      // two blocks per wave run as 4x16 tiles
    }
    // the weights of the channel block must fit LDS next to at least one input stage: narrow the block
    int rc = 1;
    for (; rc == 1 && mb >= 1; --mb) {
        if (dot && mb != mt) return 1;
        if (mt % mb != 0 || (p->split_cout && (p->split_cout / 16) % mb != 0)) continue;
        if (S == 1) {
            if (!s1d1) return 1;
            rc = launch_mb<1, 1, 1>(a, mt, mb, shape, stream);
        } else if (S == 2) {
            if (s1d1) rc = launch_mb<2, 1, 1>(a, mt, mb, shape, stream);
            else if (s2d1) rc = launch_mb<2, 2, 1>(a, mt, mb, shape, stream);
            else return 1;
        } else {
            if (s1d1) rc = launch_mb<4, 1, 1>(a, mt, mb, shape, stream);
            else if (s2d1) rc = launch_mb<4, 2, 1>(a, mt, mb, shape, stream);
            else rc = launch_mb<4, 1, 2>(a, mt, mb, shape, stream);
        }
    }
    if (rc != 0) return 1;
    return itermvs_launch_status();
}

// transposed convolutions (weight_format 2 built from the ConvTranspose2d weight with in/out channels swapped);
// returns 1 when the shape is not covered
int itermvs_deconv2d_tile(const itermvs_conv_params* p, hipStream_t stream) {
    if (p->ksize != 3 || p->stride != 2 || p->pad != 1 || p->act > 1 || p->Cin <= 4 || p->Cin > 32) return 1;
    TileArgs a;
    a.banded = 0;
    a.in = p->in; a.out = p->out; a.out2 = p->out2; a.add = p->add; a.aux1 = nullptr; a.aux2 = nullptr;
    a.in_sn = p->in_sn; a.out_sn = p->out_sn; a.add_sn = p->add_sn; a.aux1_sn = 0; a.aux2_sn = 0;
    for (int i = 0; i < 3; ++i) {
        const int k = i < p->n_seg ? i : p->n_seg - 1;
        a.weight[i] = p->weight[k];
        a.bias[i] = p->bias[k];
        a.seg_end[i] = i < p->n_seg - 1 ? p->seg_end[i] : p->N;
    }
    a.N = p->N; a.Cin = p->Cin; a.Hin = p->Hin; a.Win = p->Win;
    a.Cout = p->Cout; a.CoutPad = (p->Cout + 15) / 16 * 16; a.Hout = 2 * p->Hin; a.Wout = 2 * p->Win;
    a.pad = 1; a.act = p->act; a.add_mode = 0; a.out_nhwc = 0;
    a.split = 0; a.act_b = 0; a.out_b = nullptr; a.out_b_sn = 0;
    const int S = p->Cin <= 8 ? 2 : 4;
    a.nchunk = (p->Cin + 4 * S - 1) / (4 * S);
    const int mt = a.CoutPad / 16;
    if (mt > 2) return 1;
    int rc = 1;
    if (S == 2) rc = mt == 2 ? launch_deconv<2, 2, 1>(a, mt, stream) : launch_deconv<1, 2, 1>(a, mt, stream);
    else if (a.nchunk == 1) rc = mt == 2 ? launch_deconv<2, 4, 1>(a, mt, stream) : launch_deconv<1, 4, 1>(a, mt, stream);
    else rc = mt == 2 ? launch_deconv<2, 4, 2>(a, mt, stream) : launch_deconv<1, 4, 2>(a, mt, stream);
    if (rc != 0) return 1;
    return itermvs_launch_status();
}
