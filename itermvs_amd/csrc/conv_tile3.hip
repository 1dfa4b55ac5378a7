// itermvs_conv2d, 3x3 convolutions with >= 9 input channels on the bf16 matrix instructions: fp32 operands split into
// three bf16 terms each ("bf16x3", weight_format 3).
//
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (one instruction per 32 cycles, 64 FLOP/clk/SIMD);
// v_mfma_f32_16x16x32_bf16 at 16x that.  An fp32 value is EXACTLY the sum of three bf16 values
//     x = h + m + l,   h = x truncated to 8 significant bits, m = (x - h) truncated, l = x - h - m  (all three exact for
//     |x| >= ~1e-33; below that the low terms fall under bf16's smallest subnormal: absolute error < 1e-40),
// so a product of two fp32 values is the sum of nine bf16 x bf16 products (each exact in fp32).  The six largest,
//     xh wh  +  (xh wm + xm wh)  +  (xm wm + xh wl + xl wh),
// leave out terms below 2^-23 of |x w| -- the size of one fp32 rounding of the product itself.  They cost three MFMAs per
// 16 input channels, because the K = 32 of the instruction holds TWO 16-channel terms side by side:
//     A1 = [wh | wh]   A2 = [wm | wm]   A3 = [wl | wh]          (16 output channels x 32)
//     B1 = [xh | xm]                    B3 = [xh | xl]          (32 x 16 pixels)
//     acc += A3 B3 (wl xh + wh xl);  acc += A2 B1 (wm xh + wm xm);  acc += A1 B1 (wh xh + wh xm)      [small terms first]
// against four fp32 MFMAs of 32 cycles: 48 instead of 128 matrix-pipe cycles per (tap, 16 channels, 16 x 16 outputs), with
// the accumulation still in fp32.  (Five distinct operand registers per three MFMAs is the minimum: six products from
// three two-term instructions cannot be covered by 2 + 2.)  Measured before it was built: profiles/r05/
// r05a_conv_tile_mfma_knockout.txt -- the fp32 kernel with its MFMAs replaced by three bf16 ones ran the layer set in
// 705 instead of 896 us.
//
// Everything else follows conv_tile.hip: persistent workgroups of 4 waves own a TH x 16*TWT tile of output pixels and one
// block of 16*MB output channels; the input tile + halo of a STAGE (CPS chunks of 16 channels) is fetched into registers
// while the previous stage multiplies (loads spread over the taps), split, and stored to LDS once; the channel block's
// weights (split on the host, ops.MfmaWeight) are copied to LDS once per workgroup; every LDS address is lane base +
// immediate.  LDS images (bytes):
//   tile    [chunk][plane h,m,l][half][pixel][8 bf16]    16 B per (pixel, 8 channels); planes padded to 256 B, so that a
//           ds_read_b128 of a wave (16-lane groups mixing the two halves) and the ds_write_b128 of the staging hit every
//           bank once;
//   weights [chunk][tap][plane h,m,l][16*MB rows][16 ch bf16]   32 B per output channel.
// An MFMA lane (i = lane & 15, q = lane >> 4) holds K indices 8q .. 8q+7: half (q & 1) of the 16 channels of the first
// (q < 2) or second (q >= 2) term -- a per-lane constant plane offset selects the term.
// PAIR form (5 .. 8 input channels, stride 1, dilation 1: the 3x3 layer of PixelViewWeight, itermvs.py:337-341, 1.5 GFLOP of fp32
// MFMAs at cfg 1): eight channels fill only half of a 16-channel term, so the K = 32 carries TWO TAPS: channel slots 0..7 hold
// the 8 channels at tap 2v, slots 8..15 the same 8 channels at tap 2v + 1 (v = 0..4; the missing tenth tap has zero weights,
// ops.MfmaWeight).  Only half 0 of the tile is staged; a lane of half 1 reads it at the second tap's displacement -- a per-lane
// select between two immediates.  5 x 3 MFMAs of 16 cycles instead of 18 fp32 MFMAs of 32 per 16 x 16 outputs.
// With 3*MB + 2*NB operand reads of 1 KB for 3*MB*NB MFMAs per tap, a wave needs MB*NB >= 4 accumulator tiles to stay
// under one ds_read_b128 per MFMA, the rate the LDS sustains beside the matrix pipe (tools/ubench/mfma_bf16_rate.hip).
#include <stdlib.h>

#include "common.hpp"
#include "conv_epilogue.hpp"
#include "conv_tile.hpp"

namespace itermvs {


template <int STRIDE, int DIL, int TH, int TWT>
struct Tile3Geom {
    static constexpr int TW = 16 * TWT;
    static constexpr int NB = TH * TWT / 4;                          // 16-pixel segments per wave
    static constexpr int IN_H = (TH - 1) * STRIDE + 2 * DIL + 1;
    static constexpr int IN_W = (TW - 1) * STRIDE + 2 * DIL + 1;
    static constexpr int IN_PX = IN_H * IN_W;
    static constexpr int PLB = (IN_PX * 16 + 255) / 256 * 256;       // bytes per (plane, half)
    static constexpr int ITEMS = (2 * IN_PX + 255) / 256;            // (half, pixel) staging items per thread
    static_assert(TH * TWT % 4 == 0, "segments must divide over 4 waves");
    static_assert(NB % TWT == 0 || TWT % NB == 0, "a wave's segments must form whole rows or a row part");
};

#ifndef ITERMVS_TILE3_DBUF
#define ITERMVS_TILE3_DBUF 1
#endif

// INCL (PAIR form, exactly 8 input channels): the input is channels last, [N][H][W][8] -- a staging item (the 8 channels of a pixel)
// is two 16-byte loads instead of eight dword loads.  At 64 wave-level loads per tile and ~64 cycles of the CU's address path each,
// the dword form's staging alone was 21 of the 24 us of PixelViewWeight's 3x3 layer (3 072 tiles on 512 workgroups).
template <int MB, int STRIDE, int DIL, int TH, int TWT, int CPS, int PAIR = 0, int INCL = 0>
__global__ void __launch_bounds__(256) conv_tile3_kernel(const TileArgs a) {
    static_assert(!INCL || PAIR, "channels-last input: tap-pair form only");
    using G = Tile3Geom<STRIDE, DIL, TH, TWT>;
    constexpr int NB = G::NB;
    constexpr int TAPS = PAIR ? 5 : 9;               // K-steps per chunk: taps, or tap pairs
    constexpr int NIT = PAIR ? (G::IN_PX + 255) / 256 : G::ITEMS;      // staging items per thread (PAIR: half 0 only)
    constexpr int NLIVE = PAIR ? G::IN_PX : 2 * G::IN_PX;
    static_assert(!PAIR || (CPS == 1 && STRIDE == 1 && DIL == 1), "tap pairs: one chunk, stride 1, no dilation");
    constexpr int kDbuf = ITERMVS_TILE3_DBUF;
    constexpr int PLB = G::PLB;
    constexpr int CH_BYTES = 6 * PLB;                // one staged chunk: [plane][half][PLB]
    constexpr int WPL = 16 * MB * 32;                // one weight plane of a (chunk, tap): [16*MB rows][32 B]
    constexpr int WBLK = 3 * WPL;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    char* __restrict__ tile = smem3;                                  // [CPS][3][2][PLB]
    char* __restrict__ wlds = smem3 + CPS * CH_BYTES;                 // [nchunk][9][3][16*MB][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const uint32_t plane = (uint32_t)(a.Hin * a.Win);
    const int m0 = blockIdx.y * (MB * 16);
    // the channel block's weights of weight set `seg`: global [tap][chunk][plane][CoutPad][16] bf16 -> LDS [chunk][tap][plane][16*MB][16]
    auto fill_weights = [&](int seg) {
        constexpr int PPG = 32 * MB;                 // 16-byte pieces per (chunk, tap, plane)
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(a.weight[seg]);
        const int total = a.nchunk * (TAPS * 3) * PPG;
        constexpr int kBatch = 6;                    // loads in flight before their LDS stores
        for (int p0 = tid; p0 < total; p0 += kBatch * 256) {
            u32x4 t[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int pc = p0 + i * 256;
                const int grp = pc / PPG, within = pc - grp * PPG;
                const int cidx = grp / (TAPS * 3), rem = grp - cidx * (TAPS * 3);
                const int tap = rem / 3, pl = rem - tap * 3;
                const int64_t g = ((int64_t)((tap * a.nchunk + cidx) * 3 + pl) * a.CoutPad + m0) * 2 + within;
                if (pc < total) t[i] = src[g];
            }
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int pc = p0 + i * 256;
                if (pc < total) reinterpret_cast<u32x4*>(wlds)[pc] = t[i];
            }
        }
    };

    struct Work { int n, oy0, ox0; };
    auto decode = [&](int w) {
        Work k;
        const int t2 = a.tiles_x == 1 ? w : (int)__umulhi((uint32_t)w, a.rcp_tiles_x);
        const int tx = w - t2 * a.tiles_x;
        k.n = a.tiles_y == 1 ? t2 : (int)__umulhi((uint32_t)t2, a.rcp_tiles_y);
        const int ty = t2 - k.n * a.tiles_y;
        k.oy0 = ty * TH;
        k.ox0 = tx * G::TW;
        return k;
    };

    // staging items of this thread: item = (half of the chunk's 16 channels, pixel of the input tile): 8 channels, one
    // dword load per channel plane, three 16-byte LDS stores (h, m, l)
    int rel[NIT], loff[NIT];
    uint32_t reloff[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int item = tid + j * 256;
        const int hf = PAIR ? 0 : item / G::IN_PX;
        const int px = item - hf * G::IN_PX;
        const int y = px / G::IN_W, x = px - y * G::IN_W;
        const bool live = item < NLIVE;
        rel[j] = live ? (y << 12) | x : -1;
        reloff[j] = !live ? kTileOob : INCL ? (uint32_t)(y * a.Win + x) * 32u : ((uint32_t)(hf * 8) * plane + (uint32_t)(y * a.Win + x)) * 4u;
        loff[j] = hf * PLB + px * 16;
    }
    uint32_t goff[NIT];
    __amdgpu_buffer_rsrc_t ir;
    auto setup = [&](const Work& k) {
        ir = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (int64_t)k.n * a.in_sn), 0, (int)(a.Cin * plane * 4u), 0x00020000);
        const int iy0 = k.oy0 * STRIDE - a.pad, ix0 = k.ox0 * STRIDE - a.pad;
        constexpr uint32_t kPxB = INCL ? 32u : 4u;      // bytes per pixel step
        if (iy0 >= 0 && ix0 >= 0 && iy0 + G::IN_H <= a.Hin && ix0 + G::IN_W <= a.Win) {
            const uint32_t base = (uint32_t)(iy0 * a.Win + ix0) * kPxB;
#pragma unroll
            for (int j = 0; j < NIT; ++j) goff[j] = reloff[j] + base;
        } else {
            const int base = (iy0 * a.Win + ix0) * (int)kPxB;
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int gy = iy0 + (rel[j] >> 12), gx = ix0 + (rel[j] & 0xfff);
                const bool ok = rel[j] >= 0 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
                goff[j] = ok ? reloff[j] + (uint32_t)base : kTileOob;
            }
        }
    };
    float stage[CPS][NIT][8];
    const uint32_t chunk_b = 16u * plane * 4u;       // bytes between chunks in the input planes
    constexpr int kLoads = INCL ? NIT * 2 : CPS * NIT * 8;
    constexpr int kParts = TAPS * CPS;
    auto fetch_part = [&](uint32_t soff, int part) {
#pragma unroll
        for (int e = 0; e < kLoads; ++e)
            if (e * kParts / kLoads == part) {
                if constexpr (INCL) {
                    const int j = e / 2, hl = e % 2;
                    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ir, goff[j], soff + hl * 16u, 0));
#pragma unroll
                    for (int k = 0; k < 4; ++k) stage[0][j][hl * 4 + k] = __uint_as_float(v[k]);
                } else {
                    const int c = e / (NIT * 8), j = (e / 8) % NIT, k = e % 8;
                    stage[c][j][k] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(ir, goff[j], soff + c * chunk_b + k * plane * 4u, 0));
                }
            }
    };
    auto fetch = [&](uint32_t soff) {
#pragma unroll
        for (int part = 0; part < kParts; ++part) fetch_part(soff, part);
    };

    const int seg0 = wave * NB;
    const int row0 = seg0 / TWT, col0 = seg0 - row0 * TWT;
    const int half = q & 1, second = q >> 1;          // half of the 16 channels; first / second term of the K = 32
    const int pix0 = (row0 * STRIDE) * G::IN_W + (col0 * 16 + l16) * STRIDE;
    // (PAIR: both halves read half 0's planes; the half selects the tap, see read_operands)
    const char* __restrict__ bbase1 = tile + (second ? 2 * PLB : 0) + (PAIR ? 0 : half * PLB) + pix0 * 16;     // B1 = [xh | xm]
    const char* __restrict__ bbase3 = tile + (second ? 4 * PLB : 0) + (PAIR ? 0 : half * PLB) + pix0 * 16;     // B3 = [xh | xl]
    const char* __restrict__ abase = wlds + l16 * 32 + half * 16;                                 // A1 = [wh | wh]; A2 = + WPL
    const char* __restrict__ abase3 = abase + (second ? 0 : 2 * WPL);                             // A3 = [wl | wh]
    const int P = a.Hout * a.Wout;

    // Tile order (a.banded): workgroup column x runs on XCD x % 8 (round-robin dispatch, gridDim.x a multiple of 8); the tiles of
    // one XCD are a contiguous run of the tile list (whole image bands), so that tiles sharing halo rows meet in the same 4 MB L2
    int w = blockIdx.x, wstep = gridDim.x, wend = a.total;
    if (a.banded) {
        const int xcd = blockIdx.x & 7;
        w = (int)((int64_t)a.total * xcd / 8) + (blockIdx.x >> 3);
        wend = (int)((int64_t)a.total * (xcd + 1) / 8);
        wstep = gridDim.x >> 3;
    }
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
            __syncthreads();                        // the previous stage's LDS reads are done
            if (seg != wseg) {
                fill_weights(seg);
                wseg = seg;
            }
#pragma unroll
            for (int c = 0; c < CPS; ++c)
#pragma unroll
                for (int j = 0; j < NIT; ++j)
                    if (j < NIT - 1 || tid + j * 256 < NLIVE) {
                        u32x4 H, M, L;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint32_t h, m, l;
                            split_pair(stage[c][j][2 * k], stage[c][j][2 * k + 1], h, m, l);
                            H[k] = h; M[k] = m; L[k] = l;
                        }
                        char* d = tile + c * CH_BYTES + loff[j];
                        *reinterpret_cast<u32x4*>(d) = H;
                        *reinterpret_cast<u32x4*>(d + 2 * PLB) = M;
                        *reinterpret_cast<u32x4*>(d + 4 * PLB) = L;
                    }
            __syncthreads();
            const int wst = st * (CPS * TAPS * WBLK);
            __builtin_amdgcn_sched_barrier(0);
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
            // operands of step u+1 are read from LDS before the MFMAs of step u (two register sets), unless the operand sets
            // are so large that the second one costs a wave of occupancy (kDbuf)
            bf8 a1[kDbuf + 1][MB], a2[kDbuf + 1][MB], a3[kDbuf + 1][MB], b1[kDbuf + 1][NB], b3[kDbuf + 1][NB];
            auto read_operands = [&](int u, int set) {
                const int c = u / TAPS, tap = u % TAPS;
                // PAIR: K-step `tap` = taps 2 tap (half 0) and 2 tap + 1 (half 1; the tenth tap re-reads the ninth: zero weights)
                const int ta = PAIR ? 2 * tap : tap, tb = PAIR ? (2 * tap + 1 < 9 ? 2 * tap + 1 : 8) : tap;
                const int ky = ta / 3, kx = ta - ky * 3, kyb = tb / 3, kxb = tb - kyb * 3;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int o = wst + (c * TAPS + tap) * WBLK + mb * 16 * 32;
                    a1[set][mb] = *reinterpret_cast<const bf8*>(abase + o);
                    a2[set][mb] = *reinterpret_cast<const bf8*>(abase + o + WPL);
                    a3[set][mb] = *reinterpret_cast<const bf8*>(abase3 + o);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int r = nb / TWT, cc = nb % TWT;
                    const int oa = c * CH_BYTES + ((r * STRIDE + ky * DIL) * G::IN_W + cc * 16 * STRIDE + kx * DIL) * 16;
                    const int ob = c * CH_BYTES + ((r * STRIDE + kyb * DIL) * G::IN_W + cc * 16 * STRIDE + kxb * DIL) * 16;
                    const int o = PAIR ? (half ? ob : oa) : oa;
                    b1[set][nb] = *reinterpret_cast<const bf8*>(bbase1 + o);
                    b3[set][nb] = *reinterpret_cast<const bf8*>(bbase3 + o);
                }
            };
            if constexpr (kDbuf) read_operands(0, 0);
#pragma unroll
            for (int u = 0; u < kParts; ++u) {
                if constexpr (kDbuf) {
                    if (u + 1 < kParts) read_operands(u + 1, (u + 1) & 1);
                } else {
                    read_operands(u, 0);
                }
                if (prefetch) fetch_part(pf_soff, u);
                const int s = kDbuf ? (u & 1) : 0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[s][mb], b3[s][nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s][mb], b1[s][nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s][mb], b1[s][nb], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // D: col (pixel) = lane & 15, row (cout) = (lane >> 4) * 4 + r   (the layout of the fp32 form: shared epilogue)
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
        if (a.split) {
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
        if (wn >= wend) break;
        w = wn;
        cur = nxt;
    }
}


#ifndef ITERMVS_TILE3_SPEC
#define ITERMVS_TILE3_SPEC 0
#endif
#if ITERMVS_TILE3_SPEC       // producer / consumer form of the kernel above: A/B builds only (measured slower)
#include "experiments/conv_tile3_producer_consumer.inc"
#endif

constexpr int kLds3Budget = 80 * 1024;     // per workgroup: two workgroups fit the CU's 160 KB

template <int MB, int STRIDE, int DIL, int TH, int TWT, int CPS>
static constexpr int tile3_lds_bytes(int nchunk, int taps = 9) {
    return CPS * 6 * Tile3Geom<STRIDE, DIL, TH, TWT>::PLB + nchunk * taps * 3 * 16 * MB * 32;
}

// PAIR form: one instantiation per channel blocking (8 x 32 tiles, one chunk)
template <int MB, int INCL = 0>
static int launch_tile3_pair(TileArgs& a, int mt, hipStream_t stream) {
    constexpr int TH = 8, TWT = 2, TW = 16 * TWT;
    const int lds = tile3_lds_bytes<MB, 1, 1, TH, TWT, 1>(1, 5);
    auto kern = conv_tile3_kernel<MB, 1, 1, TH, TWT, 1, 1, INCL>;
    a.tiles_x = (a.Wout + TW - 1) / TW;
    a.tiles_y = (a.Hout + TH - 1) / TH;
    a.ncb = mt / MB;
    a.nstage = 1;
    a.total = a.N * a.tiles_y * a.tiles_x;
    a.rcp_tiles_x = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_x + 1);
    a.rcp_tiles_y = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_y + 1);
    int fit = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, 256, lds) != hipSuccess || fit < 1) fit = 1;
    int gx = itermvs_num_cus() * (4 < fit ? 4 : fit) / a.ncb;
    if (gx > a.total) gx = a.total;
    if (gx < 1) gx = 1;
    if (gx >= 16) gx &= ~7;              // a multiple of 8 workgroup columns: the XCD-banded tile order needs it (85 -> 80 costs nothing)
    a.banded = gx % 8 == 0 && a.total >= gx ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(gx, a.ncb), dim3(256), lds, stream, a);
    return 0;
}

template <int MB, int STRIDE, int DIL, int TH, int TWT, int CPS>
static int launch_tile3(TileArgs& a, int mt, hipStream_t stream) {
    constexpr int TW = 16 * TWT;
    const int lds = tile3_lds_bytes<MB, STRIDE, DIL, TH, TWT, CPS>(a.nchunk);
    if (lds > kLds3Budget) return 1;
    auto kern = conv_tile3_kernel<MB, STRIDE, DIL, TH, TWT, CPS>;
    static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds3Budget) == hipSuccess;
    if (lds > 64 * 1024 && !big_ok) return 1;
    a.tiles_x = (a.Wout + TW - 1) / TW;
    a.tiles_y = (a.Hout + TH - 1) / TH;
    a.ncb = mt / MB;
    a.nstage = (a.nchunk + CPS - 1) / CPS;
    a.total = a.N * a.tiles_y * a.tiles_x;
    a.rcp_tiles_x = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_x + 1);
    a.rcp_tiles_y = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_y + 1);
    static const int want = [] { const char* e = itermvs_tuning_env("ITERMVS_TILE_PERSIST"); const int v = e ? atoi(e) : 4; return v < 1 ? 4 : v; }();
    int fit = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, 256, lds) != hipSuccess || fit < 1) fit = 1;
    int gx = itermvs_num_cus() * (want < fit ? want : fit) / a.ncb;
    if (gx > a.total) gx = a.total;
    if (gx < 1) gx = 1;
    if (gx >= 16) gx &= ~7;              // a multiple of 8 workgroup columns: the XCD-banded tile order needs it (85 -> 80 costs nothing)
    a.banded = gx % 8 == 0 && a.total >= gx ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(gx, a.ncb), dim3(256), lds, stream, a);
    return 0;
}

// chunks per stage: all chunks of a 2..4-chunk layer at once when the stages + weights fit the LDS budget
template <int MB, int STRIDE, int DIL, int TH, int TWT>
static int launch_cps3(TileArgs& a, int mt, hipStream_t stream) {
#if ITERMVS_TILE3_SPEC
    if (a.nchunk == 3 && tile3s_lds_bytes<MB, STRIDE, DIL, TH, TWT, 3>(3) <= kLds3sBudget)
        return launch_tile3s<MB, STRIDE, DIL, TH, TWT, 3>(a, mt, stream);
    if ((a.nchunk == 2 || a.nchunk == 4) && tile3s_lds_bytes<MB, STRIDE, DIL, TH, TWT, 2>(a.nchunk) <= kLds3sBudget)
        return launch_tile3s<MB, STRIDE, DIL, TH, TWT, 2>(a, mt, stream);
    return launch_tile3s<MB, STRIDE, DIL, TH, TWT, 1>(a, mt, stream);
#endif
    if (a.nchunk == 3 && tile3_lds_bytes<MB, STRIDE, DIL, TH, TWT, 3>(3) <= kLds3Budget)
        return launch_tile3<MB, STRIDE, DIL, TH, TWT, 3>(a, mt, stream);
    if ((a.nchunk == 2 || a.nchunk == 4) && tile3_lds_bytes<MB, STRIDE, DIL, TH, TWT, 2>(a.nchunk) <= kLds3Budget)
        return launch_tile3<MB, STRIDE, DIL, TH, TWT, 2>(a, mt, stream);
    return launch_tile3<MB, STRIDE, DIL, TH, TWT, 1>(a, mt, stream);
}

// tile shapes: 2 = 8 x 32 pixels (4 segments per wave; stride 1 only), 1 = 4 x 32 (2), 0 = 4 x 16 (1)
template <int MB, int STRIDE, int DIL>
static int launch_shape3(TileArgs& a, int mt, int shape, hipStream_t stream) {
    if (shape == 2) {
        if constexpr (STRIDE == 1 && MB < 3) return launch_cps3<MB, STRIDE, DIL, 8, 2>(a, mt, stream);
        else return launch_cps3<MB, STRIDE, DIL, 4, 2>(a, mt, stream);
    } else if (shape == 1) {
        return launch_cps3<MB, STRIDE, DIL, 4, 2>(a, mt, stream);
    }
    return launch_cps3<MB, STRIDE, DIL, 4, 1>(a, mt, stream);
}

template <int STRIDE, int DIL>
static int launch_mb3(TileArgs& a, int mt, int mb, int shape, hipStream_t stream) {
    if (mb == 3) return launch_shape3<3, STRIDE, DIL>(a, mt, shape, stream);
    if (mb == 2) return launch_shape3<2, STRIDE, DIL>(a, mt, shape, stream);
    return launch_shape3<1, STRIDE, DIL>(a, mt, shape, stream);
}

}  // namespace itermvs

using namespace itermvs;

// called from itermvs_conv2d (conv.hip) when weight_format == 3; returns 1 when the shape is not covered
int itermvs_conv2d_tile3(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream) {
    if (p->ksize != 3 || p->Cin <= 4) return 1;
    if (p->in_layout == 1 && !(p->Cin == 8 && p->stride == 1 && p->dilation == 1)) return 1;      // channels-last input: the tap-pair form only
    if (p->Cin <= 8 && !(p->stride == 1 && p->dilation == 1)) return 1;      // the tap-pair form: stride 1, no dilation
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
    a.nchunk = (p->Cin + 15) / 16;
    const int mt = a.CoutPad / 16;
    const int split_blocks = p->split_cout ? p->split_cout / 16 : 0;
    auto mb_ok = [&](int m) { return mt % m == 0 && (!split_blocks || split_blocks % m == 0); };
    // Measured sweep over all (tile shape, channel blocking) pairs on the layers of the path (tools/conv_bench.py --sweep3,
    // profiles/r05/r05e_conv_tile3_sweep.txt).  A wave wants MB * NB >= 4 accumulator tiles (operand reads per MFMA, see the
    // header) but the split weights of a wide channel block crowd the LDS (48 -> 48 at MB = 3: 124 KB), and every channel
    // block of a tile stages and splits the tile again:
    //   dilated layers (ConvGRU, heads: 20 480 pixels, few tiles)      4x32 tiles, one block per wave
    //   48 input channels (three chunks)                               8x32 tiles, one block
    //   16 output channels                                             8x32 tiles
    //   32 output channels                                             4x16 tiles, two blocks per wave
    //   64+ output channels                                            4x16 tiles, one block
    int shape = 1, mb = 1;
    if (p->dilation == 2) { shape = 1; mb = 1; }
    else if (a.nchunk >= 3 || mt == 1) { shape = p->stride == 1 ? 2 : 1; mb = 1; }
    else if (mt == 2 && mb_ok(2)) { shape = 0; mb = 2; }
    else { shape = 0; mb = 1; }
    if (p->Cin <= 8) {            // PAIR form (weights packed as five tap pairs: ops.MfmaWeight)
        if (p->split_cout) return 1;
        const bool dotp = p->act == 6 || p->act == 7;
        int rcp = 1;
        if (p->in_layout == 1) {            // channels-last input: exactly 8 channels, 16-byte aligned pixels (checked in itermvs_conv2d)
            if (mt == 1) rcp = launch_tile3_pair<1, 1>(a, mt, stream);
            else if (mt == 2 && dotp) rcp = launch_tile3_pair<2, 1>(a, mt, stream);
            else if (!dotp) rcp = launch_tile3_pair<1, 1>(a, mt, stream);
        } else if (mt == 1) rcp = launch_tile3_pair<1>(a, mt, stream);
        else if (mt == 2 && dotp) rcp = launch_tile3_pair<2>(a, mt, stream);
        else if (!dotp) rcp = launch_tile3_pair<1>(a, mt, stream);
        if (rcp != 0) return 1;
        return itermvs_launch_status();
    }
    const char* force = itermvs_tuning_env("ITERMVS_TILE3_FORCE");            // "shape,mb" (tools/conv_bench.py --sweep3)
    if (force) {
        shape = force[0] - '0';
        mb = force[2] - '0';
        if (shape < 0 || shape > 2 || mb < 1 || mb > 3 || !mb_ok(mb)) return 1;
    }
    const bool dot = p->act == 6 || p->act == 7;     // the epilogue contracts over ALL output channels: one block per wave
    if (dot) {
        mb = mt;
        if (mb > 3) return 1;
    }
    // the preferred (shape, channel blocking) first; when its stage + weights exceed the LDS budget (stride-2 halos, many
    // input channels): smaller tiles, then narrower channel blocks
    int rc = 1;
    for (; rc == 1 && mb >= 1; --mb) {
        if (dot && mb != mt) return 1;
        if (!mb_ok(mb)) continue;
        for (int sh = shape; rc == 1 && sh >= 0; --sh) {
            if (s1d1) rc = launch_mb3<1, 1>(a, mt, mb, sh, stream);
            else if (s2d1) rc = launch_mb3<2, 1>(a, mt, mb, sh, stream);
            else rc = launch_mb3<1, 2>(a, mt, mb, sh, stream);
            if (force) break;
        }
    }
    if (rc != 0) return 1;
    return itermvs_launch_status();
}
