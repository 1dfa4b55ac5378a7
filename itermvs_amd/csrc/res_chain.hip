// itermvs_res_chain16: the three 3x3 16 -> 16 convolutions of FeatureNet's half-resolution stage in ONE launch
// (models/net.py:13,40 `layer1`; models/module.py:33-50 ResidualBlock), BatchNorm folded:
//     a = relu(conv(y1; Wa) + ba + ds)        layer1[0].conv2 + the down-sampling shortcut   (y1, ds: the stem's two results)
//     b = relu(conv(a;  Wb) + bb)             layer1[1].conv1
//     c = relu(conv(b;  Wc) + bc + a)         layer1[1].conv2 + skip                          = fea1
// As three launches of conv_tile3 these layers cost 24-32 us each at cfg 1 (5 x 16 x 256 x 320 maps): 5.6 us of launch / weight
// staging / barrier floor, and a 26 MB write + 26 MB read between consecutive layers (profiles/r05/r05ac_*).  Here a persistent
// 8-wave workgroup owns a TH x TW tile of c and walks the chain with a and b in LDS, recomputing the halo (a on
// (TH+4) x (TW+4), b on (TH+2) x (TW+2): 1.69x / 1.33x the positions of an 8 x 32 tile); positions outside the image are stored
// as zeros, so every layer sees the zero padding the reference's layer sees.
//
// Arithmetic: the bf16x3 form of conv_tile3.hip (exact three-term split of both operands, the six largest of the nine cross
// products on v_mfma_f32_16x16x32_bf16, fp32 accumulation; error of the size of one fp32 rounding per product):
//     A1 = [wh | wh]  A2 = [wm | wm]  A3 = [wl | wh]      B1 = [xh | xm]  B3 = [xh | xl]      acc += A3 B3 + A2 B1 + A1 B1
// The three layers' split weights stay in LDS for the whole launch (41 KB); a wave takes every 8th group of 16 consecutive
// positions (row-major over the layer's region) and reads three A operands per tap and two 16-byte B operands per tap and group.  Activations are split ONCE, when they are produced: LDS tiles [plane h, m, l][half][position][8 bf16] like
// conv_tile3's.  The skip of the third layer is a = h + m + l rebuilt exactly from its LDS tile.
#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int kRcThreads = 512, kRcWaves = 8;
constexpr uint32_t kRcOob = 0x7fffffffu;
constexpr int kRcWL = 9 * 3 * 16 * 32;           // one layer's split weights: [tap][plane h,m,l][16 rows][16 ch bf16] = 13 824 B

struct ResChainArgs {
    const float* y1;
    const float* ds;
    float* out;
    int64_t y1_sn, ds_sn, out_sn;
    const void* w[3];
    const float* bias[3];
    int N, H, W, tiles_x, tiles_y, total, banded;
};

constexpr int rc_pad256(int b) { return (b + 255) / 256 * 256; }

template <int TH, int TW>
struct RcGeom {
    static constexpr int YW = TW + 6, YH = TH + 6, YPX = YW * YH;
    static constexpr int AW = TW + 4, AH = TH + 4, APX = AW * AH;
    static constexpr int BW = TW + 2, BH = TH + 2, BPX = BW * BH;
    static constexpr int CPX = TW * TH;
    static constexpr int YPLB = rc_pad256(YPX * 16), APLB = rc_pad256(APX * 16), BPLB = rc_pad256(BPX * 16);
    static constexpr int LDS = 6 * YPLB + 6 * APLB + 3 * kRcWL + 3 * 16 * 4;
    static_assert(6 * BPLB <= 6 * YPLB, "b reuses the y1 tile");
    static constexpr int NIT = (2 * YPX + kRcThreads - 1) / kRcThreads;      // (half, position) staging items per thread
};

// one layer's MFMA loop: groups wave, wave + 8, ... of 16 consecutive output positions; acc starts from the bias.  NB = the
// number of groups THIS wave owns (the caller branches once, wave-uniformly, between the two possible counts: a guard per
// group turns every MFMA into a basic block of its own).  Per tap: three A operands (the layer's split weights in LDS,
// [tap][plane][16 rows][32 B]) and two B operands per group; the operands of tap u + 1 are read before the MFMAs of tap u
// (two register sets, the order pinned by sched_barrier -- holding a layer's 27 A operands in registers instead left the
// compiler no room for the second set and every tap waited for its reads: 76 of 121 us, profiles/r06/r06j_*).
template <int INW, int INPLB, int OUTW, int OUTPX, int NBM, int NB>
__device__ __forceinline__ void rc_layer_n(const char* __restrict__ in, const char* __restrict__ wl, f32x4 (&acc)[NBM], int wave,
                                           int l16, int half, int second) {
    const char* __restrict__ x1 = in + (second ? 2 * INPLB : 0) + half * INPLB;      // B1 = [xh | xm]
    const char* __restrict__ x3 = in + (second ? 4 * INPLB : 0) + half * INPLB;      // B3 = [xh | xl]
    const char* __restrict__ wa = wl + l16 * 32 + half * 16;                         // A1 = [wh | wh]; A2 = + 512
    const char* __restrict__ wa3 = wa + (second ? 0 : 1024);                         // A3 = [wl | wh]
    int off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pos = min((wave + nb * kRcWaves) * 16 + l16, OUTPX - 1);           // surplus lanes recompute the last position, never store
        const int oy = pos / OUTW, ox = pos - oy * OUTW;
        off[nb] = (oy * INW + ox) * 16;
    }
    bf8 a1[2], a2[2], a3[2], b1[2][NB], b3[2][NB];
    auto read = [&](int tap, int set) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int to = (ky * INW + kx) * 16;
        a1[set] = *reinterpret_cast<const bf8*>(wa + tap * 1536);
        a2[set] = *reinterpret_cast<const bf8*>(wa + tap * 1536 + 512);
        a3[set] = *reinterpret_cast<const bf8*>(wa3 + tap * 1536);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            b1[set][nb] = *reinterpret_cast<const bf8*>(x1 + off[nb] + to);
            b3[set][nb] = *reinterpret_cast<const bf8*>(x3 + off[nb] + to);
        }
    };
    read(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int s = tap & 1;
        if (tap + 1 < 9) read(tap + 1, s ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[s], b3[s][nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s], b1[s][nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s], b1[s][nb], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int INW, int INPLB, int OUTW, int OUTPX, int NBM>
__device__ __forceinline__ void rc_layer(const char* __restrict__ in, const char* __restrict__ wl, f32x4 (&acc)[NBM], int wave,
                                         int l16, int half, int second) {
    constexpr int G = (OUTPX + 15) / 16;
    constexpr int FULL = G % kRcWaves;             // waves below FULL own NBM groups, the others NBM - 1 (0: every wave owns NBM)
    if constexpr (FULL == 0) {
        rc_layer_n<INW, INPLB, OUTW, OUTPX, NBM, NBM>(in, wl, acc, wave, l16, half, second);
    } else {
        if (wave < FULL) rc_layer_n<INW, INPLB, OUTW, OUTPX, NBM, NBM>(in, wl, acc, wave, l16, half, second);
        else rc_layer_n<INW, INPLB, OUTW, OUTPX, NBM, NBM - 1>(in, wl, acc, wave, l16, half, second);
    }
}

// four channels (q*4 .. q*4+3) of one position -> the three LDS tiles (8 bytes each)
__device__ __forceinline__ void rc_store_split(char* __restrict__ tile, int plb, int q, int pos, float v0, float v1, float v2, float v3) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split_pair(v0, v1, h0, m0, l0);
    split_pair(v2, v3, h1, m1, l1);
    char* __restrict__ d = tile + (q >> 1) * plb + pos * 16 + (q & 1) * 8;
    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(d + 2 * plb) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(d + 4 * plb) = u32x2{l0, l1};
}

// C4: y1 and ds arrive as channel quads [N][4][H][W][4] (itermvs_stem's out_layout 1): an item's 8 channels are two 16-byte
// loads, a lane's four shortcut channels one -- 60 wave-level load instructions per tile instead of 241 (a wave-level load
// costs the CU ~64 cycles whatever its width; with plane loads they were 47 of the launch's 121 us, profiles/r06/r06j_*)
template <int TH, int TW, bool C4>
__global__ void __launch_bounds__(kRcThreads) res_chain16_kernel(const ResChainArgs a) {
    using G = RcGeom<TH, TW>;
    constexpr int NIT = G::NIT;
    constexpr int NBA = ((G::APX + 15) / 16 + kRcWaves - 1) / kRcWaves;
    constexpr int NBB = ((G::BPX + 15) / 16 + kRcWaves - 1) / kRcWaves;
    constexpr int NBC = ((G::CPX + 15) / 16 + kRcWaves - 1) / kRcWaves;
    extern __shared__ __attribute__((aligned(16))) char rc_smem[];
    char* __restrict__ Yt = rc_smem;
    char* __restrict__ At = rc_smem + 6 * G::YPLB;
    char* __restrict__ Wt = At + 6 * G::APLB;
    char* __restrict__ Bt = Yt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int half = q & 1, second = q >> 1;
    const uint32_t plane = (uint32_t)(a.H * a.W);

    // the three layers' split weights -> LDS, once per workgroup (global layout == LDS layout for one 16-channel chunk)
    for (int i = tid; i < 3 * kRcWL / 16; i += kRcThreads) {
        const int layer = i / (kRcWL / 16), within = i - layer * (kRcWL / 16);
        reinterpret_cast<u32x4*>(Wt)[i] = reinterpret_cast<const u32x4*>(a.w[layer])[within];
    }
    float* __restrict__ Bs = reinterpret_cast<float*>(Wt + 3 * kRcWL);      // the three biases, [layer][16]
    if (tid < 48) Bs[tid] = a.bias[tid >> 4] ? a.bias[tid >> 4][tid & 15] : 0.0f;

    // staging items of a thread: (half of the 16 channels, position of the y1 tile): 8 plane loads, three 16-byte LDS stores
    struct Work { int n, oy0, ox0; };
    auto decode = [&](int w) {
        Work k;
        const int t2 = w / a.tiles_x;
        k.n = t2 / a.tiles_y;
        k.oy0 = (t2 - k.n * a.tiles_y) * TH;
        k.ox0 = (w - t2 * a.tiles_x) * TW;
        return k;
    };
    float stage[NIT][8];
    // item j of tile k's staging loads.  The next tile's items are issued one per layer: issued together they fill the CU's
    // address path and the issuing waves stand at the instruction while the others wait at the next barrier (seen in
    // lat_conv.hip's phase stamps, profiles/r06/r06k_*)
    auto fetch_item = [&](const Work& k, int j) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ir =
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.y1 + (int64_t)k.n * a.y1_sn), 0, (int)(16u * plane * 4u), 0x00020000);
        const int iy0 = k.oy0 - 3, ix0 = k.ox0 - 3;
        {
            const int item = tid + j * kRcThreads;
            const int hf = item >= G::YPX ? 1 : 0, px = item - hf * G::YPX;
            const int y = px / G::YW, x = px - y * G::YW;
            const int gy = iy0 + y, gx = ix0 + x;
            const bool ok = item < 2 * G::YPX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            if constexpr (C4) {
                const uint32_t go = ok ? ((uint32_t)(hf * 2) * plane + (uint32_t)(gy * a.W + gx)) * 16u : kRcOob;
#pragma unroll
                for (int c4 = 0; c4 < 2; ++c4) {
                    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ir, go, (uint32_t)c4 * plane * 16u, 0));
#pragma unroll
                    for (int c = 0; c < 4; ++c) stage[j][c4 * 4 + c] = __uint_as_float(v[c]);
                }
            } else {
                const uint32_t go = ok ? ((uint32_t)(hf * 8) * plane + (uint32_t)(gy * a.W + gx)) * 4u : kRcOob;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    stage[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ir, go, (uint32_t)c * plane * 4u, 0));
            }
        }
    };

    static_assert(NIT == 3, "one staging item per layer");
    // Tile order: workgroup b runs on XCD b % 8 (round-robin dispatch); the tiles of one XCD are a contiguous run of the tile list
    // (whole image bands): tiles that share halo rows (a tile's inputs are 2.1x / 1.7x its outputs) meet in the same 4 MB L2
    // instead of being fetched from the fabric once per XCD (HBM traffic 149 MB per launch in plain order for 78 MB of tensors)
    int w, wstep, wend;
    if (a.banded) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        w = (int)((int64_t)a.total * xcd / 8) + slot;
        wend = (int)((int64_t)a.total * (xcd + 1) / 8);
        wstep = gridDim.x >> 3;
    } else {
        w = blockIdx.x;
        wend = a.total;
        wstep = gridDim.x;
    }
    if (w >= wend) return;
    Work cur = decode(w);
#pragma unroll
    for (int j = 0; j < NIT; ++j) fetch_item(cur, j);
    while (true) {
        // ---- y1 tile: split, three planes to LDS ----
#pragma unroll
        for (int j = 0; j < NIT; ++j)
            if (j < NIT - 1 || tid + j * kRcThreads < 2 * G::YPX) {
                u32x4 Hh, Mm, Ll;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t h, m, l;
                    split_pair(stage[j][2 * k], stage[j][2 * k + 1], h, m, l);
                    Hh[k] = h; Mm[k] = m; Ll[k] = l;
                }
                const int item = tid + j * kRcThreads;
                const int hf = item >= G::YPX ? 1 : 0;
                char* __restrict__ d = Yt + hf * G::YPLB + (item - hf * G::YPX) * 16;
                *reinterpret_cast<u32x4*>(d) = Hh;
                *reinterpret_cast<u32x4*>(d + 2 * G::YPLB) = Mm;
                *reinterpret_cast<u32x4*>(d + 4 * G::YPLB) = Ll;
            }
        __syncthreads();
        const int wn = w + wstep;
        Work nxt = cur;
        if (wn < wend) nxt = decode(wn);

        // ---- a = relu(conv(y1) + ba + ds) on (TH+4) x (TW+4), origin (oy0 - 2, ox0 - 2) ----
        {
            const __amdgpu_buffer_rsrc_t dr =
                __builtin_amdgcn_make_buffer_rsrc((void*)(a.ds + (int64_t)cur.n * a.ds_sn), 0, (int)(16u * plane * 4u), 0x00020000);
            float dsv[NBA][4];
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) {
                const int pos = (wave + nb * kRcWaves) * 16 + l16;
                const int pc = min(pos, G::APX - 1);
                const int oy = pc / G::AW, ox = pc - oy * G::AW;
                const int gy = cur.oy0 - 2 + oy, gx = cur.ox0 - 2 + ox;
                const bool in = pos < G::APX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                if constexpr (C4) {
                    const uint32_t go = in ? ((uint32_t)q * plane + (uint32_t)(gy * a.W + gx)) * 16u : kRcOob;
                    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, go, 0, 0));
#pragma unroll
                    for (int r = 0; r < 4; ++r) dsv[nb][r] = __uint_as_float(v[r]);
                } else {
                    const uint32_t go = in ? ((uint32_t)(q * 4) * plane + (uint32_t)(gy * a.W + gx)) * 4u : kRcOob;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dsv[nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, go, (uint32_t)r * plane * 4u, 0));
                }
            }
            if (wn < wend) fetch_item(nxt, 0);      // (behind the shortcut's loads: memory returns in order; the staging registers are free once the y1 tile is split)
            f32x4 acc[NBA];
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) acc[nb] = *reinterpret_cast<const f32x4*>(Bs + 0 + q * 4);
            rc_layer<G::YW, G::YPLB, G::AW, G::APX, NBA>(Yt, Wt, acc, wave, l16, half, second);
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) {
                const int pos = (wave + nb * kRcWaves) * 16 + l16;
                if (pos < G::APX) {
                    const int oy = pos / G::AW, ox = pos - oy * G::AW;
                    const int gy = cur.oy0 - 2 + oy, gx = cur.ox0 - 2 + ox;
                    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(acc[nb][r] + dsv[nb][r], 0.0f) : 0.0f;
                    rc_store_split(At, G::APLB, q, pos, v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();
        if (wn < wend) fetch_item(nxt, 1);

        // ---- b = relu(conv(a) + bb) on (TH+2) x (TW+2), origin (oy0 - 1, ox0 - 1); overwrites the y1 tile ----
        {
            f32x4 acc[NBB];
#pragma unroll
            for (int nb = 0; nb < NBB; ++nb) acc[nb] = *reinterpret_cast<const f32x4*>(Bs + 16 + q * 4);
            rc_layer<G::AW, G::APLB, G::BW, G::BPX, NBB>(At, Wt + kRcWL, acc, wave, l16, half, second);
#pragma unroll
            for (int nb = 0; nb < NBB; ++nb) {
                const int pos = (wave + nb * kRcWaves) * 16 + l16;
                if (pos < G::BPX) {
                    const int oy = pos / G::BW, ox = pos - oy * G::BW;
                    const int gy = cur.oy0 - 1 + oy, gx = cur.ox0 - 1 + ox;
                    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(acc[nb][r], 0.0f) : 0.0f;
                    rc_store_split(Bt, G::BPLB, q, pos, v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();
        if (wn < wend) fetch_item(nxt, 2);

        // ---- c = relu(conv(b) + bc + a) on TH x TW -> global planes ----
        {
            f32x4 acc[NBC];
#pragma unroll
            for (int nb = 0; nb < NBC; ++nb) acc[nb] = *reinterpret_cast<const f32x4*>(Bs + 32 + q * 4);
            rc_layer<G::BW, G::BPLB, TW, G::CPX, NBC>(Bt, Wt + 2 * kRcWL, acc, wave, l16, half, second);
            float* __restrict__ ob = a.out + (int64_t)cur.n * a.out_sn + (int64_t)(q * 4) * plane;
#pragma unroll
            for (int nb = 0; nb < NBC; ++nb) {
                const int pos = (wave + nb * kRcWaves) * 16 + l16;
                if (pos < G::CPX) {
                    const int oy = pos / TW, ox = pos - oy * TW;
                    const int gy = cur.oy0 + oy, gx = cur.ox0 + ox;
                    if (gy < a.H && gx < a.W) {
                        // the skip: a = h + m + l, exact (8 + 8 + 8 significant bits)
                        const char* __restrict__ s = At + (q >> 1) * G::APLB + ((oy + 2) * G::AW + ox + 2) * 16 + (q & 1) * 8;
                        const u32x2 hh = *reinterpret_cast<const u32x2*>(s);
                        const u32x2 mm = *reinterpret_cast<const u32x2*>(s + 2 * G::APLB);
                        const u32x2 ll = *reinterpret_cast<const u32x2*>(s + 4 * G::APLB);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t hw = r < 2 ? hh[0] : hh[1], mw = r < 2 ? mm[0] : mm[1], lw = r < 2 ? ll[0] : ll[1];
                            const float ah = __uint_as_float((r & 1) ? (hw & 0xffff0000u) : (hw << 16));
                            const float am = __uint_as_float((r & 1) ? (mw & 0xffff0000u) : (mw << 16));
                            const float al = __uint_as_float((r & 1) ? (lw & 0xffff0000u) : (lw << 16));
                            const float av = (ah + am) + al;
                            ob[(int64_t)r * plane + (int64_t)gy * a.W + gx] = fmaxf(acc[nb][r] + av, 0.0f);
                        }
                    }
                }
            }
        }
        if (wn >= wend) break;
        __syncthreads();            // the b tile (= the next y1 tile's place) and the a tile are free
        w = wn;
        cur = nxt;
    }
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_res_chain16(const float* y1, int64_t y1_sn, const float* shortcut, int64_t shortcut_sn, int32_t in_layout,
                                   int32_t N, int32_t H, int32_t W, const void* const* weights, const float* const* bias, float* out,
                                   int64_t out_sn, void* stream) {
    ITERMVS_RETURN_IF(!y1 || !shortcut || !weights || !bias || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(N < 1 || H < 1 || W < 1 || H > 4095 || W > 4095, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((int64_t)16 * H * W * 4 >= ((int64_t)1 << 31), ITERMVS_ERR_DIMS);       // 32-bit byte offsets inside one image
    ITERMVS_RETURN_IF(in_layout != 0 && in_layout != 1, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(in_layout == 1 && ((((uintptr_t)y1 | (uintptr_t)shortcut) % 16) || y1_sn % 4 || shortcut_sn % 4), ITERMVS_ERR_ALIGN);
    constexpr int TH = 8, TW = 32;
    using G = RcGeom<TH, TW>;
    ResChainArgs a;
    a.y1 = y1; a.ds = shortcut; a.out = out; a.y1_sn = y1_sn; a.ds_sn = shortcut_sn; a.out_sn = out_sn;
    for (int l = 0; l < 3; ++l) {
        ITERMVS_RETURN_IF(!weights[l], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(((uintptr_t)weights[l]) % 16, ITERMVS_ERR_ALIGN);
        a.w[l] = weights[l];
        a.bias[l] = bias[l];
    }
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
    const int64_t total = (int64_t)N * a.tiles_x * a.tiles_y;
    ITERMVS_RETURN_IF(total >= ((int64_t)1 << 31), ITERMVS_ERR_DIMS);
    a.total = (int)total;
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(res_chain16_kernel<TH, TW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(res_chain16_kernel<TH, TW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
    ITERMVS_RETURN_IF(!attr_ok, ITERMVS_ERR_LAUNCH);
    const int cus = itermvs_num_cus();
    const int grid = a.total < cus ? a.total : cus;
    a.banded = grid % 8 == 0 && a.total >= grid ? 1 : 0;          // (grid < 8 or ragged: plain order)
    itermvs_profile_begin(3, (hipStream_t)stream);          // bench.py's convolution roofline brackets this launch like an itermvs_conv2d one
    if (in_layout) hipLaunchKernelGGL((res_chain16_kernel<TH, TW, true>), dim3(grid), dim3(kRcThreads), G::LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((res_chain16_kernel<TH, TW, false>), dim3(grid), dim3(kRcThreads), G::LDS, (hipStream_t)stream, a);
    itermvs_profile_end(3, (hipStream_t)stream);
    return itermvs_launch_status();
}
