// itermvs_gru_conv: the two dilated 3x3 convolutions of the ConvGRU (models/module.py:53-66, models/itermvs.py:131-137: 43 input
// channels = hidden 32 + normalised depth 1 + 10 scores, dilation 2) as cooperative kernels with their gate math in the epilogue:
//   mode 0  z = sigmoid(convz([h, x]))            -> z [B,32,H,W]                               module.py:61
//           r = sigmoid(convr([h, x])); r * h     -> rh [B,32,H,W] (the first 32 channels of the second GRU buffer)   :62-63
//   mode 1  q = tanh(convq([r*h, x]));  h' = (1 - z) h + z q   -> out / out2 [B,32,H,W]          module.py:64-65
// As launches of the LDS-tiled kernels (conv_tile fp32 for z/r, conv_tile3 for q) they cost 18.7 + 13.7 us per GRU iteration on a
// 128 x 160 map: 640 / 320 workgroups that each stage a 43-channel halo tile for 16 output channels.  Here -- the structure of
// head.hip's cooperative kernel -- a persistent 8-wave workgroup walks 16-pixel row segments; wave w = (output block w % NOB, tap
// group w / NOB), so every wave multiplies ALL input channels of its taps for one block of 16 output channels, and the tap groups'
// partial sums meet in LDS.  Arithmetic: the bf16x3 form of conv_tile3.hip (operands split exactly into three bf16 terms, the six
// largest cross products on v_mfma_f32_16x16x32_bf16, fp32 accumulation):
//   channels 0..31   K = 32 = the 32 channels of ONE term (a lane's eight slots j = channels (j / 4) * 16 + 4 q + j % 4):
//                    wl xh + wh xl + wm xm + wm xh + wh xm + wh xh, six MFMAs per tap;
//   channels 32..47  (11 used) K = 32 = two 16-channel terms side by side: A1 = [wh | wh], A2 = [wm | wm], A3 = [wl | wh],
//                    B1 = [xh | xm], B3 = [xh | xl], three MFMAs per tap.
// The tile (3 rows x 20 columns, rows y - 2, y, y + 2) is staged as bf16 triples by (q | half, row, column) items = eight plane
// loads, split, three 16-byte LDS stores; a wave's weights sit half in registers (terms h, m / A1, A2) and half in LDS (l / A3).
#include <stdlib.h>

#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kGcThreads = 512, kGcWaves = 8;
constexpr int kGcCols = 20;                                 // staged columns x0 - 2 .. x0 + 17
constexpr int kGcGrpB = kGcCols * 16;                       // one channel group of a row: [column][8 channels bf16]
constexpr int kGcTermB = 6 * kGcGrpB;                       // one term of a row: groups q = 0..3 (channels 0..31), 4 + half (32..47)
constexpr int kGcRowB = 3 * kGcTermB;                       // a staged image row: [term h, m, l][group][column][16 B] = 5 760 B
constexpr int kGcRing = 8, kGcSlots = 2 * kGcRing;          // two rings of eight rows
constexpr int kGcPB = 8 * 64 * 16;                          // partial sums of one tile: [wave = tap group x output block][lane][16 B]
constexpr int kGcLds = kGcSlots * kGcRowB + 4 * kGcPB + 64 * 4;   // 122.3 KB: rows, partial sums of two pairs of tiles, bias
constexpr uint32_t kGcOob = 0x40000000u;                    // byte offset past any input (43 H W 4 <= 2^29)

struct GruConvArgs {
    const float* x;          // [B,43,H,W] planes: [h | nd | scores] (mode 0) or [r*h | nd | scores] (mode 1)
    int64_t x_sb;
    const void* w;           // bf16 [output block][tap 9][operand 6: A h, m, l; B A1, A2, A3][lane 64][8]
    const float* bias;       // [16 * NOB]
    const float* h;          // hidden state [B,32,H,W] planes (mode 0: factor of r; mode 1: the state being updated)
    int64_t h_sb;
    const float* z;          // mode 1: update gate [B,32,H,W]
    int64_t z_sb;
    float* out;              // mode 0: z; mode 1: h'
    int64_t out_sb;
    float* out2;             // mode 0: r * h; mode 1: second copy of h' or nullptr
    int64_t out2_sb;
    int H, W, tiles_x;
    int run, run_extra;      // a workgroup's run of tiles: tiles / grid, the first tiles % grid workgroups one more
};

// A pair of tiles -- rows y and y + 2 of one column tile -- and where its image rows sit
struct GruPair {
    int b, xt, k;            // batch element, column tile, index of the first tile in the strip's row order (even rows, then odd rows)
    int y, set;              // image row of the first tile; ring of the pair's rows
    int fresh;               // 1 = none of its rows is staged yet (else rows y - 2 and y are the previous pair's last two)
    int has2;                // the second tile (row y + 2) exists
};

// NOB output blocks of 16 channels (4: the z and r gates, 2: the candidate state); tap groups = 8 / NOB.
// A workgroup owns a run of consecutive tiles of the order (batch, column tile, even rows ascending, odd rows ascending) and takes them
// two at a time: tiles y and y + 2 need the rows y - 2 .. y + 4, two of which the previous pair has left in LDS.  Rows live in rings
// of eight slots (slot = (row / 2) % 8); a pair that starts a new column of rows takes the other ring.
// One barrier per pair: between two barriers a wave does the matrix work of pair i (-> partial sums P[i % 2]), its share of the
// epilogue of pair i - 1 (from P[(i - 1) % 2]; mode 0 reads the hidden state back from the tiles' centre rows), the staging of
// pair i + 1's new rows and the loads of pair i + 2's.
template <int NOB>
__global__ void __launch_bounds__(kGcThreads) gru_conv_kernel(const GruConvArgs a, const int tiles_total, const int banded) {
    constexpr int NTG = kGcWaves / NOB;                      // tap groups: 2 or 4
    constexpr int MT = NTG == 2 ? 5 : 3;                     // taps of a wave at most
    constexpr int NV = 4 / NTG;                              // epilogue values of a lane: rows r0 .. r0 + NV - 1 of its D quad
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    char* __restrict__ Pp = gsm + kGcSlots * kGcRowB;
    float* __restrict__ BS = reinterpret_cast<float*>(Pp + 4 * kGcPB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int half = q & 1, second = q >> 1;
    const int ob = wave % NOB, tg = wave / NOB;
    // taps of this wave: NTG = 2: 0..4 | 5..8;  NTG = 4: 0..2 | 3,4 | 5,6 | 7,8
    const int tap0 = NTG == 2 ? tg * 5 : (tg == 0 ? 0 : 1 + 2 * tg);
    const int ntap = NTG == 2 ? (tg == 0 ? 5 : 4) : (tg == 0 ? 3 : 2);
    const uint32_t plane = (uint32_t)(a.H * a.W);
    const int He = (a.H + 1) >> 1;

    // this workgroup's run of tiles
    const int g = banded ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t0 = g * a.run + min(g, a.run_extra);           // (no 64-bit division in the prologue: the host splits the tile list)
    int left = a.run + (g < a.run_extra ? 1 : 0);             // tiles not yet given to a pair
    auto row_of = [&](int k) { return k < He ? 2 * k : 2 * (k - He) + 1; };
    auto second_ok = [&](int k) { return left > 1 && k + 1 != He && k + 1 != a.H; };
    auto advance = [&](GruPair t) {             // the pair after t; takes its tiles from ``left``
        t.k += 1 + t.has2;
        if (t.k == a.H) {
            t.k = 0;
            if (++t.xt == a.tiles_x) { t.xt = 0; ++t.b; }
        }
        t.fresh = t.k == 0 || t.k == He;
        t.set = t.fresh ? t.set ^ 1 : t.set;
        t.y = row_of(t.k);
        t.has2 = second_ok(t.k);
        left -= 1 + t.has2;
        return t;
    };
    GruPair cur;
    {
        const int strip = t0 / a.H;
        cur.k = t0 - strip * a.H;
        cur.b = strip / a.tiles_x;
        cur.xt = strip - cur.b * a.tiles_x;
        cur.y = row_of(cur.k);
        cur.set = 0;
        cur.fresh = 1;
        cur.has2 = second_ok(cur.k);
        left -= 1 + cur.has2;
    }

    // staging: thread (channel pair p, column) of a row; p = 4 group + jj, jj = word of the group's 16 bytes
    const bool it_on = tid < 24 * kGcCols;
    const int it_p = tid / kGcCols, it_col = tid - it_p * kGcCols;
    const int it_grp = it_p >> 2, it_jj = it_p & 3;
    const int it_c0 = it_grp < 4 ? (it_jj >> 1) * 16 + 4 * it_grp + 2 * (it_jj & 1) : 32 + 8 * (it_grp - 4) + 2 * it_jj;
    const uint32_t cb0 = it_on && it_c0 < 43 ? ((uint32_t)it_c0 * plane + (uint32_t)(it_col - 2)) * 4u : kGcOob;
    const uint32_t cb1 = it_on && it_c0 + 1 < 43 ? ((uint32_t)(it_c0 + 1) * plane + (uint32_t)(it_col - 2)) * 4u : kGcOob;
    const int it_lds = it_grp * kGcGrpB + it_col * 16 + it_jj * 4;
    float st[4][2];
    // rows j = 0..3 of a pair = image rows y - 2, y, y + 2, y + 4; the ones it has to stage itself:
    auto is_new = [&](const GruPair& t, int j) { return t.fresh ? (j < 3 || t.has2) : (j == 2 || (j == 3 && t.has2)); };
    auto fetch = [&](const GruPair& t) {          // -> registers
        const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)t.b * a.x_sb), 0, (int)(43u * plane * 4u), 0x00020000);
        const int x0 = t.xt * 16, gx = x0 + it_col - 2;
        const uint32_t cbig = (gx >= 0 && gx < a.W) ? 0u : kGcOob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (is_new(t, j)) {                    // uniform
                const int gy = t.y + 2 * (j - 1);
                const uint32_t ro = (gy >= 0 && gy < a.H) ? (uint32_t)(gy * a.W + x0) * 4u + cbig : kGcOob + cbig;
                st[j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ir, cb0 + ro, 0, 0));
                st[j][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ir, cb1 + ro, 0, 0));
            }
        }
    };
    auto slot_of = [&](const GruPair& t, int j) { return t.set * kGcRing + (((t.y >> 1) + j - 1) & (kGcRing - 1)); };
    auto stash = [&](const GruPair& t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (is_new(t, j) && it_on) {
                uint32_t h, m, l;
                split_pair(st[j][0], st[j][1], h, m, l);
                char* __restrict__ d = gsm + slot_of(t, j) * kGcRowB + it_lds;
                *reinterpret_cast<uint32_t*>(d) = h;
                *reinterpret_cast<uint32_t*>(d + kGcTermB) = m;
                *reinterpret_cast<uint32_t*>(d + 2 * kGcTermB) = l;
            }
        }
    };

    fetch(cur);                                   // in flight together with the weights

    // this wave's weights, all in registers: terms h, m, l of channels 0..31 and A1, A2, A3 of channels 32..47
    bf8 wah[MT], wam[MT], wal[MT], wb1[MT], wb2[MT], wb3[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int tap = min(tap0 + min(k, ntap - 1), 8);
        const bf8* __restrict__ src = reinterpret_cast<const bf8*>(a.w) + ((ob * 9 + tap) * 6) * 64 + lane;
        wah[k] = src[0];
        wam[k] = src[64];
        wal[k] = src[128];
        wb1[k] = src[192];
        wb2[k] = src[256];
        wb3[k] = src[320];
    }
    if (tid < 16 * NOB) BS[tid] = a.bias ? a.bias[tid] : 0.0f;
    stash(cur);
    bool more1 = left > 0;                        // a pair after cur / after n1 exists
    GruPair n1 = cur, n2 = cur;
    if (more1) {
        n1 = advance(cur);
        fetch(n1);
    }

    // epilogue share of this wave: rows r0 .. r0 + NV - 1 of output block ob's D quads (lane = 16 q + pixel: channel 16 ob + 4 q + r)
    const int r0 = tg * NV;
    float bsv[NV];
    float hv[2][NV], zv[2][NV];                   // mode 1: state and update gate of the pair, requested one iteration ahead
    auto epilogue = [&](const GruPair& t, const char* __restrict__ P) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 0 || t.has2) {
                float v[NV];
#pragma unroll
                for (int r = 0; r < NV; ++r) v[r] = 0.0f;
#pragma unroll
                for (int gi = 0; gi < NTG; ++gi) {        // the tap groups' partial sums in a fixed order
                    const float* __restrict__ pp = reinterpret_cast<const float*>(P + e * kGcPB + ((gi * NOB + ob) * 64 + lane) * 16) + r0;
#pragma unroll
                    for (int r = 0; r < NV; ++r) v[r] += pp[r];
                }
                const int px = t.xt * 16 + l16;
                const size_t pix = (size_t)(t.y + 2 * e) * a.W + px;
                if constexpr (NOB == 4) {
                    // mode 0: the hidden state is in the tile's centre row -- channel 16 (ob - 2) + 4 q + r = slot 4 (ob - 2) + r of
                    // group q, column 2 + pixel; the three terms add up to the fp32 value exactly (8 + 8 + 8 significant bits)
                    if (ob >= 2) {
                        const char* __restrict__ hp = gsm + slot_of(t, 1 + e) * kGcRowB + q * kGcGrpB + (2 + l16) * 16 + (ob - 2) * 8 + tg * 4;
                        const uint32_t th = *reinterpret_cast<const uint32_t*>(hp), tm = *reinterpret_cast<const uint32_t*>(hp + kGcTermB);
                        const uint32_t tl = *reinterpret_cast<const uint32_t*>(hp + 2 * kGcTermB);
                        hv[e][0] = (__uint_as_float(th << 16) + __uint_as_float(tm << 16)) + __uint_as_float(tl << 16);      // even slot = low half
                        hv[e][1] = (__uint_as_float(th & 0xffff0000u) + __uint_as_float(tm & 0xffff0000u)) + __uint_as_float(tl & 0xffff0000u);
                    }
                }
                if (px < a.W) {
#pragma unroll
                    for (int r = 0; r < NV; ++r) {
                        const float u = v[r] + bsv[r];
                        const int c = ob * 16 + q * 4 + r0 + r;                  // output channel of the layer
                        if constexpr (NOB == 4) {
                            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
                            if (ob < 2) a.out[(int64_t)t.b * a.out_sb + (size_t)c * plane + pix] = sg;                            // z
                            else a.out2[(int64_t)t.b * a.out2_sb + (size_t)(c - 32) * plane + pix] = sg * hv[e][r];               // r * h
                        } else {
                            const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));                        // tanh
                            const float hn = (1.0f - zv[e][r]) * hv[e][r] + zv[e][r] * th;                                        // module.py:64-65
                            a.out[(int64_t)t.b * a.out_sb + (size_t)c * plane + pix] = hn;
                            if (a.out2) a.out2[(int64_t)t.b * a.out2_sb + (size_t)c * plane + pix] = hn;
                        }
                    }
                }
            }
        }
    };

    const int la = q * kGcGrpB + l16 * 16;                                     // channels 0..31: term h of this lane's operand
    const int lb = (4 + half) * kGcGrpB + l16 * 16;                            // channels 32..47
    const int lb1 = lb + (second ? kGcTermB : 0), lb3 = lb + (second ? 2 * kGcTermB : 0);     // [xh | xm], [xh | xl]
    GruPair prev = cur;
    int it = 0;
    for (bool more = true; more; ++it) {
        __syncthreads();            // this pair's rows and the previous pair's partial sums are visible
        if (it == 0) {
#pragma unroll
            for (int r = 0; r < NV; ++r) bsv[r] = BS[ob * 16 + q * 4 + r0 + r];
        }
        f32x4 acc[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[e][0] = acc[e][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            if (k < ntap) {                  // wave-uniform
                const int tap = tap0 + k;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (e == 0 || cur.has2) {
                        const char* __restrict__ R = gsm + slot_of(cur, ky + e) * kGcRowB + kx * 32;
                        const bf8 xh = *reinterpret_cast<const bf8*>(R + la), xm = *reinterpret_cast<const bf8*>(R + la + kGcTermB);
                        const bf8 xl = *reinterpret_cast<const bf8*>(R + la + 2 * kGcTermB);
                        const bf8 b1 = *reinterpret_cast<const bf8*>(R + lb1), b3 = *reinterpret_cast<const bf8*>(R + lb3);
                        acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wal[k], xh, acc[e][0], 0, 0, 0);
                        acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wah[k], xl, acc[e][1], 0, 0, 0);
                        acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb3[k], b3, acc[e][0], 0, 0, 0);
                        acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wam[k], xm, acc[e][1], 0, 0, 0);
                        acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wam[k], xh, acc[e][0], 0, 0, 0);
                        acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wah[k], xm, acc[e][1], 0, 0, 0);
                        acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb2[k], b1, acc[e][0], 0, 0, 0);
                        acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wah[k], xh, acc[e][1], 0, 0, 0);
                        acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb1[k], b1, acc[e][0], 0, 0, 0);
                    }
                }
            }
        }
        char* __restrict__ Pw = Pp + (it & 1) * 2 * kGcPB + (wave * 64 + lane) * 16;                      // wave = tg * NOB + ob
        *reinterpret_cast<f32x4*>(Pw) = acc[0][0] + acc[0][1];
        if (cur.has2) *reinterpret_cast<f32x4*>(Pw + kGcPB) = acc[1][0] + acc[1][1];
        // the previous pair's epilogue: its partial sums were completed before this iteration's barrier
        if (it > 0) epilogue(prev, Pp + ((it & 1) ^ 1) * 2 * kGcPB);
        if constexpr (NOB == 2) {            // this pair's epilogue operands (consumed one iteration later)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (e == 0 || cur.has2) {
                    const size_t pix = (size_t)(cur.y + 2 * e) * a.W + min(cur.xt * 16 + l16, a.W - 1);
#pragma unroll
                    for (int r = 0; r < NV; ++r) {
                        const int c = ob * 16 + q * 4 + r0 + r;
                        hv[e][r] = a.h[(int64_t)cur.b * a.h_sb + (size_t)c * plane + pix];
                        zv[e][r] = a.z[(int64_t)cur.b * a.z_sb + (size_t)c * plane + pix];
                    }
                }
            }
        }
        // the next pair's new rows (fetched one iteration ago) go to their slots, those of the pair after it into registers
        more = more1;
        if (more1) {
            // a pair that opens a ring right after one that did returns to the ring the previous pair's epilogue is still reading
            if (NOB == 4 && it > 0 && n1.fresh && cur.fresh) __syncthreads();
            stash(n1);
            more1 = left > 0;
            if (more1) {
                n2 = advance(n1);
                fetch(n2);
            }
        }
        prev = cur;
        cur = n1;
        n1 = n2;
    }
    __syncthreads();
    epilogue(prev, Pp + ((it & 1) ^ 1) * 2 * kGcPB);
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_gru_conv(const float* x, int64_t x_sb, int32_t B, int32_t H, int32_t W, int32_t mode, const void* w_packed,
                                const float* bias, const float* h, int64_t h_sb, const float* z, int64_t z_sb, float* out,
                                int64_t out_sb, float* out2, int64_t out2_sb, void* stream) {
    ITERMVS_RETURN_IF(!x || !w_packed || !h || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(mode != 0 && mode != 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((mode == 0 && !out2) || (mode == 1 && !z), ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || H < 1 || W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((int64_t)43 * H * W * 4 > ((int64_t)1 << 29), ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(((uintptr_t)w_packed) % 16, ITERMVS_ERR_ALIGN);
    GruConvArgs a;
    a.x = x; a.x_sb = x_sb; a.w = w_packed; a.bias = bias; a.h = h; a.h_sb = h_sb; a.z = z; a.z_sb = z_sb;
    a.out = out; a.out_sb = out_sb; a.out2 = out2; a.out2_sb = out2_sb;
    a.H = H; a.W = W; a.tiles_x = (W + 15) / 16;
    const int64_t tiles = (int64_t)a.tiles_x * H * B;
    ITERMVS_RETURN_IF(tiles > 0x7fffffff, ITERMVS_ERR_DIMS);
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(gru_conv_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kGcLds) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(gru_conv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kGcLds) == hipSuccess;
    ITERMVS_RETURN_IF(!attr_ok, ITERMVS_ERR_LAUNCH);
    const int cus = itermvs_num_cus();                       // one 8-wave workgroup per CU (256 registers per lane)
    int grid = (int)(tiles < cus ? tiles : cus);
    if (grid >= 16) grid &= ~7;
    a.run = (int)(tiles / grid); a.run_extra = (int)(tiles % grid);
    const int banded = grid % 8 == 0;     // workgroup b runs on XCD b % 8: neighbouring runs of tiles on one XCD
    itermvs_profile_begin(3, (hipStream_t)stream);          // bench.py's convolution roofline brackets this launch like an itermvs_conv2d one
    if (mode == 0) hipLaunchKernelGGL(gru_conv_kernel<4>, dim3(grid), dim3(kGcThreads), kGcLds, (hipStream_t)stream, a, (int)tiles, banded);
    else hipLaunchKernelGGL(gru_conv_kernel<2>, dim3(grid), dim3(kGcThreads), kGcLds, (hipStream_t)stream, a, (int)tiles, banded);
    itermvs_profile_end(3, (hipStream_t)stream);
    return itermvs_launch_status();
}
