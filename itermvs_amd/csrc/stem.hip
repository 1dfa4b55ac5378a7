// itermvs_stem: the first two layers of FeatureNet in ONE launch (models/net.py:13-14,39-40 with models/module.py:33-50):
//
//   f0 = relu(bn(conv3x3(x, 3 -> 8)))                                   FeatureNet.conv1 (ConvBnReLU)
//   y  = relu(bn(conv3x3 s2 (f0, 8 -> 16)))                             layer1[0].conv1
//   sc = bn(conv3x3 s2 (f0, 8 -> 16))                                   layer1[0].downsample
//
// f0 is the largest tensor of the network (8 channels at full resolution: 52 MB at cfg 1); as two launches it is written
// once and read once for 0.7 + 1.9 GFLOP of arithmetic (32 + 40 us).  Here a persistent workgroup walks 8 x 32 tiles of the
// half-resolution outputs: it stages the 19 x 67 image patch (3 channels) in LDS (the next tile's patch is fetched into
// registers meanwhile), evaluates f0 on the 17 x 65 positions the stride-2 taps touch (packed vector FMAs, two channels
// each, weights through the scalar cache; positions outside the image are stored as the zeros the next layer's padding
// sees) into LDS, and runs the stride-2 layer as an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32): A = the 32 output
// channels' weights in operand order [tap][k-step][q][co] (registers), B = f0 read from its channel-planar LDS tile at
// stride 2 (plane stride odd: the q = 0 lanes read the even banks, q = 1 the odd ones).  BatchNorm is folded into weights
// and biases by the caller (itermvs_amd/engine.py).
// Measured at cfg 1 (tools/stem_bench.py, phases knocked out one by one): ~11 us launch + prologue, ~12 us f0 (at the vector
// ALU's issue rate), ~21 us stride-2 layer incl. 4 us of stores (12 us of matrix-pipe time); they add up because fp32 MFMA and
// vector instructions do not overlap on this chip: 43-45 us against 72 us for the two launches.
#include "common.hpp"
#include "compose.hpp"
#include <cstdlib>

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(4))) float ConstF;      // constant address space: uniform loads go through the scalar cache

constexpr int kStTW = 32;                                // output tile: TH x 32 (half resolution), TH = 8 or 4
constexpr int kStThreads = 256;
constexpr int kStIP = 68;                                // pitch of the image patch (2 * 32 + 3 columns)
constexpr int kStFP = 66;                                // pitch of the f0 patch (2 * 32 + 1 columns)

struct StemArgs {
    const float* x;
    const float* w0;
    const float* w1;
    float* y;
    float* sc;
    int64_t x_sn, out_sn;
    int M, H, W, H2, W2, tiles_x, tiles_y;
    int out_c4;      // results as [M][4][H2][W2][4] (a lane's four channels of a pixel are one 16-byte store) instead of planes
};

using f32x2 = __attribute__((ext_vector_type(2))) float;

// The image patch of one tile, spread over the workgroup's registers: loaded (addresses clamped) while the previous tile
// is being computed, selected against the image bounds and written to LDS afterwards.  Wave w owns patch rows w, w+4, ...
// (a "row" = one of the 3 * IR channel rows); lane l holds columns l and, for l < 3, 64 + l: every row index and bound is
// wave-uniform, the only per-lane quantities are two column offsets.
template <int kStTH>
struct StemPatch {
    static constexpr int IR = 2 * kStTH + 3, ROWS = 3 * IR, PER = (ROWS + 3) / 4;
    float v[PER], vt[PER];
    unsigned rowok;     // bit i: row i of this wave lies inside the image
    bool c0ok, c1ok;
    __device__ __forceinline__ void fetch(const StemArgs& a, int tile, int wave, int lane) {
        int t = tile;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int iy0 = 2 * ty * kStTH - 2, ix0 = 2 * tx * kStTW - 2;
        const float* __restrict__ xp = a.x + (int64_t)n * a.x_sn;
        const int plane = a.H * a.W;
        const int gx0 = ix0 + lane, gx1 = ix0 + 64 + (lane < 3 ? lane : 2);
        c0ok = gx0 >= 0 && gx0 < a.W;
        c1ok = lane < 3 && gx1 < a.W;
        const int cx0 = min(max(gx0, 0), a.W - 1), cx1 = min(gx1, a.W - 1);
        rowok = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int row = min(wave + 4 * i, ROWS - 1);
            const int ci = row / IR, r = row - ci * IR;
            const int gy = iy0 + r;
            rowok |= (wave + 4 * i < ROWS && gy >= 0 && gy < a.H) ? 1u << i : 0u;
            const float* __restrict__ rp = xp + ci * plane + min(max(gy, 0), a.H - 1) * a.W;
            v[i] = rp[cx0];
            vt[i] = rp[cx1];
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ IMG, int wave, int lane) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int row = wave + 4 * i;
            if (row < ROWS) {
                const bool rok = rowok >> i & 1u;
                IMG[row * kStIP + lane] = rok && c0ok ? v[i] : 0.0f;
                if (lane < 3) IMG[row * kStIP + 64 + lane] = rok && c1ok ? vt[i] : 0.0f;
            }
        }
    }
};

// The first n_comp workgroups (itermvs_stem_compose; otherwise 0) evaluate compose_proj instead of tiles: the fp64 elimination of
// src @ inverse(ref) is a ~10 us dependent chain of a handful of threads that depends on the cameras only -- dispatched first, it
// runs beside the 43 us of this launch instead of being the critical path of a later, shorter one (ref_quarter).
template <int kStTH>
__global__ __launch_bounds__(kStThreads) void stem_kernel(StemArgs a, int tiles, int n_comp, ComposeArgs comp) {
    if ((int)blockIdx.x < n_comp) {
        compose_proj_body(comp, (int)blockIdx.x * kStThreads + (int)threadIdx.x);
        return;
    }
    // Tile order: workgroup b runs on XCD b % 8 (round-robin dispatch); the tiles of one XCD are a contiguous run of the tile list
    // (whole image bands), so that the image rows two vertically adjacent tiles share meet in the same 4 MB L2.  (The n_comp
    // composing workgroups in front shift which workgroups an XCD gets, not the rule.)
    int tile0 = (int)blockIdx.x - n_comp, tstep = (int)gridDim.x - n_comp, tend = tiles;
    if (tstep >= 8) {
        const int b = (int)blockIdx.x, xcd = b & 7;
        const int first = n_comp + ((xcd - n_comp) & 7);             // the XCD's first tile workgroup
        tile0 = (int)((int64_t)tiles * xcd / 8) + ((b - first) >> 3);
        tend = (int)((int64_t)tiles * (xcd + 1) / 8);
        tstep = ((int)gridDim.x - 1 - first) / 8 + 1;
    }

    constexpr int kStIR = 2 * kStTH + 3, kStFR = 2 * kStTH + 1;      // image / f0 patch rows
    constexpr int kStFPL = kStFR * kStFP + 1;                        // odd plane stride (1123 / 595)
    static_assert(kStFPL % 2 == 1, "f0 plane stride must be odd");
    __shared__ float IMG[3 * kStIR * kStIP];
    __shared__ float F0[8 * kStFPL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;

    StemPatch<kStTH> patch;
    if (tile0 < tend) patch.fetch(a, tile0, wave, lane);

    // the stride-2 layer's A operands and biases (registers for the whole kernel)
    float aw[36];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        aw[2 * i] = a.w1[(i * 4 + q) * 32 + l16];
        aw[2 * i + 1] = a.w1[(i * 4 + q) * 32 + 16 + l16];
    }
    const float* __restrict__ bias = a.w1 + 9 * 2 * 4 * 32;
    float bv[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[mb][r] = bias[mb * 16 + q * 4 + r];
    const int oplane = a.H2 * a.W2;

#pragma unroll 1
    for (int tile = tile0; tile < tend; tile += tstep) {
        int t = tile;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int oy0 = ty * kStTH, ox0 = tx * kStTW;
        const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;  // image coordinates of IMG(0,0); f0(r,c) sits at (iy0+1+r, ix0+1+c)

        patch.commit(IMG, wave, lane);
        __syncthreads();
        if (tile + tstep < tend) patch.fetch(a, tile + tstep, wave, lane);    // lands while this tile computes

        // f0 = relu(conv0(x) + b) on the patch: work item = (row, column), a thread evaluates all 8 channels of a position,
        // two channels per packed FMA.  Items FR x 64 (wave w: rows w, w+4, ..; lane = column) and the 65th column
        // (item index FR*64 + row).
        constexpr int ITEMS = kStFR * 64 + kStFR, ROUNDS = (ITEMS + kStThreads - 1) / kStThreads;
#pragma unroll 1
        for (int it = 0; it < ROUNDS; ++it) {
            // 224 loop-invariant weights do not fit the scalar registers: keep their loads inside the loop (the offset is
            // opaque to the optimiser), sixteen weights per s_load_dwordx16 out of the scalar cache
            int opaque = 0;
            asm volatile("" : "+s"(opaque));
            const ConstF* w0 = (const ConstF*)(uintptr_t)a.w0 + __builtin_amdgcn_readfirstlane(opaque);
            const int rr = wave + 4 * it;                       // wave-uniform
            const bool tail = rr >= kStFR;                      // the last wave-rounds take the 65th column
            const int r = tail ? lane : rr, c = tail ? 64 : lane;
            if (tail && (rr > kStFR || lane >= kStFR)) continue;
            float v[27];
            const float* __restrict__ ip = IMG + r * kStIP + c;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) v[(ci * 3 + ky) * 3 + kx] = ip[(ci * kStIR + ky) * kStIP + kx];
            const int gy = iy0 + 1 + r, gx = ix0 + 1 + c;
            const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            f32x2 acc[4];
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) acc[c2] = f32x2{w0[216 + 2 * c2], w0[217 + 2 * c2]};
#pragma unroll
            for (int k = 0; k < 27; ++k)
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2)
                    acc[c2] = __builtin_elementwise_fma(f32x2{w0[k * 8 + 2 * c2], w0[k * 8 + 2 * c2 + 1]}, f32x2{v[k], v[k]}, acc[c2]);
            float* __restrict__ fp = F0 + r * kStFP + c;
#pragma unroll
            for (int co = 0; co < 8; ++co) fp[co * kStFPL] = inside ? fmaxf(acc[co >> 1][co & 1], 0.0f) : 0.0f;
        }
        __syncthreads();

        // stride-2 layer: the TH x 2 groups of 16 positions, round-robin over the four waves
#pragma unroll
        for (int gi = 0; gi < kStTH / 2; ++gi) {      // unrolled: the next group's operands load under this group's MFMAs
            const int g = wave + 4 * gi;
            const int oy = g >> 1, oxl = (g & 1) * 16 + l16;
            f32x4 acc[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[mb] = f32x4{bv[mb][0], bv[mb][1], bv[mb][2], bv[mb][3]};
            const float* __restrict__ bp = F0 + q * kStFPL + 2 * oy * kStFP + 2 * oxl;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float b = bp[ks * 4 * kStFPL + ky * kStFP + kx];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(tap * 2 + ks) * 2], b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(tap * 2 + ks) * 2 + 1], b, acc[1], 0, 0, 0);
                }
            }
            const int gy = oy0 + oy, gx = ox0 + oxl;
            if (gy < a.H2 && gx < a.W2) {
                if (a.out_c4) {
                    const int64_t o = (int64_t)n * a.out_sn + ((int64_t)q * oplane + gy * a.W2 + gx) * 4;
                    *reinterpret_cast<f32x4*>(a.y + o) =
                        f32x4{fmaxf(acc[0][0], 0.0f), fmaxf(acc[0][1], 0.0f), fmaxf(acc[0][2], 0.0f), fmaxf(acc[0][3], 0.0f)};
                    *reinterpret_cast<f32x4*>(a.sc + o) = acc[1];
                } else {
                    const int64_t o = (int64_t)n * a.out_sn + (int64_t)(q * 4) * oplane + gy * a.W2 + gx;
                    float* __restrict__ yo = a.y + o;
                    float* __restrict__ so = a.sc + o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        yo[r * oplane] = fmaxf(acc[0][r], 0.0f);
                        so[r * oplane] = acc[1][r];
                    }
                }
            }
        }
        // the next commit overwrites IMG only (last read before the barrier above); F0 is rewritten after the next barrier
    }
}

}  // namespace itermvs

static int launch_stem(const float* x, int64_t x_sn, int32_t M, int32_t H, int32_t W, const float* w0, const float* w1,
                       float* y, float* sc, int64_t out_sn, int32_t out_layout, const itermvs::ComposeArgs* comp, void* stream) {
    using namespace itermvs;
    if (!x || !w0 || !w1 || !y || !sc) return ITERMVS_ERR_NULL;
    if (M < 0 || H < 1 || W < 1) return ITERMVS_ERR_DIMS;
    if (out_layout != 0 && out_layout != 1) return ITERMVS_ERR_LAYOUT;
    if (out_layout == 1 && (((uintptr_t)y | (uintptr_t)sc) % 16 || out_sn % 4)) return ITERMVS_ERR_ALIGN;
    ComposeArgs c{};
    int n_comp = 0;
    if (comp) {
        c = *comp;
        int ct = c.n_sets * (c.V - 1);
        if (c.inv_min && c.B > ct) ct = c.B;
        n_comp = (ct + kStThreads - 1) / kStThreads;
    }
    if (M == 0 && n_comp == 0) return ITERMVS_OK;
    StemArgs a;
    a.x = x; a.w0 = w0; a.w1 = w1; a.y = y; a.sc = sc;
    a.x_sn = x_sn; a.out_sn = out_sn; a.out_c4 = out_layout;
    a.M = M; a.H = H; a.W = W;
    a.H2 = (H - 1) / 2 + 1; a.W2 = (W - 1) / 2 + 1;
    static const int th = [] { const char* e = itermvs_tuning_env("ITERMVS_STEM_TH"); return e && atoi(e) == 4 ? 4 : 8; }();
    a.tiles_x = (a.W2 + kStTW - 1) / kStTW; a.tiles_y = (a.H2 + th - 1) / th;
    const int64_t tiles = (int64_t)M * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffff) return ITERMVS_ERR_DIMS;
    // persistent workgroups, tiles round-robin: as many as stay resident (LDS 51 / 28 KB per workgroup)
    static const int wg_per_cu = [] { const char* e = itermvs_tuning_env("ITERMVS_STEM_WGS"); return e ? atoi(e) : 0; }();
    const int resident = itermvs_num_cus() * (wg_per_cu > 0 ? wg_per_cu : (th == 4 ? 4 : 3));
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident) + (unsigned)n_comp;
    if (th == 4) hipLaunchKernelGGL(stem_kernel<4>, dim3(grid), dim3(kStThreads), 0, (hipStream_t)stream, a, (int)tiles, n_comp, c);
    else hipLaunchKernelGGL(stem_kernel<8>, dim3(grid), dim3(kStThreads), 0, (hipStream_t)stream, a, (int)tiles, n_comp, c);
    return itermvs_launch_status();
}

extern "C" int itermvs_stem(const float* x, int64_t x_sn, int32_t M, int32_t H, int32_t W, const float* w0, const float* w1,
                            float* y, float* sc, int64_t out_sn, int32_t out_layout, void* stream) {
    return launch_stem(x, x_sn, M, H, W, w0, w1, y, sc, out_sn, out_layout, nullptr, stream);
}

extern "C" int itermvs_stem_compose(const float* x, int64_t x_sn, int32_t M, int32_t H, int32_t W, const float* w0, const float* w1,
                                    float* y, float* sc, int64_t out_sn, int32_t out_layout, const float* mats, int32_t n_sets, int32_t V,
                                    float* proj_out, int32_t* nan_flag, const float* depth_min, const float* depth_max, int32_t Bd,
                                    float* inv_min, float* inv_max, void* stream) {
    ITERMVS_RETURN_IF(!mats || !proj_out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(inv_min && (!depth_min || !depth_max || !inv_max || Bd < 1), ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(n_sets < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(V < 2 || V - 1 > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    const itermvs::ComposeArgs c{mats, proj_out, nan_flag, depth_min, depth_max, inv_min, inv_max, n_sets, V, Bd};
    return launch_stem(x, x_sn, M, H, W, w0, w1, y, sc, out_sn, out_layout, &c, stream);
}
