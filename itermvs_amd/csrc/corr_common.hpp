// Shared device helpers of the fused warp + correlation kernels (corr.hip forward, corr_bwd.hip gradient).
#pragma once
#include <stdlib.h>

#include "common.hpp"

namespace itermvs {

constexpr int kThreads = 256;

template <int CPG>
struct Chunk {
    // The 4 lanes of one QUAD share a (pixel, hypothesis) [iteration kernel: and walk the views together].  Lane j owns the
    // float4 at channel 4*j of every 16-channel block (C=16: 1 block, C=32: 2, C=48: 3), so every load instruction of the
    // quad covers one contiguous 64-byte run (one L1 access per run; a "6 floats per lane" layout for C=48 cost 9 accesses
    // per tap instead of 3), one tap address serves VEC/4 loads, and everything the quad shares -- footprints, view
    // weights -- travels by DPP quad_perm moves (one VALU instruction each, no LDS round trip like ds_bpermute).
    //   C=16: channels 4j..4j+3           = correlation groups 2j, 2j+1 (2 channels each)
    //   C=32: channels 4j.. and 16+4j..   = groups j and 4+j (4 channels each)
    //   C=48: the lane's 12 channels straddle the 6-channel groups; partial sums are re-grouped with four quad-local
    //         DPP moves (see chunk_corr) and lane j finalises groups 2j, 2j+1
    static constexpr int VEC = 2 * CPG;   // floats per lane: 4 / 8 / 12
    static constexpr int LPT = 4;         // lanes per (pixel, hypothesis)
    static constexpr int NG = 2;          // correlation groups finalised per lane
    static __device__ __forceinline__ int group(int j, int q) { return CPG == 4 ? j + 4 * q : 2 * j + q; }
};

// channel of element c of lane j's chunk
template <int VEC>
__device__ __forceinline__ int chunk_channel(int j, int c) {
    return 16 * (c / 4) + 4 * j + (c % 4);
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC / 4; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(p + 16 * i);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}

// Feature storage types (itermvs_dtype): fp32, or 16-bit storage with fp32 arithmetic.  Pointers to feature maps travel as
// `const float*` through the argument structs; FT says how to read them.  Offsets and strides are always in ELEMENTS.
template <int FT>
__device__ __forceinline__ const float* feat_base(const float* p, int64_t elem_off) {
    if constexpr (FT == ITERMVS_F32) return p + elem_off;
    else return reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(p) + elem_off);
}
template <int FT>
__device__ __forceinline__ float cvt16(uint32_t bits) {   // low 16 bits -> float
    if constexpr (FT == ITERMVS_BF16) return __uint_as_float(bits << 16);
    else return (float)__builtin_bit_cast(_Float16, (uint16_t)bits);
}
template <int FT>
__device__ __forceinline__ float ld_feat(const float* p, int64_t idx) {
    if constexpr (FT == ITERMVS_F32) return p[idx];
    else return cvt16<FT>(reinterpret_cast<const uint16_t*>(p)[idx]);
}
// lane chunk: VEC/4 runs of 4 consecutive channels, 16 channels apart (16 bytes each in fp32, 8 bytes in 16-bit storage)
template <int VEC, int FT>
__device__ __forceinline__ void load_feat(const float* __restrict__ base, uint32_t off, float (&v)[VEC]) {
    if constexpr (FT == ITERMVS_F32) {
        load_vec<VEC>(base + off, v);
    } else {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + off;
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) {
            const uint2 t = *reinterpret_cast<const uint2*>(p + 16 * i);
            if constexpr (FT == ITERMVS_BF16) {
                v[4 * i] = __uint_as_float(t.x << 16); v[4 * i + 1] = __uint_as_float(t.x & 0xffff0000u);
                v[4 * i + 2] = __uint_as_float(t.y << 16); v[4 * i + 3] = __uint_as_float(t.y & 0xffff0000u);
            } else {
                v[4 * i] = cvt16<FT>(t.x); v[4 * i + 1] = cvt16<FT>(t.x >> 16);
                v[4 * i + 2] = cvt16<FT>(t.y); v[4 * i + 3] = cvt16<FT>(t.y >> 16);
            }
        }
    }
}

// four consecutive channels at a 32-bit BYTE offset from a wave-uniform base: the address is one SGPR pair + one VGPR
// (global_load ... v_off, s[base:base+1] offset:imm), no 64-bit vector address arithmetic per load
template <int FT>
__device__ __forceinline__ void load4_at(const char* __restrict__ base, uint32_t byte_off, float (&v)[4]) {
    if constexpr (FT == ITERMVS_F32) {
        const float4 t = *reinterpret_cast<const float4*>(base + byte_off);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const uint2 t = *reinterpret_cast<const uint2*>(base + byte_off);
        if constexpr (FT == ITERMVS_BF16) {
            v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
            v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        } else {
            v[0] = cvt16<FT>(t.x); v[1] = cvt16<FT>(t.x >> 16);
            v[2] = cvt16<FT>(t.y); v[3] = cvt16<FT>(t.y >> 16);
        }
    }
}
template <int FT>
constexpr uint32_t feat_bytes() { return FT == ITERMVS_F32 ? 4u : 2u; }

// v_mov_b32 dpp quad_perm: lane l of each quad reads lane ((CTRL >> 2*l) & 3) of the same quad
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
#define ITERMVS_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

// value of quad lane u in every lane of the quad (one v_mov_b32 dpp)
__device__ __forceinline__ float quad_bcast(float x, int u) {   // u is a constant after unrolling
    switch (u) {
        case 0: return quad_perm<ITERMVS_QP(0, 0, 0, 0)>(x);
        case 1: return quad_perm<ITERMVS_QP(1, 1, 1, 1)>(x);
        case 2: return quad_perm<ITERMVS_QP(2, 2, 2, 2)>(x);
        default: return quad_perm<ITERMVS_QP(3, 3, 3, 3)>(x);
    }
}
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x, int u) {
    return (uint32_t)__float_as_int(quad_bcast(__int_as_float((int)x), u));
}

// Everything about one bilinear footprint that is identical for the chunk lanes of a
// (pixel, hypothesis): 32-bit element offsets of the two rows / two columns and the four weights
// (already zero for out-of-range taps).  Computed by ONE lane and broadcast with shuffles.
struct Footprint {
    uint32_t r0, r1, c0, c1;
    float nw, ne, sw, se;
};

__device__ __forceinline__ Footprint make_footprint(float ix, float iy, int W1, int H1, uint32_t sy, uint32_t sx) {
    const Taps t = make_taps(ix, iy, W1, H1);
    Footprint f;
    f.r0 = (uint32_t)t.y0 * sy; f.r1 = (uint32_t)t.y1 * sy;
    f.c0 = (uint32_t)t.x0 * sx; f.c1 = (uint32_t)t.x1 * sx;
    f.nw = t.nw; f.ne = t.ne; f.sw = t.sw; f.se = t.se;
    return f;
}

// footprint held by quad lane u, in every lane of the quad: 8 DPP moves (u constant after unrolling)
__device__ __forceinline__ Footprint quad_footprint(const Footprint& f, int u) {
    Footprint o;
    o.r0 = quad_bcast(f.r0, u); o.r1 = quad_bcast(f.r1, u); o.c0 = quad_bcast(f.c0, u); o.c1 = quad_bcast(f.c1, u);
    o.nw = quad_bcast(f.nw, u); o.ne = quad_bcast(f.ne, u); o.sw = quad_bcast(f.sw, u); o.se = quad_bcast(f.se, u);
    return o;
}

// One view's contribution for one lane: group correlation of the lane's chunk -- bilinear blend of the four taps, product
// with the reference chunk, mean over the channels of each group (itermvs.py:50-51).  The chunk is walked one 16-channel block
// at a time (4 tap loads of one float4 each, blend, products with the reference block) so that only 16 tap registers are live
// at once (the fp32 iteration kernel allocates 116 VGPRs = 4 waves per SIMD; forcing more waves spills and is slower, fewer
// registers in flight per wave is what the block-wise walk buys -- profiles/r03).
// `fb`: wave-uniform base of the view's map; the footprint's row / column offsets and `joff` are in BYTES (make_footprint
// was given byte strides), so a tap address is a 32-bit add and the 16-channel blocks ride in the instruction's immediate.
template <int CPG, int FT>
__device__ __forceinline__ void chunk_corr(const float* __restrict__ fbase, uint32_t joff, const Footprint& tp,
                                           const float (&refv)[Chunk<CPG>::VEC], float (&corr)[Chunk<CPG>::NG]) {
    constexpr int NBLK = Chunk<CPG>::VEC / 4;
    constexpr uint32_t BLK = 16u * feat_bytes<FT>();      // bytes between the lane's 16-channel blocks
    const char* __restrict__ fb = reinterpret_cast<const char*>(fbase);
    const uint32_t o00 = tp.r0 + tp.c0 + joff, o01 = tp.r0 + tp.c1 + joff, o10 = tp.r1 + tp.c0 + joff, o11 = tp.r1 + tp.c1 + joff;
    float lo[NBLK], hi[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        float v00[4], v01[4], v10[4], v11[4], w[4];
        load4_at<FT>(fb, o00 + BLK * i, v00);
        load4_at<FT>(fb, o01 + BLK * i, v01);
        load4_at<FT>(fb, o10 + BLK * i, v10);
        load4_at<FT>(fb, o11 + BLK * i, v11);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            w[c] = fmaf(tp.se, v11[c], fmaf(tp.sw, v10[c], fmaf(tp.ne, v01[c], tp.nw * v00[c])));
        const float* r = refv + 4 * i;
        if constexpr (CPG == 4) {   // one 4-channel group per block: a single fma chain (itermvs.py:103-104 order)
            lo[i] = fmaf(w[3], r[3], fmaf(w[2], r[2], fmaf(w[1], r[1], w[0] * r[0])));
            hi[i] = 0.0f;
        } else {
            lo[i] = fmaf(w[1], r[1], w[0] * r[0]);
            hi[i] = fmaf(w[3], r[3], w[2] * r[2]);
        }
        if (i + 1 < NBLK) __builtin_amdgcn_sched_barrier(0);     // the next block's loads stay behind this block's blend
    }
    if constexpr (CPG == 2) {
        corr[0] = lo[0] * 0.5f;
        corr[1] = hi[0] * 0.5f;
    } else if constexpr (CPG == 4) {   // channels 4j..4j+3 = group j, 16+4j.. = group 4+j
        corr[0] = lo[0] * 0.25f;
        corr[1] = lo[1] * 0.25f;
    } else {
        // lane j (= lane & 3) holds channels 16i + 4j + k (i = 0..2, k = 0..3); group g = channels 6g .. 6g+5.
        // lo_i / hi_i = products of the lower / upper channel pair of block i:
        //   g0 = s0[j0] + lo0[j1]   g1 = hi0[j1] + s0[j2]      (s_i = lo_i + hi_i)
        //   g2 = s0[j3] + lo1[j0]   g3 = hi1[j0] + s1[j1]
        //   g4 = s1[j2] + lo1[j3]   g5 = hi1[j3] + s2[j0]
        //   g6 = s2[j1] + lo2[j2]   g7 = hi2[j2] + s2[j3]
        // lane d finalises groups 2d and 2d+1; each source lane selects what it owes and one quad_perm per term delivers it.
        const float s0 = lo[0] + hi[0], s1 = lo[1] + hi[1], s2 = lo[2] + hi[2];
        const int j = threadIdx.x & 3;
        const float ta = (j == 0 || j == 3) ? s0 : (j == 2 ? s1 : s2);
        const float tb = (j == 1) ? lo[0] : (j == 2 ? lo[2] : lo[1]);
        const float tc = (j == 1) ? hi[0] : (j == 2 ? hi[2] : hi[1]);
        const float tdd = (j == 2) ? s0 : (j == 1 ? s1 : s2);
        const float g_first = quad_perm<ITERMVS_QP(0, 3, 2, 1)>(ta) + quad_perm<ITERMVS_QP(1, 0, 3, 2)>(tb);
        const float g_second = quad_perm<ITERMVS_QP(1, 0, 3, 2)>(tc) + quad_perm<ITERMVS_QP(2, 1, 0, 3)>(tdd);
        // mean over the 6 channels of the group (itermvs.py:103-104): three instructions instead of the ~10 of an IEEE division
        corr[0] = div_rcp(g_first, 6.0f, 1.0f / 6.0f);
        corr[1] = div_rcp(g_second, 6.0f, 1.0f / 6.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-bit feature storage (fp16 / bf16), 16-BYTE lanes.  A wave-level vector load occupies the CU's texture-address path for
// one quad of lanes per cycle whatever the lanes' width (profiles/r03: with 8-byte lanes the fp16 kernels issued exactly the
// fp32 kernels' load instructions and ran no faster), so with 2-byte elements every lane fetches 16 bytes = 8 channels and a
// pixel's taps need HALF the load instructions of the fp32 layout:
//   C = 32 (64 B per pixel): one quad load per tap; lane j owns channels 8j .. 8j+7 = correlation groups 2j, 2j+1
//   C = 16 (32 B per pixel): one quad load per ROW of the footprint: lanes 0,1 fetch the x0 tap, lanes 2,3 the x1 tap (a
//                            contiguous 64-byte run when x1 = x0 + 1), then the halves are exchanged inside the quad
//                            (exchange16) so that lane j holds 4 channels of BOTH taps: channels cb(j) .. cb(j)+3
//   C = 48 (96 B per pixel): per row two full loads (channels 0..31 of x0 and of x1) + one half-quad load of channels
//                            32..47 of both taps: six loads per footprint instead of twelve; lane j owns channels
//                            8j .. 8j+7 and 32 + cb(j) .. 32 + cb(j) + 3
// cb(j) = 8 * (j & 1) + 4 * (j >> 1)   (lane 0: 0, lane 1: 8, lane 2: 4, lane 3: 12).
// The arithmetic per channel (fma chain over the four taps), per channel pair and per group is EXACTLY that of the fp32
// layout above (same operations, same association), only the lane that holds a channel differs: results are bit-identical
// to the fp32 kernels run on the same values (tests/test_kernels_gpu.py::test_16bit_feature_storage_matches_...).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int chunk16_cb(int j) { return 8 * (j & 1) + 4 * (j >> 1); }

// correlation groups lane j finalises (slot q = 0, 1) with 16-byte lanes
template <int CPG>
__device__ __forceinline__ int group16(int j, int q) {
    if constexpr (CPG == 2) return chunk16_cb(j) / 2 + q;       // lane 0: 0,1   lane 1: 4,5   lane 2: 2,3   lane 3: 6,7
    else if constexpr (CPG == 4) return 2 * j + q;
    else return q == 0 ? (j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 3 : 4) : (j == 0 ? 1 : j == 1 ? 7 : j == 2 ? 6 : 5);   // see regroup48
}

// floats of the reference chunk a lane multiplies with (element offsets relative to the level's first channel):
// VEC floats in the order the lane's channels are listed above
template <int CPG>
__device__ __forceinline__ void load_ref16(const float* __restrict__ r, int j, float (&refv)[2 * CPG]) {
    if constexpr (CPG == 2) {
        load_vec<4>(r + chunk16_cb(j), refv);
    } else {
        const float4 a = *reinterpret_cast<const float4*>(r + 8 * j), b = *reinterpret_cast<const float4*>(r + 8 * j + 4);
        refv[0] = a.x; refv[1] = a.y; refv[2] = a.z; refv[3] = a.w; refv[4] = b.x; refv[5] = b.y; refv[6] = b.z; refv[7] = b.w;
        if constexpr (CPG == 6) {
            const float4 c = *reinterpret_cast<const float4*>(r + 32 + chunk16_cb(j));
            refv[8] = c.x; refv[9] = c.y; refv[10] = c.z; refv[11] = c.w;
        }
    }
}
// channel of element c of lane j's 16-byte-lane chunk (load_ref16 order)
template <int CPG>
__device__ __forceinline__ int chunk16_channel(int j, int c) {
    if constexpr (CPG == 2) return chunk16_cb(j) + c;
    else return c < 8 ? 8 * j + c : 32 + chunk16_cb(j) + (c - 8);
}
// the same chunk from a 16-bit channels-last map (initialisation branch: the reference features are stored like the sources)
template <int FT>
__device__ __forceinline__ void cvt_pair(uint32_t d, float& lo, float& hi) {   // two packed 16-bit values -> fp32
    if constexpr (FT == ITERMVS_BF16) { lo = __uint_as_float(d << 16); hi = __uint_as_float(d & 0xffff0000u); }
    else { lo = cvt16<FT>(d); hi = cvt16<FT>(d >> 16); }
}
template <int CPG, int FT>
__device__ __forceinline__ void load_ref16_stored(const char* __restrict__ base, uint32_t byte_off, int j, float (&refv)[2 * CPG]) {
    if constexpr (CPG == 2) {
        const uint2 t = *reinterpret_cast<const uint2*>(base + byte_off + 2u * (uint32_t)chunk16_cb(j));
        cvt_pair<FT>(t.x, refv[0], refv[1]); cvt_pair<FT>(t.y, refv[2], refv[3]);
    } else {
        const uint4 t = *reinterpret_cast<const uint4*>(base + byte_off + 16u * (uint32_t)j);
        cvt_pair<FT>(t.x, refv[0], refv[1]); cvt_pair<FT>(t.y, refv[2], refv[3]);
        cvt_pair<FT>(t.z, refv[4], refv[5]); cvt_pair<FT>(t.w, refv[6], refv[7]);
        if constexpr (CPG == 6) {
            const uint2 u = *reinterpret_cast<const uint2*>(base + byte_off + 64u + 2u * (uint32_t)chunk16_cb(j));
            cvt_pair<FT>(u.x, refv[8], refv[9]); cvt_pair<FT>(u.y, refv[10], refv[11]);
        }
    }
}

__device__ __forceinline__ uint4 load16_at(const char* __restrict__ base, uint32_t byte_off) {
    return *reinterpret_cast<const uint4*>(base + byte_off);
}
__device__ __forceinline__ uint32_t quad_swap2(uint32_t x) {       // lane l <- lane l ^ 2 of its quad
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, ITERMVS_QP(2, 3, 0, 1), 0xf, 0xf, true);
}
// Half-quad load -> both taps of the lane's four channels.  `d` = the 16 bytes the lane loaded: lanes 0,1 hold channels
// 0..7 / 8..15 of the LEFT tap (x0), lanes 2,3 the same channels of the RIGHT tap (x1).  Afterwards every lane holds, for
// its channels cb(j) .. cb(j)+3, the left tap in (l0, l1) and the right tap in (r0, r1) (two packed 16-bit values per dword):
//   low lanes  (0,1): left = own d.x, d.y              right = partner's d.x, d.y
//   high lanes (2,3): left = partner's d.z, d.w        right = own d.z, d.w
__device__ __forceinline__ void exchange16(const uint4& d, bool high, uint32_t& l0, uint32_t& l1, uint32_t& r0, uint32_t& r1) {
    const uint32_t sx = quad_swap2(d.x), sy = quad_swap2(d.y), sz = quad_swap2(d.z), sw = quad_swap2(d.w);
    l0 = high ? sz : d.x;
    l1 = high ? sw : d.y;
    r0 = high ? d.z : sx;
    r1 = high ? d.w : sy;
}

// bilinear blend of four channels (two packed dwords per tap): the per-channel fma chain of chunk_corr
template <int FT>
__device__ __forceinline__ void blend4_pairs(const Footprint& tp, uint32_t a00, uint32_t b00, uint32_t a01, uint32_t b01,
                                             uint32_t a10, uint32_t b10, uint32_t a11, uint32_t b11, float (&w)[4]) {
    float v00[4], v01[4], v10[4], v11[4];
    cvt_pair<FT>(a00, v00[0], v00[1]); cvt_pair<FT>(b00, v00[2], v00[3]);
    cvt_pair<FT>(a01, v01[0], v01[1]); cvt_pair<FT>(b01, v01[2], v01[3]);
    cvt_pair<FT>(a10, v10[0], v10[1]); cvt_pair<FT>(b10, v10[2], v10[3]);
    cvt_pair<FT>(a11, v11[0], v11[1]); cvt_pair<FT>(b11, v11[2], v11[3]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        w[c] = fmaf(tp.se, v11[c], fmaf(tp.sw, v10[c], fmaf(tp.ne, v01[c], tp.nw * v00[c])));
}

// One view's group correlations for one lane with 16-byte lanes (see the table above).  `fb`: wave-uniform base of the view's
// map; the footprint's offsets are BYTES.  `refv`: the lane's reference chunk in load_ref16 order.  corr[q] belongs to group
// group16<CPG>(j, q).
template <int CPG, int FT>
__device__ __forceinline__ void chunk_corr16(const float* __restrict__ fbase, int j, const Footprint& tp,
                                             const float (&refv)[2 * CPG], float (&corr)[2]) {
    const char* __restrict__ fb = reinterpret_cast<const char*>(fbase);
    const bool high = (j & 2) != 0;
    if constexpr (CPG == 4) {
        // a full 64-byte pixel per quad load: channels 8j .. 8j+7
        const uint32_t jo = 16u * (uint32_t)j;
        const uint4 t00 = load16_at(fb, tp.r0 + tp.c0 + jo), t01 = load16_at(fb, tp.r0 + tp.c1 + jo);
        const uint4 t10 = load16_at(fb, tp.r1 + tp.c0 + jo), t11 = load16_at(fb, tp.r1 + tp.c1 + jo);
        float wa[4], wb[4];
        blend4_pairs<FT>(tp, t00.x, t00.y, t01.x, t01.y, t10.x, t10.y, t11.x, t11.y, wa);
        blend4_pairs<FT>(tp, t00.z, t00.w, t01.z, t01.w, t10.z, t10.w, t11.z, t11.w, wb);
        // one 4-channel group per half: a single fma chain (itermvs.py:103-104 order), like chunk_corr<4>
        corr[0] = fmaf(wa[3], refv[3], fmaf(wa[2], refv[2], fmaf(wa[1], refv[1], wa[0] * refv[0]))) * 0.25f;
        corr[1] = fmaf(wb[3], refv[7], fmaf(wb[2], refv[6], fmaf(wb[1], refv[5], wb[0] * refv[4]))) * 0.25f;
    } else {
        // half-quad addressing: lanes 0,1 -> column x0, lanes 2,3 -> column x1; 16 bytes at piece (j & 1)
        const uint32_t col = (high ? tp.c1 : tp.c0) + 16u * (uint32_t)(j & 1);
        if constexpr (CPG == 2) {
            const uint4 d0 = load16_at(fb, tp.r0 + col), d1 = load16_at(fb, tp.r1 + col);
            uint32_t a00, b00, a01, b01, a10, b10, a11, b11;
            exchange16(d0, high, a00, b00, a01, b01);
            exchange16(d1, high, a10, b10, a11, b11);
            float w[4];
            blend4_pairs<FT>(tp, a00, b00, a01, b01, a10, b10, a11, b11, w);
            corr[0] = fmaf(w[1], refv[1], w[0] * refv[0]) * 0.5f;
            corr[1] = fmaf(w[3], refv[3], w[2] * refv[2]) * 0.5f;
        } else {
            const uint32_t jo = 16u * (uint32_t)j;
            float p[4], e[2];
            {   // channels 8j .. 8j+7 of the four taps: pairs p0..p3
                const uint4 t00 = load16_at(fb, tp.r0 + tp.c0 + jo), t01 = load16_at(fb, tp.r0 + tp.c1 + jo);
                const uint4 t10 = load16_at(fb, tp.r1 + tp.c0 + jo), t11 = load16_at(fb, tp.r1 + tp.c1 + jo);
                float wa[4], wb[4];
                blend4_pairs<FT>(tp, t00.x, t00.y, t01.x, t01.y, t10.x, t10.y, t11.x, t11.y, wa);
                blend4_pairs<FT>(tp, t00.z, t00.w, t01.z, t01.w, t10.z, t10.w, t11.z, t11.w, wb);
                p[0] = fmaf(wa[1], refv[1], wa[0] * refv[0]);
                p[1] = fmaf(wa[3], refv[3], wa[2] * refv[2]);
                p[2] = fmaf(wb[1], refv[5], wb[0] * refv[4]);
                p[3] = fmaf(wb[3], refv[7], wb[2] * refv[6]);
            }
            __builtin_amdgcn_sched_barrier(0);      // the tail's loads stay behind the main block's blend (register pressure)
            {   // channels 32 + cb(j) .. +3 of the four taps: pairs e0, e1
                const uint4 d0 = load16_at(fb, tp.r0 + col + 64u), d1 = load16_at(fb, tp.r1 + col + 64u);
                uint32_t a00, b00, a01, b01, a10, b10, a11, b11;
                exchange16(d0, high, a00, b00, a01, b01);
                exchange16(d1, high, a10, b10, a11, b11);
                float w[4];
                blend4_pairs<FT>(tp, a00, b00, a01, b01, a10, b10, a11, b11, w);
                e[0] = fmaf(w[1], refv[9], w[0] * refv[8]);
                e[1] = fmaf(w[3], refv[11], w[2] * refv[10]);
            }
            // regroup48.  P_k = channel pair (2k, 2k+1); lane j holds p_i = P_{4j+i} and (e0, e1) = P16,17 / P20,21 / P18,19 /
            // P22,23 for j = 0 / 1 / 2 / 3.  Group g = P_{3g} , P_{3g+1}, P_{3g+2} with the association of chunk_corr<6>:
            // even groups (A + B) + C, odd groups A + (B + C):
            //   g0 = (P0+P1)+P2      lane 0: (p0+p1) + p2             g1 = P3+(P4+P5)      lane 0: p3 + [lane 1: p0+p1]
            //   g2 = (P6+P7)+P8      lane 1: (p2+p3) + [lane 2: p0]   g3 = P9+(P10+P11)    lane 2: p1 + (p2+p3)
            //   g4 = (P12+P13)+P14   lane 3: (p0+p1) + p2             g5 = P15+(P16+P17)   lane 3: p3 + [lane 0: e0+e1]
            //   g6 = (P18+P19)+P20   lane 2: (e0+e1) + [lane 1: e0]   g7 = P21+(P22+P23)   lane 1: e1 + [lane 3: e0+e1]
            const float s01 = p[0] + p[1], s23 = p[2] + p[3], se = e[0] + e[1];
            const float send1 = j == 0 ? se : (j == 1 ? s01 : p[0]);
            const float send2 = j == 1 ? e[0] : se;
            const float r1 = quad_perm<ITERMVS_QP(1, 2, 1, 0)>(send1);      // lane 0 <- 1, lane 1 <- 2, lane 3 <- 0
            const float r2 = quad_perm<ITERMVS_QP(0, 3, 1, 0)>(send2);      // lane 1 <- 3, lane 2 <- 1
            const float xa = j == 2 ? p[1] : (j == 1 ? s23 : s01);
            const float ya = j == 1 ? r1 : (j == 2 ? s23 : p[2]);
            const float xb = (j == 0 || j == 3) ? p[3] : (j == 1 ? e[1] : se);
            const float yb = (j == 0 || j == 3) ? r1 : r2;
            corr[0] = div_rcp(xa + ya, 6.0f, 1.0f / 6.0f);
            corr[1] = div_rcp(xb + yb, 6.0f, 1.0f / 6.0f);
        }
    }
}

// (A PAIR form -- two lanes per (pixel, hypothesis), each owning a contiguous half of the channels: no exchange, no
// regrouping, half the per-lane broadcast / address overhead, the same number of load instructions -- was measured and
// removed: a quad of lanes then reads two separate 16-byte pieces per load, which the texture-address path serves at
// half the rate.  fp16: 22.1 / 20.6 us vs 19.8 / 18.8 us (noise / smooth depth map); fp32: 35.7 / 33.7 vs 24.2 / 21.7 us;
// profiles/r04/r04c_corr_lane_forms_ab.txt.  The lanes of a quad must cover ONE contiguous run.)

// XCD-aware tile order: block k is observed to run on XCD k % 8 (a speed assumption only), so give
// each XCD a contiguous band of pixel tiles; neighbouring tiles then share one L2 instead of having
// every L2 fetch its own copy of the same source lines.  grid.x is a multiple of 8.
__device__ __forceinline__ int xcd_tile(int tiles) {
    const int per_xcd = (tiles + 7) / 8;
    return (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
}

// The same with NARROW bands dealt round-robin: band k (`chunk` consecutive tiles in row-major order) belongs to XCD k % 8, the
// j-th workgroup of an XCD takes tile j % chunk of its (j / chunk)-th band.  All eight XCDs then sweep the SAME part of the maps
// at a time (each with its own bands of it): on maps whose ten views' footprints of one XCD's band no longer fit its 4 MB L2 (the
// cfg-5 shape: L2 hit rate 0.49, HBM traffic 2.3x the algorithmic bytes with eight far-apart bands live at once) the lines one
// XCD drops are still in the memory-side cache for the next.  The launcher sizes the grid to 8 * ceil(bands / 8) * chunk.
__device__ __forceinline__ int xcd_tile_chunked(int chunk) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int band = j / chunk;
    return (band * 8 + xcd) * chunk + (j - band * chunk);
}
static inline unsigned xcd_chunked_grid(int tiles, int chunk) {
    const int bands = (tiles + chunk - 1) / chunk;
    return (unsigned)(8 * ((bands + 7) / 8) * chunk);
}

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
    int64_t vw_sb, vw_ss, vw_sp;     // element strides of view_w: batch, view, pixel (p = y * W + x)
    const float* nd;
    int64_t nd_sb;
    const float* inv_min;
    const float* inv_max;
    int B, S, H, W, CQ;
    int band;                        // tiles per XCD band (xcd_tile_chunked); 0 = one contiguous band per XCD (xcd_tile)
};

}  // namespace itermvs

// argument checks shared by the forward and backward entry points
static inline int itermvs_check_level(const itermvs_level_src& s, int S) {
    ITERMVS_RETURN_IF(s.C != 16 && s.C != 32 && s.C != 48, ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(s.H < 1 || s.W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(s.sc != 1, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(s.dtype < ITERMVS_F32 || s.dtype > ITERMVS_BF16, ITERMVS_ERR_DTYPE);
    ITERMVS_RETURN_IF((s.sx % 4) || (s.sy % 4) || (s.sb % 4), ITERMVS_ERR_ALIGN);
    // 16-bit storage is read with 16-byte lanes: every pixel vector starts on a 16-byte boundary
    ITERMVS_RETURN_IF(s.dtype != ITERMVS_F32 && ((s.sx % 8) || (s.sy % 8) || (s.sb % 8)), ITERMVS_ERR_ALIGN);
    // 32-bit BYTE offsets of the taps inside one view's map
    ITERMVS_RETURN_IF(s.sx <= 0 || s.sy <= 0 || (int64_t)s.H * s.sy * (s.dtype == ITERMVS_F32 ? 4 : 2) >= (int64_t)1 << 32, ITERMVS_ERR_DIMS);
    for (int v = 0; v < S; ++v) {
        ITERMVS_RETURN_IF(!s.view[v], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(((uintptr_t)s.view[v]) % 16, ITERMVS_ERR_ALIGN);
    }
    return ITERMVS_OK;
}

