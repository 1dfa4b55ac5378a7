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
    //         DPP moves (see blend_corr) and lane j finalises groups 2j, 2j+1
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

__device__ __forceinline__ Footprint shfl_footprint(const Footprint& f, int src_lane) {
    Footprint o;
    o.r0 = (uint32_t)__shfl((int)f.r0, src_lane, 64); o.r1 = (uint32_t)__shfl((int)f.r1, src_lane, 64);
    o.c0 = (uint32_t)__shfl((int)f.c0, src_lane, 64); o.c1 = (uint32_t)__shfl((int)f.c1, src_lane, 64);
    o.nw = __shfl(f.nw, src_lane, 64); o.ne = __shfl(f.ne, src_lane, 64);
    o.sw = __shfl(f.sw, src_lane, 64); o.se = __shfl(f.se, src_lane, 64);
    return o;
}

template <int VEC>
struct TapData {
    float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
};

// `fb` is wave-uniform (SGPR base), tap offsets are 32-bit element offsets (saddr + voffset loads).
template <int VEC, int FT>
__device__ __forceinline__ void load_taps(const float* __restrict__ fb, uint32_t joff, const Footprint& tp, TapData<VEC>& t) {
    const uint32_t r0 = tp.r0 + joff, r1 = tp.r1 + joff;
    load_feat<VEC, FT>(fb, r0 + tp.c0, t.v00);
    load_feat<VEC, FT>(fb, r0 + tp.c1, t.v01);
    load_feat<VEC, FT>(fb, r1 + tp.c0, t.v10);
    load_feat<VEC, FT>(fb, r1 + tp.c1, t.v11);
}

// footprint held by quad lane u, in every lane of the quad: 8 DPP moves (u constant after unrolling)
__device__ __forceinline__ Footprint quad_footprint(const Footprint& f, int u) {
    Footprint o;
    o.r0 = quad_bcast(f.r0, u); o.r1 = quad_bcast(f.r1, u); o.c0 = quad_bcast(f.c0, u); o.c1 = quad_bcast(f.c1, u);
    o.nw = quad_bcast(f.nw, u); o.ne = quad_bcast(f.ne, u); o.sw = quad_bcast(f.sw, u); o.se = quad_bcast(f.se, u);
    return o;
}

// group correlation of one lane's chunk for one view: bilinear blend of the four taps, product
// with the reference chunk, mean over the channels of each group (itermvs.py:50-51).
template <int CPG>
__device__ __forceinline__ void blend_corr(const TapData<Chunk<CPG>::VEC>& t, const Footprint& tp,
                                           const float (&refv)[Chunk<CPG>::VEC], float (&corr)[Chunk<CPG>::NG]) {
    constexpr int VEC = Chunk<CPG>::VEC;
    float w[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c)
        w[c] = fmaf(tp.se, t.v11[c], fmaf(tp.sw, t.v10[c], fmaf(tp.ne, t.v01[c], tp.nw * t.v00[c])));
    if constexpr (CPG == 2) {
        corr[0] = fmaf(w[1], refv[1], w[0] * refv[0]) * 0.5f;
        corr[1] = fmaf(w[3], refv[3], w[2] * refv[2]) * 0.5f;
    } else if constexpr (CPG == 4) {   // channels 4j..4j+3 = group j, 16+4j.. = group 4+j
        corr[0] = fmaf(w[3], refv[3], fmaf(w[2], refv[2], fmaf(w[1], refv[1], w[0] * refv[0]))) * 0.25f;
        corr[1] = fmaf(w[7], refv[7], fmaf(w[6], refv[6], fmaf(w[5], refv[5], w[4] * refv[4]))) * 0.25f;
    } else {
        // lane j (= lane & 3) holds channels 16i + 4j + k (i = 0..2, k = 0..3); group g = channels
        // 6g .. 6g+5.  lo_i / hi_i = products of the lower / upper channel pair of block i:
        //   g0 = s0[j0] + lo0[j1]   g1 = hi0[j1] + s0[j2]      (s_i = lo_i + hi_i)
        //   g2 = s0[j3] + lo1[j0]   g3 = hi1[j0] + s1[j1]
        //   g4 = s1[j2] + lo1[j3]   g5 = hi1[j3] + s2[j0]
        //   g6 = s2[j1] + lo2[j2]   g7 = hi2[j2] + s2[j3]
        // lane d finalises groups 2d and 2d+1; each source lane selects what it owes and one
        // quad_perm per term delivers it.
        float lo[3], hi[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lo[i] = fmaf(w[4 * i + 1], refv[4 * i + 1], w[4 * i] * refv[4 * i]);
            hi[i] = fmaf(w[4 * i + 3], refv[4 * i + 3], w[4 * i + 2] * refv[4 * i + 2]);
        }
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

// One view's contribution for one lane: the chunk is walked one 16-channel block at a time (4 tap loads of one float4 each,
// blend, products with the reference block) so that only 16 tap registers are live at once -- with all 12 loads of a C=48
// chunk in flight the kernel needed 114 VGPRs (4 waves per SIMD); block by block it fits 64 (8 waves), and the memory-level
// parallelism comes from the doubled occupancy instead.  Arithmetic and summation order are those of blend_corr.
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

// XCD-aware tile order: block k is observed to run on XCD k % 8 (a speed assumption only), so give
// each XCD a contiguous band of pixel tiles; neighbouring tiles then share one L2 instead of having
// every L2 fetch its own copy of the same source lines.  grid.x is a multiple of 8.
__device__ __forceinline__ int xcd_tile(int tiles) {
    const int per_xcd = (tiles + 7) / 8;
    return (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
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
    const float* nd;
    int64_t nd_sb;
    const float* inv_min;
    const float* inv_max;
    int B, S, H, W, CQ;
};

}  // namespace itermvs

// argument checks shared by the forward and backward entry points
static inline int itermvs_check_level(const itermvs_level_src& s, int S) {
    ITERMVS_RETURN_IF(s.C != 16 && s.C != 32 && s.C != 48, ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(s.H < 1 || s.W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(s.sc != 1, ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(s.dtype < ITERMVS_F32 || s.dtype > ITERMVS_BF16, ITERMVS_ERR_DTYPE);
    ITERMVS_RETURN_IF((s.sx % 4) || (s.sy % 4) || (s.sb % 4), ITERMVS_ERR_ALIGN);
    // 32-bit BYTE offsets of the taps inside one view's map
    ITERMVS_RETURN_IF(s.sx <= 0 || s.sy <= 0 || (int64_t)s.H * s.sy * (s.dtype == ITERMVS_F32 ? 4 : 2) >= (int64_t)1 << 32, ITERMVS_ERR_DIMS);
    for (int v = 0; v < S; ++v) {
        ITERMVS_RETURN_IF(!s.view[v], ITERMVS_ERR_NULL);
        ITERMVS_RETURN_IF(((uintptr_t)s.view[v]) % 16, ITERMVS_ERR_ALIGN);
    }
    return ITERMVS_OK;
}

