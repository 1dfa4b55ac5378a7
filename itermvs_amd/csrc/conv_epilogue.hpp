// Shared epilogue of the matrix-core convolution kernels (conv_mfma.hip, conv_tile.hip).
//
// A wave holds MB x NB accumulator tiles of v_mfma_f32_16x16x4_f32: lane l owns output channels
// m0 + mb*16 + (l >> 4)*4 + r (r = 0..3) of pixel slot nb.  Bias / residual / activation / ConvGRU gate
// math are applied on the accumulators and written to the NCHW planes (16 lanes = one 64-byte run).
//
// gfx9 retires loads AND stores in order through one counter (vmcnt): an epilogue that loads an optional
// operand per element forces `s_waitcnt vmcnt(0)` between consecutive stores -- one L2 round trip per
// element (measured: 590 cycles per store, 9.4k of the 20k cycles a workgroup spent per tile).  Here every
// access goes through a buffer descriptor (invalid pixels / padded channels get an out-of-range offset and
// are dropped by the hardware bounds check: no divergent branches), optional operands are fetched for a
// whole pixel slot before its first store, and a slot without optional operands issues its stores back to
// back with no wait at all.
#pragma once

#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr uint32_t kEpiOob = 0x7fffffffu;

template <int ACT>
__device__ __forceinline__ float conv_activation(float v, float a1, float a2) {
    if constexpr (ACT == 1) return fmaxf(v, 0.0f);
    else if constexpr (ACT == 2) return sigmoidf_(v);
    else if constexpr (ACT == 3) return tanhf(v);
    else if constexpr (ACT == 4) return sigmoidf_(v) * a1;                  // r * h            (module.py:63-64)
    else if constexpr (ACT == 5) return (1.0f - a2) * a1 + a2 * tanhf(v);   // (1-z) h + z q    (module.py:64-65)
    else return v;
}

struct EpilogueArgs {
    float* out;          // + n * out_sn applied by the caller
    float* out2;         // dense [Cout][P] copy or nullptr
    const float* add;    // optional, same indexing as out (already offset to batch item n)
    const float* aux1;
    const float* aux2;
    int Cout, P, act;
    int out_nhwc;               // != 0: `out` is channels-last [P][Cout] (per batch item) instead of [Cout][P];
                                // 1 fp32, 2 fp16, 3 bf16 storage (itermvs_conv_params.out_layout)
    int add_mode, Hout, Wout;   // add_mode 1: `add` is the half-resolution tensor, up-sampled x2 (bilinear) here
};

// `out` advanced by `elems` ELEMENTS of the output's storage type
__device__ __forceinline__ float* epi_out_base(float* out, int64_t elems, int out_nhwc) {
    return out_nhwc >= 2 ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(out) + elems) : out + elems;
}
// fp32 -> 16-bit storage, round to nearest even (what torch's .half() / .bfloat16() do)
__device__ __forceinline__ uint32_t epi_to_f16(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v); }
__device__ __forceinline__ uint32_t epi_to_bf16(float v) {
    const uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float epi_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff = 0) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void epi_store(float v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}

// The bias is NOT added here: the kernels start their accumulators from it (conv_bias_init), which saves
// one VALU instruction per output -- on gfx950 fp32 MFMAs and ordinary VALU instructions of other waves do
// not overlap (tools/ubench/mfma_valu_overlap.hip: the times add), so every VALU instruction in these
// kernels is paid for in matrix-core time.
template <int MB, int NB>
__device__ __forceinline__ void conv_bias_init(f32x4 (&acc)[MB][NB], const float* bias, int Cout, int m0, int q) {
    float bs[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[mb][r] = 0.0f;
    if (bias) {
        const __amdgpu_buffer_rsrc_t rb = epi_rsrc(bias, (uint32_t)Cout * 4u);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) bs[mb][r] = epi_load(rb, (uint32_t)(m0 + mb * 16 + q * 4 + r) * 4u);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{bs[mb][0], bs[mb][1], bs[mb][2], bs[mb][3]};
}

// pix_off[nb]: byte offset of this lane's pixel of slot nb inside a channel plane, or kEpiOob.
// Element (mb, r) of a lane is channel m0 + mb*16 + q*4 + r: the lane part (q*4 planes + pixel) is ONE
// vector offset per slot, the (m0 + mb*16 + r) planes go into the instruction's scalar offset -- no vector
// address arithmetic per element, and padded channels (>= Cout) land beyond the descriptor by themselves.
// The activation is a template parameter (one uniform switch in conv_epilogue below): inside the element
// loops the code is straight-line.
template <int ACT, int ADD, int MB, int NB>
__device__ __forceinline__ void conv_epilogue_act(const EpilogueArgs& e, const f32x4 (&acc)[MB][NB], int m0, int q,
                                                  const uint32_t (&pix_off)[NB], const int (&py)[NB], const int (&px)[NB]) {
    const uint32_t plane_b = (uint32_t)e.P * 4u;
    const uint32_t bytes = (uint32_t)e.Cout * plane_b;
    const __amdgpu_buffer_rsrc_t ro = epi_rsrc(e.out, bytes);
    const uint32_t lane_ch = (uint32_t)(q * 4) * plane_b;
    const uint32_t s0 = (uint32_t)m0 * plane_b;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t voff = pix_off[nb] == kEpiOob ? kEpiOob : pix_off[nb] + lane_ch;
        float ad[MB][4], a1[MB][4], a2[MB][4], v[MB][4];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) ad[mb][r] = a1[mb][r] = a2[mb][r] = 0.0f;
        if constexpr (ADD == 1) {
            const __amdgpu_buffer_rsrc_t rr = epi_rsrc(e.add, bytes);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ad[mb][r] = epi_load(rr, voff, s0 + (uint32_t)(mb * 16 + r) * plane_b);
        }
        if constexpr (ADD == 2) {
            // F.interpolate(scale_factor=2, 'bilinear', align_corners=False) of the half-resolution tensor:
            // same arithmetic as bilinear_up_kernel (update.hip), models/net.py:46,49
            const int Hc = e.Hout >> 1, Wc = e.Wout >> 1;
            const uint32_t cplane_b = (uint32_t)(Hc * Wc) * 4u;
            const __amdgpu_buffer_rsrc_t rr = epi_rsrc(e.add, (uint32_t)e.Cout * cplane_b);
            float sy = ((float)py[nb] + 0.5f) * 0.5f - 0.5f, sx = ((float)px[nb] + 0.5f) * 0.5f - 0.5f;
            sy = sy < 0.0f ? 0.0f : sy;
            sx = sx < 0.0f ? 0.0f : sx;
            int y0 = (int)sy, x0 = (int)sx;
            y0 = y0 > Hc - 1 ? Hc - 1 : y0;
            x0 = x0 > Wc - 1 ? Wc - 1 : x0;
            const int y1 = y0 + (y0 < Hc - 1 ? 1 : 0), x1 = x0 + (x0 < Wc - 1 ? 1 : 0);
            const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
            const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
            const uint32_t lane_c = (uint32_t)(q * 4) * cplane_b;
            const uint32_t o00 = (uint32_t)(y0 * Wc + x0) * 4u + lane_c, o01 = (uint32_t)(y0 * Wc + x1) * 4u + lane_c;
            const uint32_t o10 = (uint32_t)(y1 * Wc + x0) * 4u + lane_c, o11 = (uint32_t)(y1 * Wc + x1) * 4u + lane_c;
            const uint32_t sc0 = (uint32_t)m0 * cplane_b;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t so = sc0 + (uint32_t)(mb * 16 + r) * cplane_b;
                    const float v00 = epi_load(rr, o00, so), v01 = epi_load(rr, o01, so);
                    const float v10 = epi_load(rr, o10, so), v11 = epi_load(rr, o11, so);
                    const float top = v00 * lx0 + v01 * lx1;
                    const float bot = v10 * lx0 + v11 * lx1;
                    ad[mb][r] = top * ly0 + bot * ly1;
                }
        }
        if constexpr (ACT == 4 || ACT == 5) {
            const __amdgpu_buffer_rsrc_t rr = epi_rsrc(e.aux1, bytes);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) a1[mb][r] = epi_load(rr, voff, s0 + (uint32_t)(mb * 16 + r) * plane_b);
        }
        if constexpr (ACT == 5) {
            const __amdgpu_buffer_rsrc_t rr = epi_rsrc(e.aux2, bytes);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[mb][r] = epi_load(rr, voff, s0 + (uint32_t)(mb * 16 + r) * plane_b);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = ADD != 0 ? acc[mb][nb][r] + ad[mb][r] : acc[mb][nb][r];
                v[mb][r] = conv_activation<ACT>(x, a1[mb][r], a2[mb][r]);
                epi_store(v[mb][r], ro, voff, s0 + (uint32_t)(mb * 16 + r) * plane_b);
            }
        if (e.out2) {
            const __amdgpu_buffer_rsrc_t r2 = epi_rsrc(e.out2, bytes);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) epi_store(v[mb][r], r2, voff, s0 + (uint32_t)(mb * 16 + r) * plane_b);
        }
    }
}

// Channels-last output (the feature maps the correlation kernels gather from): a lane's four channels of one
// pixel are 16 contiguous bytes -> one dwordx4 store per (mb, nb) instead of four dword stores into four
// planes.  act 0, no residual (checked on the host); `out2` may still take the planar copy.
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue_nhwc(const EpilogueArgs& e, const f32x4 (&acc)[MB][NB], int m0, int q,
                                                   const uint32_t (&pix_off)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (pix_off[nb] == kEpiOob) continue;
        if (e.out_nhwc == 1) {
            float* __restrict__ row = e.out + (size_t)(pix_off[nb] >> 2) * (size_t)e.Cout;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int co = m0 + mb * 16 + q * 4;
                if (co < e.Cout) *reinterpret_cast<f32x4*>(row + co) = acc[mb][nb];
            }
        } else {   // 16-bit storage: the lane's four channels are 8 contiguous bytes
            uint16_t* __restrict__ row = reinterpret_cast<uint16_t*>(e.out) + (size_t)(pix_off[nb] >> 2) * (size_t)e.Cout;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int co = m0 + mb * 16 + q * 4;
                const f32x4 v = acc[mb][nb];
                uint2 pk;
                if (e.out_nhwc == 2) {
                    pk.x = epi_to_f16(v[0]) | (epi_to_f16(v[1]) << 16);
                    pk.y = epi_to_f16(v[2]) | (epi_to_f16(v[3]) << 16);
                } else {
                    pk.x = epi_to_bf16(v[0]) | (epi_to_bf16(v[1]) << 16);
                    pk.y = epi_to_bf16(v[2]) | (epi_to_bf16(v[3]) << 16);
                }
                if (co < e.Cout) *reinterpret_cast<uint2*>(row + co) = pk;
            }
        }
    }
    if (e.out2) {
        const uint32_t plane_b = (uint32_t)e.P * 4u;
        const __amdgpu_buffer_rsrc_t r2 = epi_rsrc(e.out2, (uint32_t)e.Cout * plane_b);
        const uint32_t lane_ch = (uint32_t)(q * 4) * plane_b, s0 = (uint32_t)m0 * plane_b;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const uint32_t voff = pix_off[nb] == kEpiOob ? kEpiOob : pix_off[nb] + lane_ch;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) epi_store(acc[mb][nb][r], r2, voff, s0 + (uint32_t)(mb * 16 + r) * plane_b);
        }
    }
}

// act 6 -- a 1x1 convolution to ONE channel folded into the epilogue of a 16-channel layer (PixelViewWeight, itermvs.py:
// 337-346: conv3x3 8 -> 16, ReLU, conv1x1 16 -> 1): out[p] = sum_co relu(acc[co][p]) * w[co] + w[16].  A lane holds channels
// q*4 .. q*4+3 of its pixel: four FMAs, then the four q-lanes of the pixel are added with two xor-shuffles and lane q = 0
// stores ONE value -- the 16-channel tensor (42 MB at cfg 1) is never written.  `aux1` = the 17 floats {w[0..15], bias}.
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue_dot(const EpilogueArgs& e, const f32x4 (&acc)[MB][NB], int q,
                                                  const uint32_t (&pix_off)[NB]) {
    // the wave holds ALL MB*16 channels of its pixels (the launcher forces the channel blocking for act 6 / 7):
    // aux1 = {w[0 .. 16*MB-1], bias}; act 7 applies a sigmoid to the result (the confidence head, itermvs.py:147-151,198)
    const __amdgpu_buffer_rsrc_t ro = epi_rsrc(e.out, (uint32_t)e.P * 4u);
    float wv[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[mb][r] = e.aux1[mb * 16 + q * 4 + r];
    const float bias = e.aux1[16 * MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float s = 0.0f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(fmaxf(acc[mb][nb][r], 0.0f), wv[mb][r], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        s += bias;
        if (e.act == 7) s = sigmoidf_(s);
        epi_store(s, ro, q == 0 ? pix_off[nb] : kEpiOob, 0);
    }
}

// The optional residual operand is a template parameter as well: a run-time `if (add)` around its loads
// makes the compiler wait for vmcnt(0) in front of EVERY pixel slot -- i.e. for the previous slot's stores.
// py / px: output coordinates of this lane's pixel per slot (only read by the bilinear residual).
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue(const EpilogueArgs& e, const f32x4 (&acc)[MB][NB], int m0, int q,
                                              const uint32_t (&pix_off)[NB], const int (&py)[NB], const int (&px)[NB]) {
    if (e.out_nhwc) return conv_epilogue_nhwc<MB, NB>(e, acc, m0, q, pix_off);
    if (e.act == 6 || e.act == 7) return conv_epilogue_dot<MB, NB>(e, acc, q, pix_off);
    const int key = e.act * 3 + (e.add ? 1 + e.add_mode : 0);
    switch (key) {
        case 0: conv_epilogue_act<0, 0, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 1: conv_epilogue_act<0, 1, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 2: conv_epilogue_act<0, 2, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 3: conv_epilogue_act<1, 0, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 4: conv_epilogue_act<1, 1, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 6: conv_epilogue_act<2, 0, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 9: conv_epilogue_act<3, 0, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 12: conv_epilogue_act<4, 0, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        case 15: conv_epilogue_act<5, 0, MB, NB>(e, acc, m0, q, pix_off, py, px); break;
        default: break;   // other combinations are rejected on the host (itermvs_conv2d)
    }
}

}  // namespace itermvs
