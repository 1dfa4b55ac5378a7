// itermvs_conv2d, matrix-core path: implicit-GEMM 2-D convolution on v_mfma_f32_16x16x4_f32.
//
// The convolutions of the path are small dense contractions (K = Cin*k*k = 27..432, Cout = 8..256,
// 5k..400k pixels).  A one-thread-per-pixel VALU kernel cannot fill 256 CUs on them (the ConvGRU
// gate conv is 253 MFLOP-equivalents but only 320 waves), so they run on the matrix cores:
//   D[cout, pixel] += W[cout, k] * X[k, pixel],  k = (tap, ci),  16 x 16 x 4 per instruction,
// f32 in / f32 accumulate -- bit-for-bit a k-ordered fmaf chain (exact fp32, no TF32 on gfx950).
//   * A operand = weights: lane l holds W[cout = m0 + (l & 15)][k = 4t + (l >> 4)], read from the packed
//     [tap][Cin_pad][Cout_pad] layout (64-byte runs, identical for every wave -> L1/L2 resident);
//   * B operand = inputs: lane l holds X[k][pixel = p0 + (l & 15)], gathered straight from the NCHW
//     planes (16 consecutive pixels = one 64-byte run; zero for out-of-image taps);
//   * a wave owns MB x NB tiles (16*MB output channels x 16*NB pixels): MB + NB loads feed MB*NB MFMAs;
//   * D has pixels along lanes (col = lane & 15), so every store instruction writes 64-byte runs of
//     one output plane; bias / residual / activation / ConvGRU gate math are applied on the
//     accumulators (same epilogues as conv.hip).
// K is ordered (tap, ci) with Cin padded to a multiple of 4 (zero weights), so the four k-slots of a
// step share one tap: the spatial offset is computed once per tap and the inner loop is
// `load A, load B, mfma` with two pointer bumps.
#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct MfmaArgs {
    const float* in;
    float* out;
    float* out2;
    const float* add;
    const float* aux1;
    const float* aux2;
    int64_t in_sn, out_sn, add_sn, aux1_sn, aux2_sn;
    const float* weight[3];   // packed [k*k][CinPad][CoutPad]
    const float* bias[3];
    int seg_end[3];
    int N, Cin, CinPad, Hin, Win, Cout, CoutPad, Hout, Wout;
    int ksize, stride, pad, dil, act;
};

__device__ __forceinline__ float mfma_epilogue(float v, int act, float add, float a1, float a2) {
    v += add;
    switch (act) {
        case 1: return fmaxf(v, 0.0f);
        case 2: return sigmoidf_(v);
        case 3: return tanhf(v);
        case 4: return sigmoidf_(v) * a1;                  // r * h            (module.py:63-64)
        case 5: return (1.0f - a2) * a1 + a2 * tanhf(v);   // (1-z) h + z q    (module.py:64-65)
        default: return v;
    }
}

template <int MB, int NB>
__global__ void __launch_bounds__(256) conv_mfma_kernel(const MfmaArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kslot = lane >> 4, l16 = lane & 15;
    const int n = blockIdx.z;
    const int seg = (n >= a.seg_end[0]) + (n >= a.seg_end[1]);
    const int P = a.Hout * a.Wout;
    const int m0 = blockIdx.y * (MB * 16);
    const int pbase = (blockIdx.x * 4 + wave) * (NB * 16);
    if (pbase >= P) return;   // whole wave; no block-level synchronisation in this kernel

    int oy[NB], ox[NB];
    bool pv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int p = pbase + nb * 16 + l16;
        pv[nb] = p < P;
        const int pc = pv[nb] ? p : 0;
        oy[nb] = pc / a.Wout;
        ox[nb] = pc - oy[nb] * a.Wout;
    }
    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int plane = a.Hin * a.Win;
    const float* __restrict__ inb = a.in + (int64_t)n * a.in_sn;
    const float* __restrict__ wb = a.weight[seg] + (size_t)kslot * a.CoutPad + m0 + l16;
    const int taps = a.ksize * a.ksize;
    for (int tap = 0; tap < taps; ++tap) {
        const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
        int off[NB];
        bool ok[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int iy = oy[nb] * a.stride - a.pad + ky * a.dil;
            const int ix = ox[nb] * a.stride - a.pad + kx * a.dil;
            ok[nb] = pv[nb] && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            off[nb] = ok[nb] ? iy * a.Win + ix : 0;
        }
        const float* __restrict__ wt = wb + (size_t)tap * a.CinPad * a.CoutPad;
        for (int c0 = 0; c0 < a.CinPad; c0 += 4) {
            const int ci = min(c0 + kslot, a.Cin - 1);       // padded channels: weight is zero
            const float* __restrict__ ip = inb + (int64_t)ci * plane;
            float av[MB], bv[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = wt[(size_t)c0 * a.CoutPad + mb * 16];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float x = ip[off[nb]];
                bv[nb] = ok[nb] ? x : 0.0f;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        }
    }
    // D: col (pixel) = lane & 15, row (cout) = (lane >> 4) * 4 + r
    const float* __restrict__ bias = a.bias[seg];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + mb * 16 + kslot * 4 + r;
            if (co >= a.Cout) continue;
            const float bs = bias ? bias[co] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (!pv[nb]) continue;
                const int p = pbase + nb * 16 + l16;
                const int64_t ch = (int64_t)co * P + p;
                const float ad = a.add ? a.add[(int64_t)n * a.add_sn + ch] : 0.0f;
                const float a1 = a.aux1 ? a.aux1[(int64_t)n * a.aux1_sn + ch] : 0.0f;
                const float a2 = a.aux2 ? a.aux2[(int64_t)n * a.aux2_sn + ch] : 0.0f;
                const float v = mfma_epilogue(acc[mb][nb][r] + bs, a.act, ad, a1, a2);
                a.out[(int64_t)n * a.out_sn + ch] = v;
                if (a.out2) a.out2[((int64_t)n * a.Cout) * P + ch] = v;
            }
        }
}

}  // namespace itermvs

using namespace itermvs;

// called from itermvs_conv2d (conv.hip) for the non-transposed convolutions when weight_format == 1
int itermvs_conv2d_mfma(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream) {
    MfmaArgs a;
    a.in = p->in; a.out = p->out; a.out2 = p->out2; a.add = p->add; a.aux1 = p->aux1; a.aux2 = p->aux2;
    a.in_sn = p->in_sn; a.out_sn = p->out_sn; a.add_sn = p->add_sn; a.aux1_sn = p->aux1_sn; a.aux2_sn = p->aux2_sn;
    for (int i = 0; i < 3; ++i) {
        const int k = i < p->n_seg ? i : p->n_seg - 1;
        a.weight[i] = p->weight[k];
        a.bias[i] = p->bias[k];
        a.seg_end[i] = i < p->n_seg - 1 ? p->seg_end[i] : p->N;
    }
    a.N = p->N; a.Cin = p->Cin; a.CinPad = (p->Cin + 3) / 4 * 4; a.Hin = p->Hin; a.Win = p->Win;
    a.Cout = p->Cout; a.CoutPad = (p->Cout + 15) / 16 * 16; a.Hout = hout; a.Wout = wout;
    a.ksize = p->ksize; a.stride = p->stride; a.pad = p->pad; a.dil = p->dilation; a.act = p->act;
    const int P = hout * wout;
    const int mt = a.CoutPad / 16;
    // largest register blocking that still yields >= 4096 waves (fills 256 CUs x 4 SIMDs x 4 waves)
    struct Cfg { int mb, nb; };
    const Cfg cfgs[] = {{2, 4}, {2, 2}, {1, 2}, {1, 1}};
    Cfg pick = cfgs[3];
    for (const Cfg& c : cfgs) {
        if (c.mb > mt || (mt % c.mb) != 0) continue;
        const int64_t waves = (int64_t)p->N * ((P + 16 * c.nb - 1) / (16 * c.nb)) * (mt / c.mb);
        if (waves >= 4096) { pick = c; break; }
    }
    const int px_per_block = 4 * 16 * pick.nb;
    const dim3 grid((P + px_per_block - 1) / px_per_block, mt / pick.mb, p->N);
    if (pick.mb == 2 && pick.nb == 4) hipLaunchKernelGGL((conv_mfma_kernel<2, 4>), grid, dim3(256), 0, stream, a);
    else if (pick.mb == 2 && pick.nb == 2) hipLaunchKernelGGL((conv_mfma_kernel<2, 2>), grid, dim3(256), 0, stream, a);
    else if (pick.mb == 1 && pick.nb == 2) hipLaunchKernelGGL((conv_mfma_kernel<1, 2>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_mfma_kernel<1, 1>), grid, dim3(256), 0, stream, a);
    return itermvs_launch_status();
}
