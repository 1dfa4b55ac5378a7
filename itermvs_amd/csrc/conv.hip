// Direct 2-D convolutions for the SMALL-channel layers of the hot path (CorrNet 8..32 channels,
// PixelViewWeight, ConvGRU 43->32, heads, FeatureNet 3..48 channels) -- models/itermvs.py:333-381,
// models/module.py:6-66, models/net.py:7-66.
//
// Why not MIOpen: at these sizes (0.05-1 GFLOP per layer) the library kernels are launch/latency
// bound (15-60 us each, plus NCHW<->NHWC transposes it inserts and separate bias / ReLU / add
// kernels); a step of the reference network spends 94 % of its GPU time there (profiles/r01_*).
//
// Design (fp32, NCHW planes, wave64):
//   * thread = one output pixel (2x2 output pixels for the stride-2 transposed conv), lanes run
//     along x so every input tap is one coalesced 256-byte wave load;
//   * a block handles CT output channels held in CT accumulators per thread; the CT weights of a
//     (ci, ky, kx) are contiguous in the packed [Cin][k][k][Cout] layout and wave-uniform, so they
//     are fetched with scalar loads and enter v_fmac as SGPR operands: the inner loop is pure
//     VALU FMA, one vector load per CT FMAs;
//   * bias, residual add, ReLU / sigmoid / tanh and the ConvGRU gate formulas are fused in the
//     epilogue (module.py:59-66), so activations are written exactly once;
//   * up to three weight sets per launch: the three CorrNets of one iteration (levels 1..3, batch
//     items [0,4), [4,8), [8,10)) run as ONE launch per layer instead of three.
// Numerics: plain fp32 fma chains in (ci, ky, kx) order -- same class of rounding as any other
// convolution back-end; parity is checked against F.conv2d in tests/test_conv_gpu.py.
#include <stdlib.h>

#include "common.hpp"

namespace itermvs {

struct ConvArgs {
    const float* in;
    float* out;
    float* out2;          // optional second copy of the result (contiguous [N,Cout,P])
    const float* add;     // residual, added before the activation
    const float* aux1;    // epilogue operand (h for the GRU forms)
    const float* aux2;    // epilogue operand (z for the GRU update)
    int64_t in_sn, out_sn, add_sn, aux1_sn, aux2_sn;
    const float* weight[3];
    const float* bias[3];
    int seg_end[3];
    int N, Cin, Hin, Win, Cout, Hout, Wout;
    int stride, pad, dil, act;
};

__device__ __forceinline__ float epilogue(float v, int act, float add, float a1, float a2) {
    v += add;
    switch (act) {
        case 1: return fmaxf(v, 0.0f);
        case 2: return sigmoidf_(v);
        case 3: return tanhf(v);
        case 4: return sigmoidf_(v) * a1;                       // r * h            (module.py:63-64)
        case 5: return (1.0f - a2) * a1 + a2 * tanhf(v);        // (1-z) h + z q    (module.py:64-65)
        default: return v;
    }
}

template <int CT, int KS>
__global__ void __launch_bounds__(256) conv_direct_kernel(const ConvArgs a) {
    const int P = a.Hout * a.Wout;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * CT;
    const int n = blockIdx.z;
    const int seg = (n >= a.seg_end[0]) + (n >= a.seg_end[1]);
    const float* __restrict__ w = a.weight[seg] + co0;
    const float* __restrict__ bias = a.bias[seg];
    if (p >= P) return;
    const int oy = p / a.Wout, ox = p - oy * a.Wout;

    int off[KS * KS];
    bool ok[KS * KS];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int iy = oy * a.stride - a.pad + ky * a.dil;
            const int ix = ox * a.stride - a.pad + kx * a.dil;
            const bool in = iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            ok[ky * KS + kx] = in;
            off[ky * KS + kx] = in ? iy * a.Win + ix : 0;
        }
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = bias ? bias[co0 + c] : 0.0f;

    const float* __restrict__ ip = a.in + (int64_t)n * a.in_sn;
    const int plane = a.Hin * a.Win;
    const int wstep = KS * KS * a.Cout;
    for (int ci = 0; ci < a.Cin; ++ci) {
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
            const float v = ok[t] ? ip[off[t]] : 0.0f;
            const float* __restrict__ wt = w + t * a.Cout;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = fmaf(v, wt[c], acc[c]);
        }
        ip += plane;
        w += wstep;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int64_t ch = (int64_t)(co0 + c) * P + p;
        const float ad = a.add ? a.add[(int64_t)n * a.add_sn + ch] : 0.0f;
        const float a1 = a.aux1 ? a.aux1[(int64_t)n * a.aux1_sn + ch] : 0.0f;
        const float a2 = a.aux2 ? a.aux2[(int64_t)n * a.aux2_sn + ch] : 0.0f;
        const float r = epilogue(acc[c], a.act, ad, a1, a2);
        a.out[(int64_t)n * a.out_sn + ch] = r;
        if (a.out2) a.out2[((int64_t)n * a.Cout) * P + ch] = r;
    }
}

template <int KS>
static int launch_direct(const ConvArgs& a, int ct, hipStream_t stream) {
    const int P = a.Hout * a.Wout;
    const dim3 grid((P + 255) / 256, a.Cout / ct, a.N);
    switch (ct) {
        case 32: hipLaunchKernelGGL((conv_direct_kernel<32, KS>), grid, dim3(256), 0, stream, a); break;
        case 16: hipLaunchKernelGGL((conv_direct_kernel<16, KS>), grid, dim3(256), 0, stream, a); break;
        case 8: hipLaunchKernelGGL((conv_direct_kernel<8, KS>), grid, dim3(256), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((conv_direct_kernel<4, KS>), grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL((conv_direct_kernel<1, KS>), grid, dim3(256), 0, stream, a); break;
    }
    return itermvs_launch_status();
}

}  // namespace itermvs

using namespace itermvs;

int itermvs_conv2d_mfma(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream);      // conv_mfma.hip
int itermvs_conv2d_tile(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream);      // conv_tile.hip
int itermvs_deconv2d_tile(const itermvs_conv_params* p, hipStream_t stream);                        // conv_tile.hip
int itermvs_conv2d_tile3(const itermvs_conv_params* p, int hout, int wout, hipStream_t stream);     // conv_tile3.hip

static int conv2d_impl(const itermvs_conv_params* p, void* stream);

extern "C" int itermvs_conv2d(const itermvs_conv_params* p, void* stream) {
    itermvs_profile_begin(3, (hipStream_t)stream);
    const int rc = conv2d_impl(p, stream);
    itermvs_profile_end(3, (hipStream_t)stream);
    return rc;
}

static int conv2d_impl(const itermvs_conv_params* p, void* stream) {
    ITERMVS_RETURN_IF(!p, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(!p->in || !p->out || !p->weight[0], ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->N < 1 || p->Cin < 1 || p->Cout < 1 || p->Hin < 1 || p->Win < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->ksize != 1 && p->ksize != 3, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->n_seg < 1 || p->n_seg > 3 || p->act < 0 || p->act > 7, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((p->act == 6 || p->act == 7) && (!p->aux1 || p->add || p->out2 || p->out_layout != 0 || (p->Cout != 16 && p->Cout != 32) || (p->weight_format != 2 && p->weight_format != 3) || p->ksize != 3 ||
                                      p->split_cout != 0 || p->transposed || p->n_seg != 1), ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((p->act == 4 || p->act == 5 || p->act == 6 || p->act == 7) && !p->aux1, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->act == 5 && !p->aux2, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(p->act >= 2 && p->add, ITERMVS_ERR_DIMS);   // residual add only with none / relu
    ITERMVS_RETURN_IF(p->add_mode < 0 || p->add_mode > 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->out_layout < 0 || p->out_layout > 3, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->in_layout != 0 && p->in_layout != 1, ITERMVS_ERR_LAYOUT);
    // channels-last input: the 8-channel tap-pair form of the bf16x3 kernel only (conv_tile3.hip)
    ITERMVS_RETURN_IF(p->in_layout == 1 && (p->weight_format != 3 || p->ksize != 3 || p->Cin != 8 || p->stride != 1 || p->dilation != 1 || p->transposed),
                      ITERMVS_ERR_LAYOUT);
    ITERMVS_RETURN_IF(p->in_layout == 1 && (((uintptr_t)p->in) % 16 || p->in_sn % 4), ITERMVS_ERR_ALIGN);
    if (p->split_cout != 0) {
        ITERMVS_RETURN_IF((p->weight_format != 2 && p->weight_format != 3) || p->transposed || p->out_layout != 0 || p->add || p->out2, ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(p->split_cout < 16 || p->split_cout >= p->Cout || (p->split_cout & 15), ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(!p->out_b || p->act_b < 0 || p->act_b > 4 || p->act > 4, ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(p->act_b == 4 && !p->aux1, ITERMVS_ERR_NULL);
    }
    ITERMVS_RETURN_IF(p->out_layout >= 1 && (p->weight_format == 0 || p->transposed || p->act != 0 || p->add || (p->Cout & 3)),
                      ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->add_mode == 1 && (!p->add || p->weight_format == 0 || p->transposed || p->act != 0), ITERMVS_ERR_DIMS);
    ConvArgs a;
    a.in = p->in; a.out = p->out; a.out2 = p->out2; a.add = p->add; a.aux1 = p->aux1; a.aux2 = p->aux2;
    a.in_sn = p->in_sn; a.out_sn = p->out_sn; a.add_sn = p->add_sn; a.aux1_sn = p->aux1_sn; a.aux2_sn = p->aux2_sn;
    for (int i = 0; i < 3; ++i) {
        const int k = i < p->n_seg ? i : p->n_seg - 1;
        ITERMVS_RETURN_IF(!p->weight[k], ITERMVS_ERR_NULL);
        a.weight[i] = p->weight[k];
        a.bias[i] = p->bias[k];
        a.seg_end[i] = i < p->n_seg - 1 ? p->seg_end[i] : p->N;
    }
    a.N = p->N; a.Cin = p->Cin; a.Hin = p->Hin; a.Win = p->Win; a.Cout = p->Cout;
    a.stride = p->stride; a.pad = p->pad; a.dil = p->dilation; a.act = p->act;
    // largest channel tile that divides Cout ...
    int ct = 1;
    for (int c : {32, 16, 8, 4})
        if (p->Cout % c == 0) { ct = c; break; }
    if (p->transposed && p->weight_format == 2) {
        const int rc = itermvs_deconv2d_tile(p, (hipStream_t)stream);
        return rc == 1 ? ITERMVS_ERR_DIMS : rc;
    }
    ITERMVS_RETURN_IF(p->transposed, ITERMVS_ERR_DIMS);   // transposed convolutions exist in weight_format 2 only
    ITERMVS_RETURN_IF(p->stride < 1 || p->dilation < 1 || p->pad < 0, ITERMVS_ERR_DIMS);
    const int span = (p->ksize - 1) * p->dilation + 1;
    a.Hout = (p->Hin + 2 * p->pad - span) / p->stride + 1;
    a.Wout = (p->Win + 2 * p->pad - span) / p->stride + 1;
    ITERMVS_RETURN_IF(a.Hout < 1 || a.Wout < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(p->add_mode == 1 && ((a.Hout | a.Wout) & 1), ITERMVS_ERR_DIMS);
    if (p->weight_format == 2) {   // LDS-tiled 3x3 kernels; the packed layout fits no other kernel
        const int rc = itermvs_conv2d_tile(p, a.Hout, a.Wout, (hipStream_t)stream);
        return rc == 1 ? ITERMVS_ERR_DIMS : rc;
    }
    if (p->weight_format == 3) {   // bf16x3 split on the bf16 MFMA (3x3, more than 8 input channels)
        const int rc = itermvs_conv2d_tile3(p, a.Hout, a.Wout, (hipStream_t)stream);
        return rc == 1 ? ITERMVS_ERR_DIMS : rc;
    }
    if (p->weight_format == 1) return itermvs_conv2d_mfma(p, a.Hout, a.Wout, (hipStream_t)stream);
    // ... that still leaves >= 1024 workgroups (the VALU kernel keeps all CT channels in one thread)
    while (ct > 4 && (int64_t)((a.Hout * a.Wout + 255) / 256) * (p->Cout / ct) * p->N < 1024) ct /= 2;
    return p->ksize == 3 ? launch_direct<3>(a, ct, (hipStream_t)stream) : launch_direct<1>(a, ct, (hipStream_t)stream);
}
