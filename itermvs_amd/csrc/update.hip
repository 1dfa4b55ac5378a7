// Per-iteration probability update: soft-max over 256 depth bins, first-max arg-max, +-4 window
// regression (models/itermvs.py:171-190, 201-219), ConvGRU gate math (models/module.py:59-66),
// convex x4 up-sampling (models/module.py:127-140 + itermvs.py:262-264) and integer-factor
// bilinear up-sampling (F.interpolate, align_corners=False).
#include "common.hpp"

namespace itermvs {

// ---------------------------------------------------------------------------------------------
// prob_regress.  Block = 256 threads = TP pixels x 8 bin groups of 32 bins.  Each thread keeps
// its 32 logits in registers (read exactly once, coalesced along pixels); block-level max / sum /
// arg-max go through a few hundred bytes of LDS.  The 256-bin probability volume only goes
// to memory if the caller asks for it (training).
// ---------------------------------------------------------------------------------------------
constexpr int kBins = ITERMVS_PROB_BINS;
constexpr int kTP = 32;               // pixels per block
constexpr int kGroups = 256 / kTP;    // bin groups per block (8)
constexpr int kPer = kBins / kGroups; // bins per thread (32)
constexpr int kWin = 2 * ITERMVS_WINDOW_RADIUS + 1;

__global__ void __launch_bounds__(256) prob_regress_kernel(const float* __restrict__ logits, int64_t sb, int64_t sc,
                                                           int64_t sp, int P, float* __restrict__ nd0, int64_t nd_sb0,
                                                           float* __restrict__ nd1, int64_t nd_sb1,
                                                           float* __restrict__ prob, int64_t* __restrict__ best_out) {
    __shared__ float red[kGroups][kTP];
    __shared__ int redi[kGroups][kTP];
    __shared__ float win[kWin][kTP];
    const int px = threadIdx.x % kTP;
    const int grp = threadIdx.x / kTP;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kTP + px;
    const bool live = p < P;
    const int k0 = grp * kPer;

    float v[kPer];
    const float* lp = logits + b * sb + (int64_t)(live ? p : 0) * sp + (int64_t)k0 * sc;
#pragma unroll
    for (int i = 0; i < kPer; ++i) v[i] = lp[i * sc];

    float m = v[0];
#pragma unroll
    for (int i = 1; i < kPer; ++i) m = fmaxf(m, v[i]);
    red[grp][px] = m;
    __syncthreads();
    m = red[0][px];
#pragma unroll
    for (int g = 1; g < kGroups; ++g) m = fmaxf(m, red[g][px]);
    __syncthreads();

    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        v[i] = expf(v[i] - m);
        s += v[i];
    }
    red[grp][px] = s;
    __syncthreads();
    s = red[0][px];
#pragma unroll
    for (int g = 1; g < kGroups; ++g) s += red[g][px];
    __syncthreads();

    // probabilities; first maximal index within this thread's bins
    float bv = -1.0f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        v[i] = v[i] / s;
        if (v[i] > bv) {
            bv = v[i];
            bi = k0 + i;
        }
    }
    red[grp][px] = bv;
    redi[grp][px] = bi;
    __syncthreads();
    bv = red[0][px];
    bi = redi[0][px];
#pragma unroll
    for (int g = 1; g < kGroups; ++g)
        if (red[g][px] > bv) {  // strict: lower index wins ties (torch.argmax first-max rule)
            bv = red[g][px];
            bi = redi[g][px];
        }
    // publish the (unclamped) window k*-4 .. k*+4
    const int lo = bi - ITERMVS_WINDOW_RADIUS;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int off = k0 + i - lo;
        if (off >= 0 && off < kWin) win[off][px] = v[i];
    }
    if (prob && live) {
        float* pp = prob + ((size_t)b * kBins + k0) * P + p;
#pragma unroll
        for (int i = 0; i < kPer; ++i) pp[(size_t)i * P] = v[i];
    }
    __syncthreads();
    if (grp == 0 && live) {
        float num = 0.0f, den = 1e-6f;  // itermvs.py:212
        for (int i = 0; i < kWin; ++i) {
            int k = lo + i;
            k = k < 0 ? 0 : (k > kBins - 1 ? kBins - 1 : k);  // clamp; duplicates double-counted
            const float pk = win[k - lo][px];
            num = num + (float)k * pk;
            den = den + pk;
        }
        const float nd = (num / den) / (float)(kBins - 1);
        if (nd0) nd0[b * nd_sb0 + p] = nd;
        if (nd1) nd1[b * nd_sb1 + p] = nd;
        if (best_out) best_out[(size_t)b * P + p] = bi;
    }
}

// ---------------------------------------------------------------------------------------------
// ConvGRU gate math, element-wise form (the inference engine fuses it into the epilogue of the dilated 3x3 convolutions,
// conv_epilogue.hpp; this form serves the traced / training path whose convolutions run in PyTorch-ROCm autograd)
// ---------------------------------------------------------------------------------------------
__global__ void gru_rh_kernel(const float* __restrict__ zr, const float* __restrict__ h, int64_t h_sb,
                              float* __restrict__ rh, int64_t rh_sb, int B, int hid, int P) {
    const int64_t per = (int64_t)hid * P;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * per) return;
    const int b = (int)(t / per);
    const int64_t r = t - (int64_t)b * per;
    const float rg = sigmoidf_(zr[(size_t)b * 2 * per + per + r]);
    rh[b * rh_sb + r] = rg * h[b * h_sb + r];
}

__global__ void gru_out_kernel(const float* __restrict__ zr, const float* __restrict__ q, float* __restrict__ h,
                               int64_t h_sb, float* __restrict__ h_copy, int B, int hid, int P) {
    const int64_t per = (int64_t)hid * P;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * per) return;
    const int b = (int)(t / per);
    const int64_t r = t - (int64_t)b * per;
    const float z = sigmoidf_(zr[(size_t)b * 2 * per + r]);
    const float qv = tanhf(q[t]);
    const float hv = h[b * h_sb + r];
    const float hn = (1.0f - z) * hv + z * qv;  // module.py:65
    h[b * h_sb + r] = hn;
    if (h_copy) h_copy[t] = hn;
}

__global__ void pack_scores_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                   const float* __restrict__ s2, int n0, int n1, int n2, int B, int P,
                                   float* __restrict__ d0, float* __restrict__ d1, int64_t dst_sb, int ch0) {
    const int nt = n0 + n1 + n2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * nt * P) return;
    const int p = (int)(t % P);
    const int c = (int)((t / P) % nt);
    const int b = (int)(t / ((int64_t)P * nt));
    float v;
    if (c < n0) v = s0[((size_t)b * n0 + c) * P + p];
    else if (c < n0 + n1) v = s1[((size_t)b * n1 + (c - n0)) * P + p];
    else v = s2[((size_t)b * n2 + (c - n0 - n1)) * P + p];
    const int64_t o = b * dst_sb + (int64_t)(ch0 + c) * P + p;
    if (d0) d0[o] = v;
    if (d1) d1[o] = v;
}

// ---------------------------------------------------------------------------------------------
// convex x4 up-sampling fused with the 9-tap soft-max and depth un-normalisation.
// thread = (b, y, sub-row i, x): 36 logits + 9 neighbours in, one float4 (4 sub-columns) out.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void convex_upsample_body(const float* __restrict__ logits, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                                     const float* __restrict__ nd, int64_t nd_sb, const float* __restrict__ inv_min,
                                                     const float* __restrict__ inv_max, int B, int H, int W,
                                                     float* __restrict__ depth, float* __restrict__ norm_out, int64_t t) {
    if (t >= (int64_t)B * H * 4 * W) return;
    const int x = (int)(t % W);
    const int i = (int)((t / W) % 4);
    const int y = (int)((t / ((int64_t)4 * W)) % H);
    const int b = (int)(t / ((int64_t)4 * W * H));
    float nb[9];
    const float* ndb = nd + b * nd_sb;
#pragma unroll
    for (int k = 0; k < 9; ++k) {  // replicate padding (module.py:133) + unfold order (k = ky*3+kx)
        int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        nb[k] = ndb[(size_t)yy * W + xx];
    }
    const float* lp = logits + b * sb + y * sy + x * sx + (int64_t)(i * 4) * sc;
    const float imin = inv_min[b], imax = inv_max[b];
    float res[4], resn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float l[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) l[k] = lp[(int64_t)(k * 16 + j) * sc];
        float m = l[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) m = fmaxf(m, l[k]);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            l[k] = expf(l[k] - m);
            s += l[k];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc = acc + nb[k] * (l[k] / s);
        resn[j] = acc;
        res[j] = unnormalize_depth(acc, imin, imax);
    }
    const size_t o = ((size_t)b * 4 * H + (4 * y + i)) * (4 * (size_t)W) + 4 * (size_t)x;
    *reinterpret_cast<float4*>(depth + o) = make_float4(res[0], res[1], res[2], res[3]);
    if (norm_out) *reinterpret_cast<float4*>(norm_out + o) = make_float4(resn[0], resn[1], resn[2], resn[3]);
}

// blocks [0, n_convex): convex up-sampling of the depth; blocks after them (if any): bilinear up-sampling of another map
// (the confidence, itermvs.py:323-324) -- two independent pieces of work in one launch (itermvs_final_upsample)
__global__ void convex_upsample_kernel(const float* __restrict__ logits, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                       const float* __restrict__ nd, int64_t nd_sb, const float* __restrict__ inv_min,
                                       const float* __restrict__ inv_max, int B, int H, int W,
                                       float* __restrict__ depth, float* __restrict__ norm_out, int n_convex,
                                       const float* __restrict__ x2, int M2, int scale2, float* __restrict__ out2) {
    if ((int)blockIdx.x < n_convex)
        convex_upsample_body(logits, sb, sc, sy, sx, nd, nd_sb, inv_min, inv_max, B, H, W, depth, norm_out,
                             (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    else
        bilinear_up_body(x2, M2, H, W, scale2, 0, out2, (int64_t)(blockIdx.x - n_convex) * blockDim.x + threadIdx.x);
}

__global__ void bilinear_up_kernel(const float* __restrict__ x, int M, int H, int W, int scale, int act,
                                   float* __restrict__ out) {
    bilinear_up_body(x, M, H, W, scale, act, out, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// bilinear_up with up to two destinations addressed by batch strides (channel slices of wider buffers):
// the initial hidden state goes to `hidden` and to channels 0..31 of the GRU input buffer in one launch
__global__ void bilinear_up2_kernel(const float* __restrict__ x, int B, int C, int H, int W, int scale, int act,
                                    float* __restrict__ out, int64_t out_sb, float* __restrict__ out2, int64_t out2_sb) {
    const int OH = H * scale, OW = W * scale;
    const int64_t per = (int64_t)C * OH * OW;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * per) return;
    const int b = (int)(t / per);
    const int64_t r = t - (int64_t)b * per;
    const int ox = (int)(r % OW);
    const int oy = (int)((r / OW) % OH);
    const int c = (int)(r / ((int64_t)OW * OH));
    const float rs = 1.0f / (float)scale;
    float sy = ((float)oy + 0.5f) * rs - 0.5f;
    float sx = ((float)ox + 0.5f) * rs - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > H - 1 ? H - 1 : y0;
    x0 = x0 > W - 1 ? W - 1 : x0;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* xm = x + ((size_t)b * C + c) * H * W;
    const float top = xm[(size_t)y0 * W + x0] * lx0 + xm[(size_t)y0 * W + x1] * lx1;
    const float bot = xm[(size_t)y1 * W + x0] * lx0 + xm[(size_t)y1 * W + x1] * lx1;
    float v = top * ly0 + bot * ly1;
    if (act == 1) v = tanhf(v);
    out[b * out_sb + r] = v;
    if (out2) out2[b * out2_sb + r] = v;
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_bilinear_up2(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, int32_t scale, int32_t act,
                                    float* out, int64_t out_sb, float* out2, int64_t out2_sb, void* stream) {
    ITERMVS_RETURN_IF(!x || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || C < 1 || H < 1 || W < 1 || scale < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * C * H * W * scale * scale;
    hipLaunchKernelGGL(bilinear_up2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       B, C, H, W, scale, act, out, out_sb, out2, out2_sb);
    return itermvs_launch_status();
}

extern "C" int itermvs_prob_regress(const float* logits, int64_t sb, int64_t sc, int64_t sp, int32_t B, int32_t P,
                                    float* nd_out0, int64_t nd_sb0, float* nd_out1, int64_t nd_sb1, float* prob,
                                    int64_t* best, void* stream) {
    ITERMVS_RETURN_IF(!logits, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || P < 1, ITERMVS_ERR_DIMS);
    hipLaunchKernelGGL(prob_regress_kernel, dim3((P + kTP - 1) / kTP, B), dim3(256), 0, (hipStream_t)stream, logits, sb,
                       sc, sp, P, nd_out0, nd_sb0, nd_out1, nd_sb1, prob, best);
    return itermvs_launch_status();
}

extern "C" int itermvs_gru_rh(const float* zr, const float* h, int64_t h_sb, float* rh, int64_t rh_sb, int32_t B,
                              int32_t hid, int32_t P, void* stream) {
    ITERMVS_RETURN_IF(!zr || !h || !rh, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || hid < 1 || P < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * hid * P;
    hipLaunchKernelGGL(gru_rh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, zr, h,
                       h_sb, rh, rh_sb, B, hid, P);
    return itermvs_launch_status();
}

extern "C" int itermvs_gru_out(const float* zr, const float* q, float* h, int64_t h_sb, float* h_copy, int32_t B,
                               int32_t hid, int32_t P, void* stream) {
    ITERMVS_RETURN_IF(!zr || !q || !h, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || hid < 1 || P < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * hid * P;
    hipLaunchKernelGGL(gru_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, zr, q,
                       h, h_sb, h_copy, B, hid, P);
    return itermvs_launch_status();
}

extern "C" int itermvs_pack_scores(const float* s0, const float* s1, const float* s2, const int32_t N[3], int32_t B,
                                   int32_t P, float* dst0, float* dst1, int64_t dst_sb, int32_t ch0, void* stream) {
    ITERMVS_RETURN_IF(!s0 || !s1 || !s2 || !N || (!dst0 && !dst1), ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || P < 1 || N[0] < 1 || N[1] < 1 || N[2] < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * (N[0] + N[1] + N[2]) * P;
    hipLaunchKernelGGL(pack_scores_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s0,
                       s1, s2, N[0], N[1], N[2], B, P, dst0, dst1, dst_sb, ch0);
    return itermvs_launch_status();
}

extern "C" int itermvs_convex_upsample(const float* logits, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                       const float* nd, int64_t nd_sb, const float* inv_depth_min,
                                       const float* inv_depth_max, int32_t B, int32_t H, int32_t W, float* depth,
                                       float* norm_out, void* stream) {
    ITERMVS_RETURN_IF(!logits || !nd || !inv_depth_min || !inv_depth_max || !depth, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || H < 1 || W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(((uintptr_t)depth) % 16 || ((uintptr_t)norm_out) % 16, ITERMVS_ERR_ALIGN);
    const int64_t total = (int64_t)B * H * 4 * W;
    const int nc = (int)((total + 255) / 256);
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)nc), dim3(256), 0, (hipStream_t)stream,
                       logits, sb, sc, sy, sx, nd, nd_sb, inv_depth_min, inv_depth_max, B, H, W, depth, norm_out, nc,
                       (const float*)nullptr, 0, 1, (float*)nullptr);
    return itermvs_launch_status();
}

extern "C" int itermvs_final_upsample(const float* logits, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                      const float* nd, int64_t nd_sb, const float* inv_depth_min,
                                      const float* inv_depth_max, int32_t B, int32_t H, int32_t W, float* depth,
                                      const float* conf, int32_t M, float* conf_up, void* stream) {
    ITERMVS_RETURN_IF(!logits || !nd || !inv_depth_min || !inv_depth_max || !depth || !conf || !conf_up, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || H < 1 || W < 1 || M < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(((uintptr_t)depth) % 16, ITERMVS_ERR_ALIGN);
    const int nc = (int)(((int64_t)B * H * 4 * W + 255) / 256);
    const int nb = (int)(((int64_t)M * H * W * 16 + 255) / 256);
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)(nc + nb)), dim3(256), 0, (hipStream_t)stream,
                       logits, sb, sc, sy, sx, nd, nd_sb, inv_depth_min, inv_depth_max, B, H, W, depth, (float*)nullptr, nc,
                       conf, M, 4, conf_up);
    return itermvs_launch_status();
}

extern "C" int itermvs_bilinear_up(const float* x, int32_t M, int32_t H, int32_t W, int32_t scale, int32_t act,
                                   float* out, void* stream) {
    ITERMVS_RETURN_IF(!x || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(M < 1 || H < 1 || W < 1 || scale < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)M * H * W * scale * scale;
    hipLaunchKernelGGL(bilinear_up_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       M, H, W, scale, act, out);
    return itermvs_launch_status();
}
