// Training-mode BatchNorm (+ ReLU) of the FeatureNet layers, forward and backward (models/module.py:33-50 ConvBnReLU /
// ConvBn in train() mode under train.py:194-243; torch.nn.BatchNorm2d semantics: batch statistics with the biased variance
// for the normalisation, running_mean / running_var updated with momentum and the UNBIASED variance, eps inside the sqrt).
//
// Pure HBM streaming over NCHW fp32 tensors (the largest is 210 MB at the cfg-4 batch): the vendor kernels behind
// F.batch_norm took 8.2 ms forward + 7.0 ms backward per training step for the 16 layers (21 % of the step) where the
// bytes need ~2 ms.  Here a workgroup owns a SLAB of up to 8192 contiguous floats of one (image, channel) plane -- 8 float4
// per thread, consecutive threads on consecutive 16-byte pieces:
//   forward   stats    slab kept in registers: sum -> slab mean -> sum of squared deviations (exact two-pass M2) -> partial
//             finalize per channel: partials merged pairwise (Chan et al.) in fp64 -> mean, 1/sqrt(var + eps), running stats
//             apply    y = [relu]((x - mean) * invstd * gamma + beta)
//   backward  stats    g = dy * [y > 0] (y recomputed with the forward's expression), partial sums of g and g * xhat
//             finalize dbeta = sum g, dgamma = sum g * xhat, per-channel coefficients
//             apply    dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))
// 3 passes over the tensor forward, 5 backward; no atomics, results independent of the launch geometry's timing.
#include "common.hpp"

namespace itermvs {

constexpr int kBnThreads = 256;
constexpr int kBnPer = 32;                          // floats per thread
constexpr int kBnSlab = kBnThreads * kBnPer;        // 8192 floats per workgroup

struct BnArgs {
    const float* x;
    const float* dy;
    float* out;                 // y (forward apply) or dx (backward apply)
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* invstd;
    float* ws;                  // partials
    const float* coef;          // backward: [C][3] = gamma * invstd, mean(g), mean(g * xhat)
    int N, C, HW, parts, relu, vec;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the workgroup, returned to every thread (red: 4 floats of LDS per call site, guarded by the barriers inside)
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

struct BnSlab {
    size_t base;
    int len;
};
__device__ __forceinline__ BnSlab bn_slab(const BnArgs& a) {
    const int chunk = blockIdx.x, c = blockIdx.y;
    const int n = chunk / a.parts, part = chunk - n * a.parts;
    BnSlab s;
    s.base = ((size_t)n * a.C + c) * (size_t)a.HW + (size_t)part * kBnSlab;
    s.len = min(kBnSlab, a.HW - part * kBnSlab);
    return s;
}

// element e of the thread's 32: slab index.  Vector form: 8 float4 at (tid + 256 i) * 4; scalar form: tid + 256 i
template <bool VEC>
__device__ __forceinline__ int bn_index(int e) {
    return VEC ? ((int)threadIdx.x + kBnThreads * (e >> 2)) * 4 + (e & 3) : (int)threadIdx.x + kBnThreads * e;
}
template <bool VEC>
__device__ __forceinline__ void bn_load(const float* __restrict__ p, int len, float (&v)[kBnPer]) {
    if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < kBnPer / 4; ++i) {
            const int idx = ((int)threadIdx.x + kBnThreads * i) * 4;
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (idx < len) t = *reinterpret_cast<const float4*>(p + idx);       // len % 4 == 0 in this form
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < kBnPer; ++e) {
            const int idx = (int)threadIdx.x + kBnThreads * e;
            v[e] = idx < len ? p[idx] : 0.0f;
        }
    }
}
template <bool VEC>
__device__ __forceinline__ void bn_store(float* __restrict__ p, int len, const float (&v)[kBnPer]) {
    if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < kBnPer / 4; ++i) {
            const int idx = ((int)threadIdx.x + kBnThreads * i) * 4;
            if (idx < len) *reinterpret_cast<float4*>(p + idx) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < kBnPer; ++e) {
            const int idx = (int)threadIdx.x + kBnThreads * e;
            if (idx < len) p[idx] = v[e];
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(kBnThreads) bn_stats_kernel(const BnArgs a) {
    __shared__ float red[4];
    const BnSlab s = bn_slab(a);
    float v[kBnPer];
    bn_load<VEC>(a.x + s.base, s.len, v);
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < kBnPer; ++e) sum += v[e];                       // out-of-range elements are zeros
    const float mean = block_sum(sum, red) / (float)s.len;
    float m2 = 0.0f;
#pragma unroll
    for (int e = 0; e < kBnPer; ++e) {
        const float d = bn_index<VEC>(e) < s.len ? v[e] - mean : 0.0f;
        m2 = fmaf(d, d, m2);
    }
    m2 = block_sum(m2, red);
    if (threadIdx.x == 0) {
        float* w = a.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        w[0] = (float)s.len; w[1] = mean; w[2] = m2;
    }
}

// one wave per channel: merge the slab partials (count, mean, M2)
__global__ void __launch_bounds__(64) bn_finalize_kernel(const float* __restrict__ ws, int chunks, float eps, float momentum,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int k = lane; k < chunks; k += 64) {
        const float* w = ws + ((size_t)c * chunks + k) * 3;
        const double nb = w[0], mb = w[1], sb = w[2];
        const double nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        m2 += sb + d * d * (n * nb / nn);
        n = nn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), sb = __shfl_xor(m2, o, 64);
        const double nn = n + nb;
        if (nn > 0.0) {
            const double d = mb - mean;
            mean += d * (nb / nn);
            m2 += sb + d * d * (n * nb / nn);
            n = nn;
        }
    }
    if (lane == 0) {
        const float var = (float)(m2 / n);                                  // biased: the normalisation's variance
        save_mean[c] = (float)mean;
        save_invstd[c] = 1.0f / sqrtf(var + eps);
        if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(m2 / fmax(n - 1.0, 1.0));
    }
}

template <bool VEC>
__global__ void __launch_bounds__(kBnThreads) bn_apply_kernel(const BnArgs a) {
    const BnSlab s = bn_slab(a);
    const int c = blockIdx.y;
    const float mean = a.mean[c], invstd = a.invstd[c], gamma = a.gamma[c], beta = a.beta[c];
    float v[kBnPer];
    bn_load<VEC>(a.x + s.base, s.len, v);
#pragma unroll
    for (int e = 0; e < kBnPer; ++e) {
        const float y = (v[e] - mean) * invstd * gamma + beta;
        v[e] = a.relu ? fmaxf(y, 0.0f) : y;
    }
    bn_store<VEC>(a.out + s.base, s.len, v);
}

template <bool VEC>
__global__ void __launch_bounds__(kBnThreads) bn_bwd_stats_kernel(const BnArgs a) {
    __shared__ float red[4];
    const BnSlab s = bn_slab(a);
    const int c = blockIdx.y;
    const float mean = a.mean[c], invstd = a.invstd[c], gamma = a.gamma[c], beta = a.beta[c];
    float v[kBnPer], g[kBnPer];
    bn_load<VEC>(a.x + s.base, s.len, v);
    bn_load<VEC>(a.dy + s.base, s.len, g);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int e = 0; e < kBnPer; ++e) {
        const float xhat = (v[e] - mean) * invstd;
        const float y = xhat * gamma + beta;                               // the forward's expression: same sign decisions
        const float ge = (a.relu && !(y > 0.0f)) ? 0.0f : g[e];           // out-of-range elements: dy loaded as zero
        s1 += ge;
        s2 = fmaf(ge, xhat, s2);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* w = a.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        w[0] = s1; w[1] = s2;
    }
}

__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const float* __restrict__ ws, int chunks, double count,
                                                             const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ coef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < chunks; k += 64) {
        s1 += ws[((size_t)c * chunks + k) * 2];
        s2 += ws[((size_t)c * chunks + k) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        coef[3 * c] = gamma[c] * invstd[c];
        coef[3 * c + 1] = (float)(s1 / count);
        coef[3 * c + 2] = (float)(s2 / count);
    }
}

template <bool VEC>
__global__ void __launch_bounds__(kBnThreads) bn_bwd_apply_kernel(const BnArgs a) {
    const BnSlab s = bn_slab(a);
    const int c = blockIdx.y;
    const float mean = a.mean[c], invstd = a.invstd[c], gamma = a.gamma[c], beta = a.beta[c];
    const float scale = a.coef[3 * c], k1 = a.coef[3 * c + 1], k2 = a.coef[3 * c + 2];
    float v[kBnPer], g[kBnPer];
    bn_load<VEC>(a.x + s.base, s.len, v);
    bn_load<VEC>(a.dy + s.base, s.len, g);
#pragma unroll
    for (int e = 0; e < kBnPer; ++e) {
        const float xhat = (v[e] - mean) * invstd;
        const float y = xhat * gamma + beta;
        const float ge = (a.relu && !(y > 0.0f)) ? 0.0f : g[e];
        v[e] = scale * (ge - k1 - xhat * k2);
    }
    bn_store<VEC>(a.out + s.base, s.len, v);
}

static inline int bn_parts(int HW) { return (HW + kBnSlab - 1) / kBnSlab; }

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_bn_workspace_floats(int32_t N, int32_t C, int32_t HW) {
    if (N < 1 || C < 1 || HW < 1) return ITERMVS_ERR_DIMS;
    const int64_t f = (int64_t)C * N * bn_parts(HW) * 3 + (int64_t)C * 3;
    return f > 0x7fffffff ? ITERMVS_ERR_DIMS : (int)f;
}

static int bn_check(const void* x, const void* y, int N, int C, int HW, const void* g, const void* b, const void* m, const void* is,
                    const void* ws) {
    ITERMVS_RETURN_IF(!x || !y || !g || !b || !m || !is || !ws, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(N < 1 || C < 1 || HW < 1 || C > 65535 || (int64_t)N * bn_parts(HW) > 0x7fffffff, ITERMVS_ERR_DIMS);
    return ITERMVS_OK;
}

extern "C" int itermvs_bn_train_forward(const float* x, float* y, int32_t N, int32_t C, int32_t HW, const float* gamma,
                                        const float* beta, float eps, float momentum, int32_t relu, float* running_mean,
                                        float* running_var, float* save_mean, float* save_invstd, float* workspace, void* stream) {
    const int rc = bn_check(x, y, N, C, HW, gamma, beta, save_mean, save_invstd, workspace);
    if (rc) return rc;
    ITERMVS_RETURN_IF((int64_t)N * HW < 2, ITERMVS_ERR_DIMS);               // torch refuses one value per channel in training
    BnArgs a{};
    a.x = x; a.out = y; a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd; a.ws = workspace;
    a.N = N; a.C = C; a.HW = HW; a.parts = bn_parts(HW); a.relu = relu;
    a.vec = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    const int chunks = N * a.parts;
    const dim3 grid(chunks, C);
    hipStream_t st = (hipStream_t)stream;
    if (a.vec) hipLaunchKernelGGL(bn_stats_kernel<true>, grid, dim3(kBnThreads), 0, st, a);
    else hipLaunchKernelGGL(bn_stats_kernel<false>, grid, dim3(kBnThreads), 0, st, a);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, chunks, eps, momentum, save_mean, save_invstd,
                       running_mean, running_var);
    if (a.vec) hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(kBnThreads), 0, st, a);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(kBnThreads), 0, st, a);
    return itermvs_launch_status();
}

extern "C" int itermvs_bn_train_backward(const float* x, const float* dy, float* dx, int32_t N, int32_t C, int32_t HW,
                                         const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                                         int32_t relu, float* dgamma, float* dbeta, float* workspace, void* stream) {
    const int rc = bn_check(x, dx, N, C, HW, gamma, beta, save_mean, save_invstd, workspace);
    if (rc) return rc;
    ITERMVS_RETURN_IF(!dy || !dgamma || !dbeta, ITERMVS_ERR_NULL);
    BnArgs a{};
    a.x = x; a.dy = dy; a.out = dx; a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd;
    a.N = N; a.C = C; a.HW = HW; a.parts = bn_parts(HW); a.relu = relu;
    const int chunks = N * a.parts;
    a.ws = workspace;
    float* coef = workspace + (size_t)C * chunks * 3;
    a.coef = coef;
    a.vec = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0);
    const dim3 grid(chunks, C);
    hipStream_t st = (hipStream_t)stream;
    if (a.vec) hipLaunchKernelGGL(bn_bwd_stats_kernel<true>, grid, dim3(kBnThreads), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_stats_kernel<false>, grid, dim3(kBnThreads), 0, st, a);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, chunks, (double)N * (double)HW, gamma, save_invstd,
                       dgamma, dbeta, coef);
    if (a.vec) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(kBnThreads), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(kBnThreads), 0, st, a);
    return itermvs_launch_status();
}
