// itermvs_head_regress: the tail of the depth head fused with the softmax regression.
//   x [B,32,P] = relu(conv3x3(hidden))              models/itermvs.py:121-126 (depth_head[0:2], run by itermvs_conv2d)
//   y = relu(W1 x)            W1 [64,32]  (1x1)     depth_head[2:4]
//   logits = W2 y + b2        W2 [256,64] (1x1)     depth_head[4]
//   nd = window regression of softmax(logits)       models/itermvs.py:171-190 / 201-219 (== itermvs_prob_regress)
// Unfused this is three launches and a 21 MB logits tensor written and read back per GRU iteration.
//
// One wave owns 16 pixels and both GEMMs on v_mfma_f32_16x16x4_f32 (D: lane l holds channels
// mb*16 + (l>>4)*4 + r of pixel l&15):
//   * GEMM 1 (K = 32): input channel of (step group u, k-slot q, step s) is u*16 + q*4 + s, so a lane's four
//     operands of a group are one ds_read_b128 of the packed W1 and four plane loads of x;
//   * GEMM 2 (K = 64) is chained IN REGISTERS: its k-step (mb1, r) takes hidden channel mb1*16 + q*4 + r --
//     exactly accumulator acc1[mb1][r] of the lane that needs it as B operand -- so y never leaves the VGPRs;
//   * each lane ends with 64 of its pixel's 256 logits; max / sum / first-arg-max are combined over the four
//     lanes of a pixel with two xor-shuffles; the nine window bins go through LDS and lane q = 0 evaluates
//     the regression with the arithmetic of prob_regress_kernel (update.hip).
// W1 (8 KB) and W2 (64 KB) sit in LDS, shared by the four waves of a workgroup (64 pixels).
#include "common.hpp"

namespace itermvs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kHeadBins = ITERMVS_PROB_BINS;
constexpr int kHeadWin = 2 * ITERMVS_WINDOW_RADIUS + 1;
constexpr int kW1Floats = 4 * 2 * 4 * 16 * 4;      // [mb1][u][q][i][s]
constexpr int kW2Floats = 16 * 4 * 4 * 16 * 4;     // [mb2][mb1][q][i][r]
constexpr int kHeadLds = (kW1Floats + kW2Floats + 4 * kHeadWin * 16) * 4;

struct HeadArgs {
    const float* x;
    int64_t x_sb;
    const float* w1p;
    const float* w2p;
    const float* bias2;
    float* nd0;
    float* nd1;
    int64_t nd_sb0, nd_sb1;
    int64_t* best;
    int P;
};

__global__ void __launch_bounds__(256) head_regress_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ w1 = smem;
    float* __restrict__ w2 = smem + kW1Floats;
    float* __restrict__ win = smem + kW1Floats + kW2Floats;      // [wave][9][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l16 = lane & 15;
    const int b = blockIdx.y;
    const int p = (blockIdx.x * 4 + wave) * 16 + l16;
    const bool live = p < a.P;

    // x operands first (plane loads, out of range -> 0), then the weight copy: both are in flight together
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (int64_t)b * a.x_sb), 0, (int)(32u * (uint32_t)a.P * 4u), 0x00020000);
    const uint32_t xoff = live ? ((uint32_t)(q * 4) * (uint32_t)a.P + (uint32_t)p) * 4u : 0x7fffffffu;
    float xv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            xv[u][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xoff, (uint32_t)(u * 16 + s) * (uint32_t)a.P * 4u, 0));
    {
        const f32x4* __restrict__ s1 = reinterpret_cast<const f32x4*>(a.w1p);
        const f32x4* __restrict__ s2 = reinterpret_cast<const f32x4*>(a.w2p);
        f32x4 t[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) t[i] = s1[tid + i * 256];
#pragma unroll
        for (int i = 0; i < 2; ++i) reinterpret_cast<f32x4*>(w1)[tid + i * 256] = t[i];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = s2[tid + (h * 8 + i) * 256];
#pragma unroll
            for (int i = 0; i < 8; ++i) reinterpret_cast<f32x4*>(w2)[tid + (h * 8 + i) * 256] = t[i];
        }
    }
    __syncthreads();

    // GEMM 1: y[64] = relu(W1 x)
    f32x4 acc1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc1[mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(w1 + (((mb * 2 + u) * 4 + q) * 16 + l16) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], xv[u][s], acc1[mb], 0, 0, 0);
        }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[mb][r] = fmaxf(acc1[mb][r], 0.0f);

    // GEMM 2: logits[256] = W2 y + b2; accumulators start from the bias
    f32x4 acc2[16];
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
        const f32x4 bs = *reinterpret_cast<const f32x4*>(a.bias2 + mb * 16 + q * 4);
        acc2[mb] = bs;
    }
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(w2 + (((mb * 4 + m1) * 4 + q) * 16 + l16) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], acc1[m1][r], acc2[mb], 0, 0, 0);
        }

    // softmax statistics over the pixel's 256 bins: own 64 (bin = mb*16 + q*4 + r), then the four q-lanes
    float m = acc2[0][0];
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc2[mb][r]);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float s = 0.0f, bv = -1.0f;
    int bi = 0;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = expf(acc2[mb][r] - m);
            acc2[mb][r] = e;
            s += e;
            if (e > bv) {          // strict: the lowest own bin wins ties (bins are visited in increasing order)
                bv = e;
                bi = mb * 16 + q * 4 + r;
            }
        }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    // First arg-max of p = e / s (torch.argmax first-max rule on the probabilities, itermvs.py:176).
    // Dividing by the common s is monotonic, so the arg-max of e decides -- unless another bin lies within
    // 2^-22 of the maximum, where the quotients may round to the same float: then (rare, wave-uniform
    // branch) the quotients themselves are compared, exactly like prob_regress_kernel.
    float gm = fmaxf(bv, __shfl_xor(bv, 16));
    gm = fmaxf(gm, __shfl_xor(gm, 32));
    const float thresh = gm * 0.99999976f;
    int close = 0;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) close += acc2[mb][r] >= thresh ? 1 : 0;
    close += __shfl_xor(close, 16);
    close += __shfl_xor(close, 32);
    float bp = bv;
    if (__any(close > 1)) {
        bp = -1.0f;
#pragma unroll
        for (int mb = 0; mb < 16; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pk = acc2[mb][r] / s;
                if (pk > bp) {
                    bp = pk;
                    bi = mb * 16 + q * 4 + r;
                }
            }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float op = __shfl_xor(bp, sh);
        const int oi = __shfl_xor(bi, sh);
        if (op > bp || (op == bp && oi < bi)) {
            bp = op;
            bi = oi;
        }
    }
    // window k*-4 .. k*+4 (unclamped positions) -> LDS, as e; lane q == 0 divides and regresses
    const int lo = bi - ITERMVS_WINDOW_RADIUS;
    float* __restrict__ wv = win + wave * (kHeadWin * 16);
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = mb * 16 + q * 4 + r - lo;
            if (off >= 0 && off < kHeadWin) wv[off * 16 + l16] = acc2[mb][r];
        }
    __syncthreads();
    if (q == 0 && live) {
        float num = 0.0f, den = 1e-6f;   // itermvs.py:212
        for (int i = 0; i < kHeadWin; ++i) {
            int k = lo + i;
            k = k < 0 ? 0 : (k > kHeadBins - 1 ? kHeadBins - 1 : k);   // clamp; duplicates double-counted
            const float pk = wv[(k - lo) * 16 + l16] / s;
            num = num + (float)k * pk;
            den = den + pk;
        }
        const float nd = (num / den) / (float)(kHeadBins - 1);
        if (a.nd0) a.nd0[b * a.nd_sb0 + p] = nd;
        if (a.nd1) a.nd1[b * a.nd_sb1 + p] = nd;
        if (a.best) a.best[(size_t)b * a.P + p] = bi;
    }
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_head_regress(const float* x, int64_t x_sb, int32_t B, int32_t P, const float* w1_packed,
                                    const float* w2_packed, const float* bias2, float* nd_out0, int64_t nd_sb0,
                                    float* nd_out1, int64_t nd_sb1, int64_t* best, void* stream) {
    ITERMVS_RETURN_IF(!x || !w1_packed || !w2_packed || !bias2, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || P < 1, ITERMVS_ERR_DIMS);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(head_regress_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLds) == hipSuccess;
    ITERMVS_RETURN_IF(!attr_ok, ITERMVS_ERR_LAUNCH);
    HeadArgs a;
    a.x = x; a.x_sb = x_sb; a.w1p = w1_packed; a.w2p = w2_packed; a.bias2 = bias2;
    a.nd0 = nd_out0; a.nd1 = nd_out1; a.nd_sb0 = nd_sb0; a.nd_sb1 = nd_sb1; a.best = best; a.P = P;
    hipLaunchKernelGGL(head_regress_kernel, dim3((P + 63) / 64, B), dim3(256), kHeadLds, (hipStream_t)stream, a);
    return itermvs_launch_status();
}
