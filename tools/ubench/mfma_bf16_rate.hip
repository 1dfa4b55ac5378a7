// micro-benchmark for the bf16x3 convolution design (gfx950): issue rate of the bf16 MFMA shapes, with the operand traffic
// a split-fp32 implicit GEMM needs (ds_read_b128 per MFMA) and with vector work of the same wave / other waves beside it.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_rate mfma_bf16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using s4 = __attribute__((ext_vector_type(4))) short;

// MODE 0: 16x16x32 bf16   1: legacy 16x16x16 bf16 (_1k)   2: 32x32x16 bf16
// RD: ds_read_b128 per MFMA * 2 (0, 1 = one read per two MFMAs, 2 = one per MFMA, 4 = two per MFMA)
// VA: v_and/v_sub pairs per MFMA issued by the SAME wave
template <int MODE, int NACC, int RD, int VA>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = 1e-3f * i;
    __syncthreads();
    f32x4 acc[NACC];
    f32x16 acc32[MODE == 2 ? NACC : 1];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    if (MODE == 2)
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    f32x4 a = {threadIdx.x * 1e-3f, 1.f, 2.f, 3.f}, b = {1.0f + threadIdx.x * 1e-4f, 2.f, 3.f, 4.f};
    float v0 = threadIdx.x * 0.37f, v1 = 1.0f;
    int idx = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (RD == 4 || RD == 2 || (RD == 1 && (i & 1) == 0)) { b = *reinterpret_cast<const f32x4*>(sm + (idx & 8188)); idx += 256; }
            if (RD == 4) { a = *reinterpret_cast<const f32x4*>(sm + ((idx + 1024) & 8188)); }
#pragma unroll
            for (int v = 0; v < VA; ++v) {
                const float h = __uint_as_float(__float_as_uint(v0) & 0xffff0000u);
                v1 = v0 - h + v1;
                v0 = v1 * 1.0001f;
            }
            if constexpr (MODE == 0)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[i], 0, 0, 0);
            else if constexpr (MODE == 1)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s4{(short)__float_as_uint(a[0]), 1, 2, 3}, s4{(short)__float_as_uint(b[0]), 2, 3, 4}, acc[i], 0, 0, 0);
            else
                acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc32[i], 0, 0, 0);
        }
    }
    float s = v0 + v1;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (MODE == 2)
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) s += acc32[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC, int RD, int VA>
void run(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE, NACC, RD, VA>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC, RD, VA>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = MODE == 0 ? 2.0 * 16 * 16 * 32 : MODE == 1 ? 2.0 * 16 * 16 * 16 : 2.0 * 32 * 32 * 16;
    const double n_mfma = (double)NACC * iters * blocks * 4;
    // cycles per MFMA per SIMD at 2.4 GHz, waves per SIMD = blocks * 4 / 1024
    const double wps = blocks * 4 / 1024.0;
    printf("%-58s blocks=%5d: %7.2f ms  %7.1f TFLOP/s  %6.1f cyc/MFMA/SIMD (2.4 GHz)\n", name, blocks, ms, per * n_mfma / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps));
    hipFree(out);
}

int main() {
    run<0, 1, 0, 0>("16x16x32 bf16, 1 acc (dependent chain)", 1024);
    run<0, 4, 0, 0>("16x16x32 bf16, 4 acc", 1024);
    run<0, 4, 0, 0>("16x16x32 bf16, 4 acc, 2 waves/SIMD", 2048);
    run<1, 4, 0, 0>("legacy 16x16x16 bf16_1k, 4 acc", 1024);
    run<2, 2, 0, 0>("32x32x16 bf16, 2 acc", 1024);
    run<0, 4, 1, 0>("16x16x32, 4 acc, 1 ds_read_b128 per 2 MFMA", 1024);
    run<0, 4, 2, 0>("16x16x32, 4 acc, 1 ds_read_b128 per MFMA", 1024);
    run<0, 4, 2, 0>("16x16x32, 4 acc, 1 ds_read_b128 per MFMA, 2 waves/SIMD", 2048);
    run<0, 4, 2, 0>("16x16x32, 4 acc, 1 ds_read_b128 per MFMA, 4 waves/SIMD", 4096);
    run<0, 4, 4, 0>("16x16x32, 4 acc, 2 ds_read_b128 per MFMA, 4 waves/SIMD", 4096);
    run<0, 4, 0, 2>("16x16x32, 4 acc, 6 VALU per MFMA same wave", 1024);
    run<0, 4, 0, 2>("16x16x32, 4 acc, 6 VALU per MFMA, 2 waves/SIMD", 2048);
    run<0, 4, 0, 2>("16x16x32, 4 acc, 6 VALU per MFMA, 4 waves/SIMD", 4096);
    run<0, 4, 0, 4>("16x16x32, 4 acc, 12 VALU per MFMA, 4 waves/SIMD", 4096);
    return 0;
}
