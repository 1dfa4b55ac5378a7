// micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 with NACC independent accumulators per wave,
// optionally with one LDS read per MFMA (operands changing every step).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = 1e-3f * i;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (LDS) { b = sm[idx & 4095]; idx += 64; }
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 4000;
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * blocks * 4;
    printf("%-28s blocks=%5d: %7.2f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1, false>("1 acc/wave", 1024);
    run<2, false>("2 acc/wave", 1024);
    run<4, false>("4 acc/wave", 1024);
    run<4, false>("4 acc/wave, 8 waves/SIMD", 2048);
    run<4, true>("4 acc + 1 ds_read per mfma", 1024);
    run<4, true>("4 acc + ds_read, 8w/SIMD", 2048);
    return 0;
}
