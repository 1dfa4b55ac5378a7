// Do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and ordinary fp32 VALU work of ANOTHER wave on the same SIMD overlap?
// One workgroup of 8 waves per CU (2 per SIMD): waves 0-3 run an MFMA chain, waves 4-7 a v_fma chain.
// mode 0: MFMA waves only, 1: VALU waves only, 2: both.  If the two share an execution resource, t(2) ~ t(0)+t(1).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, int valu_per_iter_x4) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (mode == 1) return;
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = threadIdx.x * 0.001f, y = 1.0f + blockIdx.x * 1e-6f;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (mode == 0) return;
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
        const float m = 1.0001f, c = 0.5f;
        for (int i = 0; i < iters; ++i)
            for (int j = 0; j < valu_per_iter_x4; ++j) {
                v0 = __builtin_fmaf(v0, m, c); v1 = __builtin_fmaf(v1, m, c);
                v2 = __builtin_fmaf(v2, m, c); v3 = __builtin_fmaf(v3, m, c);
            }
        out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3;
    }
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int vpi : {1, 2, 4, 8})
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 1000, mode, vpi);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, vpi);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("valu/iter=%2d mode %d (%s): %8.3f ms   [4 MFMA/iter = 128 MFMA-cycles; %d VALU/iter = %d cycles]\n", vpi * 4, mode,
                   mode == 0 ? "mfma only" : mode == 1 ? "valu only" : "both     ", ms, vpi * 4, vpi * 16);
        }
    return 0;
}
