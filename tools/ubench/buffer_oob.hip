// Does the gfx950 raw-buffer bounds check include the SGPR offset?  (conv kernels rely on OOB -> 0.)
// build: hipcc --offload-arch=gfx950 -O3 -o buffer_oob buffer_oob.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const float* base, int nbytes, uint32_t voff, uint32_t soff, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nbytes, 0x00020000);
    const uint32_t s = __builtin_amdgcn_readfirstlane(soff);
    out[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + threadIdx.x * 4, s, 0));
}

int main() {
    float* d; float* o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 64 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 1000.0f + i;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    struct { uint32_t v, s; const char* what; } cases[] = {
        {0, 0, "in range"},
        {1024, 0, "voffset past range (range 1024 B)"},
        {0, 1024, "soffset past range"},
        {512, 768, "voffset+soffset past range, each in range"},
        {960, 0, "straddles end via lane offset"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024, c.v, c.s, o);
        float r[64];
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("%-45s lane0=%g lane15=%g lane16=%g lane63=%g\n", c.what, r[0], r[15], r[16], r[63]);
    }
    return 0;
}
