// Can a kernel INSIDE a captured hipGraph be timed with HIP events?  (bench.py brackets itermvs_corr_iter with
// events; today that launch stays outside the graph segments for exactly this reason.)
// build: hipcc --offload-arch=gfx950 -O3 -o graph_event_timing graph_event_timing.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

__global__ void spin(float* p, int n) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}

int main() {
    float* d;
    CK(hipMalloc(&d, 4096));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 1000);
    CK(hipEventRecordWithFlags(e0, s, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000);
    CK(hipEventRecordWithFlags(e1, s, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 1000);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float ms = -1.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("replay %d: middle kernel %.3f ms by external events\n", rep, ms);
    }
    // reference: the same kernel timed eagerly
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = -1.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager: %.3f ms\n", ms);
    return 0;
}
