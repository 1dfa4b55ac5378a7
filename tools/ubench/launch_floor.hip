// What does a kernel boundary cost?  N dependent launches of a small kernel (a) from a host loop on a stream, (b) replayed
// from a captured hipGraph; for kernels of ~0, ~5 and ~20 us.  Wall time per launch = kernel time + boundary cost.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

__global__ void work(float* p, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[t];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[t] = v;
}

int main() {
    const int N = 60, REPS = 50;
    float* d;
    CK(hipMalloc(&d, 1024 * 256 * 4));
    CK(hipMemset(d, 0, 1024 * 256 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters[3] = {1, 2500, 10000};
    for (int k = 0; k < 3; ++k) {
        // one launch alone
        hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, d, iters[k]);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, d, iters[k]);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float single = 0;
        CK(hipEventElapsedTime(&single, e0, e1));
        // (a) host loop
        float eager = 0;
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r)
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, d, iters[k]);
        CK(hipEventRecord(e1, s));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&eager, e0, e1));
        const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (REPS * N);
        // (b) graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, d, iters[k]);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float graph = 0;
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&graph, e0, e1));
        printf("kernel of %d iterations: alone %.1f us (event pair) | host loop %.2f us per launch (host enqueue %.2f us) | graph %.2f us per launch\n",
               iters[k], single * 1e3, eager * 1e3 / (REPS * N), host_us, graph * 1e3 / (REPS * N));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
