// Cost of a device-wide barrier between two phases of one persistent kernel on MI355X (all workgroups resident):
// the alternative to a kernel boundary (~4.5 us launch floor + the consumer's start-up) for chaining dependent layers.
// Each phase writes 4 MB, the barrier = agent-scope release fence + atomic counter + spin + acquire fence.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void grid_sync(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this workgroup's stores reach memory
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                   // acquire
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) phases(float* buf, unsigned* counter, int n_phase, int per_block, int check) {
    const int nb = gridDim.x;
    for (int ph = 0; ph < n_phase; ++ph) {
        float* dst = buf + (size_t)(ph & 1) * nb * per_block;
        const float* src = buf + (size_t)((ph + 1) & 1) * nb * per_block;
        // read the NEIGHBOUR workgroup's slice of the previous phase (another CU, usually another XCD), write ours
        const int other = (blockIdx.x + 1) % nb;
        for (int i = threadIdx.x; i < per_block; i += 256) {
            const float v = ph ? src[(size_t)other * per_block + i] : 0.0f;
            dst[(size_t)blockIdx.x * per_block + i] = v + 1.0f;
        }
        grid_sync(counter, (unsigned)(ph + 1) * nb);
    }
    (void)check;
}

int main() {
    const int nb = 1024, per_block = 1024;       // 4 MB per phase
    float* buf; unsigned* ctr;
    (void)hipMalloc(&buf, (size_t)2 * nb * per_block * 4);
    (void)hipMalloc(&ctr, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int n_phase : {1, 11, 51}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(ctr, 0, 4);
            (void)hipMemset(buf, 0, (size_t)2 * nb * per_block * 4);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(phases, dim3(nb), dim3(256), 0, 0, buf, ctr, n_phase, per_block, rep == 4);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        float v = 0;
        (void)hipMemcpy(&v, buf + (size_t)((n_phase - 1) & 1) * nb * per_block, 4, hipMemcpyDeviceToHost);
        printf("%2d phases: %.1f us   (value %g, expect %d)\n", n_phase, best * 1e3f, v, n_phase);
    }
    // the same phases as separate kernel launches
    for (int n_phase : {11, 51}) {
        (void)hipMemset(ctr, 0, 4);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < n_phase; ++i) {
            hipLaunchKernelGGL(phases, dim3(nb), dim3(256), 0, 0, buf, ctr, 1, per_block, 0);
            (void)hipMemsetAsync(ctr, 0, 4, 0);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%2d launches (+memset each): %.1f us\n", n_phase, ms * 1e3f);
    }
    return 0;
}
