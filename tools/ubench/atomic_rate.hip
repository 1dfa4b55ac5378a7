// Throughput of fp32 no-return global atomics (global_atomic_add_f32) on MI355X by ACCESS SHAPE and by CONTENTION: the
// decision input for the gradient scatter of the fused correlation backward (csrc/corr_bwd.hip), which issues 94-126 M
// lane-atomics per launch.  Every wave-instruction adds 64 floats; what varies is how the 64 lanes are laid out:
//   quad4x4    16 segments of 64 B per instruction, 4 lanes per segment 16 B apart (lane j -> dword 4j + k): the scatter as
//              written in round 2 (a lane owns a float4 of channels, one channel per instruction)
//   quad16     16 segments, 4 lanes per segment CONTIGUOUS (16 B per segment)
//   seg64      4 segments per instruction, 16 contiguous dwords (64 B) each: the quad-transposed scatter
//   line256    one contiguous 256 B run
//   lds+flush  the same adds accumulated in LDS (ds_add_f32), one global atomic per LDS word at the end (window form)
// and where the segments land:
//   spread     anywhere in a 32 MB buffer
//   local      inside an 8 KB neighbourhood owned by the workgroup's position (neighbouring workgroups overlap by half):
//              the collision pattern of neighbouring pixels' footprints
// Prints G lane-atomics/s.  build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// shape: 0 quad4x4, 1 quad16, 2 seg64, 3 line256.  local: 0 spread, 1 neighbourhood
template <int SHAPE, int LOCAL>
__global__ __launch_bounds__(256) void scatter(float* __restrict__ buf, unsigned nfloats, int iters) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned wg_base = LOCAL ? (unsigned)(((unsigned long long)blockIdx.x * 1024u) % (nfloats - 4096u)) : 0u;   // 4 KB apart, 8 KB wide
    for (int it = 0; it < iters; ++it) {
        const unsigned inst = (blockIdx.x * 4u + wave) * 4096u + (unsigned)it;
        unsigned idx;
        if (SHAPE == 0) {
            const unsigned seg = mix(inst * 16u + (lane >> 2));
            idx = (seg % (LOCAL ? 128u : nfloats / 16u)) * 16u + (lane & 3u) * 4u + ((unsigned)it & 3u);
        } else if (SHAPE == 1) {
            const unsigned seg = mix(inst * 16u + (lane >> 2));
            idx = (seg % (LOCAL ? 128u : nfloats / 16u)) * 16u + ((unsigned)it & 3u) * 4u + (lane & 3u);
        } else if (SHAPE == 2) {
            const unsigned seg = mix(inst * 4u + (lane >> 4));
            idx = (seg % (LOCAL ? 128u : nfloats / 16u)) * 16u + (lane & 15u);
        } else {
            const unsigned seg = mix(inst);
            idx = (seg % (LOCAL ? 32u : nfloats / 64u)) * 64u + lane;
        }
        unsafeAtomicAdd(buf + wg_base + idx, 1.0f);
    }
}

// the window form: the adds of one workgroup go to a 2048-float LDS window (8 KB), flushed with one global atomic per word
template <int SHAPE>
__global__ __launch_bounds__(256) void scatter_lds(float* __restrict__ buf, unsigned nfloats, int iters) {
    __shared__ float win[2048];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned wg_base = (unsigned)(((unsigned long long)blockIdx.x * 1024u) % (nfloats - 4096u));
    for (int i = threadIdx.x; i < 2048; i += 256) win[i] = 0.0f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const unsigned inst = (blockIdx.x * 4u + wave) * 4096u + (unsigned)it;
        unsigned idx;
        if (SHAPE == 0) {
            const unsigned seg = mix(inst * 16u + (lane >> 2));
            idx = (seg % 128u) * 16u + (lane & 3u) * 4u + ((unsigned)it & 3u);
        } else {
            const unsigned seg = mix(inst * 4u + (lane >> 4));
            idx = (seg % 128u) * 16u + (lane & 15u);
        }
        __hip_atomic_fetch_add(&win[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) {
        const float v = win[i];
        if (v != 0.0f) unsafeAtomicAdd(buf + wg_base + i, v);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const unsigned nfloats = 8u << 20;   // 32 MB
    float* buf;
    CK(hipMalloc(&buf, (size_t)nfloats * 4));
    CK(hipMemset(buf, 0, (size_t)nfloats * 4));
    const int blocks = 5120, iters = 256;          // 5120 x 4 waves x 256 instr x 64 lanes = 335 M lane-atomics
    const double lanes = (double)blocks * 4 * iters * 64;
    const char* shape[4] = {"quad4x4", "quad16", "seg64", "line256"};
#define RUN(S, L) { double ms = time_ms([&] { hipLaunchKernelGGL((scatter<S, L>), dim3(blocks), dim3(256), 0, 0, buf, nfloats, iters); }); \
        printf("%-9s %-6s %8.3f ms  %7.1f G lane-atomics/s\n", shape[S], L ? "local" : "spread", ms, lanes / ms * 1e-6); }
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0)
    RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(3, 1)
    for (int it2 : {16, 64, 256}) {
        const double l2 = (double)blocks * 4 * it2 * 64;
        double ms = time_ms([&] { hipLaunchKernelGGL((scatter_lds<0>), dim3(blocks), dim3(256), 0, 0, buf, nfloats, it2); });
        printf("lds+flush quad4x4 %3d adds/word  %8.3f ms  %7.1f G lane-adds/s\n", it2 * 256 / 2048, ms, l2 / ms * 1e-6);
        ms = time_ms([&] { hipLaunchKernelGGL((scatter_lds<2>), dim3(blocks), dim3(256), 0, 0, buf, nfloats, it2); });
        printf("lds+flush seg64   %3d adds/word  %8.3f ms  %7.1f G lane-adds/s\n", it2 * 256 / 2048, ms, l2 / ms * 1e-6);
    }
    // sanity: every add arrived
    float* host = (float*)malloc((size_t)nfloats * 4);
    CK(hipMemcpy(host, buf, (size_t)nfloats * 4, hipMemcpyDeviceToHost));
    double sum = 0;
    for (unsigned i = 0; i < nfloats; ++i) sum += host[i];
    printf("sum of the buffer %.0f\n", sum);
    return 0;
}
