// How fast can a kernel READ planar activations tile by tile?  Tensor [80 planes][256][320] fp32 (26 MB, the 16-channel
// half-resolution maps of 5 views).  Patterns: streaming float4; (8+2) x (32+2) tile x 16 planes per workgroup with dword
// loads (what conv_tile / stem stage); the same interior rows as aligned float4 loads; 8 x 64 tiles; 16 x 32 tiles.
// Every variant sums what it reads (one store per thread block keeps the loads alive).
// build: hipcc --offload-arch=gfx950 -O3 -o tile_read tile_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
constexpr int NPL = 80, H = 256, W = 320, PLANE = H * W;

__global__ void stream4(const float4* __restrict__ x, float* out, int n4) {
    float s = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) { float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[blockIdx.x] = s;
}

// tile TH x TW (+1 halo each side), 16 planes per workgroup, dword loads, element e -> (plane, row, col) row-major
template <int TH, int TW>
__global__ void tile_dword(const float* __restrict__ x, float* out, int tiles_x, int tiles_y) {
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int g = t / tiles_y;      // g = group of 16 planes
    const float* xp = x + (size_t)g * 16 * PLANE;
    constexpr int R = TH + 2, C = TW + 2, N = 16 * R * C;
    float s = 0.f;
    for (int e = threadIdx.x; e < N; e += 256) {
        const int c = e / (R * C), rem = e - c * (R * C), r = rem / C, col = rem - r * C;
        const int gy = min(max(ty * TH - 1 + r, 0), H - 1), gx = min(max(tx * TW - 1 + col, 0), W - 1);
        s += xp[c * PLANE + gy * W + gx];
    }
    if (s == 123.456f) out[blockIdx.x] = s;
}

// interior TH+2 rows x TW columns as aligned float4 loads (the two halo columns are skipped: 6 % of the bytes)
template <int TH, int TW>
__global__ void tile_vec4(const float* __restrict__ x, float* out, int tiles_x, int tiles_y) {
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int g = t / tiles_y;
    const float* xp = x + (size_t)g * 16 * PLANE;
    constexpr int R = TH + 2, V = TW / 4, N = 16 * R * V;
    float s = 0.f;
    for (int e = threadIdx.x; e < N; e += 256) {
        const int c = e / (R * V), rem = e - c * (R * V), r = rem / V, v = rem - r * V;
        const int gy = min(max(ty * TH - 1 + r, 0), H - 1);
        const float4 q = *reinterpret_cast<const float4*>(xp + c * PLANE + gy * W + tx * TW + v * 4);
        s += q.x + q.y + q.z + q.w;
    }
    if (s == 123.456f) out[blockIdx.x] = s;
}

template <class F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.1f us  %6.2f TB/s\n", name, ms * 1e3 / 50, bytes / (ms * 1e-3 / 50) / 1e12);
}

int main() {
    float *x, *out;
    CK(hipMalloc(&x, (size_t)NPL * PLANE * 4));
    CK(hipMemset(x, 0, (size_t)NPL * PLANE * 4));
    CK(hipMalloc(&out, 1 << 20));
    const double all = (double)NPL * PLANE * 4;
    timeit("streaming float4", all, [&] { hipLaunchKernelGGL(stream4, dim3(2048), dim3(256), 0, 0, (const float4*)x, out, NPL * PLANE / 4); });
    timeit("tile 8x32 dword (halo 1.33x)", all * (10 * 34) / (8 * 32), [&] { hipLaunchKernelGGL((tile_dword<8, 32>), dim3(10 * 32 * 5), dim3(256), 0, 0, x, out, 10, 32); });
    timeit("tile 4x32 dword (halo 1.59x)", all * (6 * 34) / (4 * 32), [&] { hipLaunchKernelGGL((tile_dword<4, 32>), dim3(10 * 64 * 5), dim3(256), 0, 0, x, out, 10, 64); });
    timeit("tile 8x64 dword (halo 1.29x)", all * (10 * 66) / (8 * 64), [&] { hipLaunchKernelGGL((tile_dword<8, 64>), dim3(5 * 32 * 5), dim3(256), 0, 0, x, out, 5, 32); });
    timeit("tile 16x32 dword (halo 1.20x)", all * (18 * 34) / (16 * 32), [&] { hipLaunchKernelGGL((tile_dword<16, 32>), dim3(10 * 16 * 5), dim3(256), 0, 0, x, out, 10, 16); });
    timeit("tile 8x32 float4 interior", all * 10 / 8, [&] { hipLaunchKernelGGL((tile_vec4<8, 32>), dim3(10 * 32 * 5), dim3(256), 0, 0, x, out, 10, 32); });
    timeit("tile 8x64 float4 interior", all * 10 / 8, [&] { hipLaunchKernelGGL((tile_vec4<8, 64>), dim3(5 * 32 * 5), dim3(256), 0, 0, x, out, 5, 32); });
    return 0;
}
