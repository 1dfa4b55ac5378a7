// Phase timeline of one persistent workgroup of conv_tile_kernel (wave 0 of workgroup 0).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DITERMVS_TILE_TRACE -I include -I itermvs_amd/csrc \
//        -o tools/ubench/conv_tile_trace tools/ubench/conv_tile_trace.hip
// usage: conv_tile_trace [Cin Cout H W N persist dilation]
#include "../../itermvs_amd/csrc/conv_tile.hip"

#include <stdio.h>
#include <vector>

void itermvs_profile_begin(int, hipStream_t) {}
void itermvs_profile_end(int, hipStream_t) {}

int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16;
    const int H = argc > 3 ? atoi(argv[3]) : 256, W = argc > 4 ? atoi(argv[4]) : 320, N = argc > 5 ? atoi(argv[5]) : 5;
    if (argc > 6) setenv("ITERMVS_TILE_PERSIST", argv[6], 1);
    const int dil = argc > 7 ? atoi(argv[7]) : 1;
    const int S = cin <= 4 ? 1 : cin <= 8 ? 2 : 4, nch = (cin + 4 * S - 1) / (4 * S), coutp = (cout + 15) / 16 * 16;
    float *in, *out, *wt;
    const size_t nin = (size_t)N * cin * H * W, nout = (size_t)N * cout * H * W, nw = (size_t)9 * nch * 4 * coutp * S;
    (void)hipMalloc(&in, nin * 4); (void)hipMalloc(&out, nout * 4); (void)hipMalloc(&wt, nw * 4);
    std::vector<float> h(nin, 0.5f), hw(nw, 0.01f);
    (void)hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(wt, hw.data(), nw * 4, hipMemcpyHostToDevice);
    itermvs_conv_params p = {};
    p.in = in; p.out = out; p.in_sn = (int64_t)cin * H * W; p.out_sn = (int64_t)cout * H * W;
    p.weight[0] = wt; p.n_seg = 1; p.N = N; p.Cin = cin; p.Hin = H; p.Win = W; p.Cout = cout;
    p.ksize = 3; p.stride = 1; p.pad = dil; p.dilation = dil; p.act = 1; p.weight_format = 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) itermvs_conv2d_tile(&p, H, W, 0);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) itermvs_conv2d_tile(&p, H, W, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.1f us/launch\n", ms * 100.0f);
    unsigned long long t[8 * 64];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(itermvs::g_tile_trace), sizeof(t));
    printf("tile: sync+ldswrite | weights+decode+setup | taps (MFMA + prefetch) | epilogue   [cycles of s_memtime]\n");
    for (int i = 0; i < 8; ++i)
        printf("%2d: %6llu %6llu %6llu %6llu   total %6llu\n", i, t[i * 8 + 1] - t[i * 8 + 0], t[i * 8 + 2] - t[i * 8 + 1],
               t[i * 8 + 3] - t[i * 8 + 2], t[i * 8 + 4] - t[i * 8 + 3], t[i * 8 + 4] - t[i * 8 + 0]);
    return 0;
}
