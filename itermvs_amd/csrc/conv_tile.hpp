// Argument block shared by the LDS-tiled 3x3 convolution kernels: conv_tile.hip (exact fp32 MFMA, weight_format 2) and
// conv_tile3.hip (bf16x3 split on the bf16 MFMA, weight_format 3).
#pragma once

#include "common.hpp"

namespace itermvs {

struct TileArgs {
    const float* in;
    float* out;
    float* out2;
    const float* add;
    const float* aux1;
    const float* aux2;
    int64_t in_sn, out_sn, add_sn, aux1_sn, aux2_sn;
    const float* weight[3];   // packed [9][nchunk][4][CoutPad][S] (format 2) / [9][nchunk][3][CoutPad][16] bf16 (format 3)
    const float* bias[3];
    int seg_end[3];
    int N, Cin, Hin, Win, Cout, CoutPad, Hout, Wout;
    float* out_b;             // second result (channels >= split) or nullptr
    int64_t out_b_sn;
    int split, act_b;
    int pad, act, add_mode, out_nhwc, nchunk, nstage, tiles_x, tiles_y, ncb, total;
    uint32_t rcp_tiles_x, rcp_tiles_y;   // floor(2^32 / d) + 1
    int banded;                          // XCD-banded tile order (see the kernels)
};

constexpr uint32_t kTileOob = 0x7fffffffu;

}  // namespace itermvs
