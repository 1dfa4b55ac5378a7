// Camera composition shared by the kernels that carry it as a few extra threads (geometry.hip: compose_proj / ref_quarter_compose;
// stem.hip: stem_compose).
#pragma once
#include "common.hpp"

namespace itermvs {

// ---------------------------------------------------------------------------------------------
// compose_proj: out[set, s-1, 0:12] = rows of (src_s @ inverse(ref))[:3, :4]   (module.py:77-90)
// One thread per (set, source view); 4x4 Gauss-Jordan with partial pivoting in fp64.
// ---------------------------------------------------------------------------------------------
struct ComposeArgs {
    const float* mats;
    float* out;
    int* nan_flag;
    const float* depth_min;
    const float* depth_max;
    float* inv_min;
    float* inv_max;
    int n_sets, V, B;
};

__device__ __forceinline__ void compose_proj_body(const ComposeArgs& c, int t) {
    const float* __restrict__ mats = c.mats;
    float* __restrict__ out = c.out;
    int* __restrict__ nan_flag = c.nan_flag;
    const float* __restrict__ depth_min = c.depth_min;
    const float* __restrict__ depth_max = c.depth_max;
    float* __restrict__ inv_min = c.inv_min;
    float* __restrict__ inv_max = c.inv_max;
    const int n_sets = c.n_sets, V = c.V, B = c.B;
    const int S = V - 1;
    // inverse depth range of the batch (1 / depth_min, 1 / depth_max: itermvs.py:240-241), IEEE division
    if (inv_min && t < B) {
        inv_min[t] = 1.0f / depth_min[t];
        inv_max[t] = 1.0f / depth_max[t];
    }
    if (t >= n_sets * S) return;
    const int set = t / S, s = t - set * S + 1;
    const float* ref = mats + (size_t)set * V * 16;
    const float* src = ref + (size_t)s * 16;
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = (double)ref[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > best) {
                best = fabs(a[r][c]);
                piv = r;
            }
        if (piv != c)
            for (int j = 0; j < 8; ++j) {
                double tmp = a[c][j];
                a[c][j] = a[piv][j];
                a[piv][j] = tmp;
            }
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    bool bad = false;
    float* o = out + (size_t)t * 12;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += (double)src[i * 4 + k] * a[k][4 + j];
            const float v = (float)acc;
            bad |= (v != v);
            o[i * 4 + j] = v;
        }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
}

}  // namespace itermvs
