// Camera composition, the plain (un-fused) warp seam and reference-feature resampling.
#include "corr_common.hpp"
#include "compose.hpp"

namespace itermvs {

__global__ void compose_proj_kernel(ComposeArgs c) { compose_proj_body(c, blockIdx.x * blockDim.x + threadIdx.x); }

// ---------------------------------------------------------------------------------------------
// warp: thread per (b, n, y, x); loops over channels.  out[B,C,N,H,W]      (module.py:68-125)
// ---------------------------------------------------------------------------------------------
__global__ void warp_kernel(itermvs_fmap src, const float* __restrict__ proj, const float* __restrict__ depth,
                            int B, int N, int H, int W, float* __restrict__ out, uint8_t* __restrict__ mask) {
    const int64_t total = (int64_t)B * N * H * W;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % W);
    const int y = (int)((t / W) % H);
    const int n = (int)((t / ((int64_t)W * H)) % N);
    const int b = (int)(t / ((int64_t)W * H * N));
    const WarpGeom g = make_geom(W, H, src.W, src.H);
    const float* m = proj + (size_t)b * 12;
    float rx, ry, rz, ix, iy;
    ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
    bool valid;
    project(g, m, rx, ry, rz, depth[t], ix, iy, &valid);
    if (mask) mask[t] = valid ? 1 : 0;
    const Taps tp = make_taps(ix, iy, src.W, src.H);
    const float* base = (const float*)src.data + (int64_t)b * src.sb;
    const int64_t o00 = tp.y0 * src.sy + tp.x0 * src.sx, o01 = tp.y0 * src.sy + tp.x1 * src.sx;
    const int64_t o10 = tp.y1 * src.sy + tp.x0 * src.sx, o11 = tp.y1 * src.sy + tp.x1 * src.sx;
    const int64_t plane = (int64_t)N * H * W;
    float* o = out + ((int64_t)b * src.C * N + n) * H * W + (int64_t)y * W + x;
    for (int c = 0; c < src.C; ++c) {
        const float* f = base + c * src.sc;
        const float v = fmaf(tp.se, f[o11], fmaf(tp.sw, f[o10], fmaf(tp.ne, f[o01], tp.nw * f[o00])));
        o[c * plane] = v;
    }
}

// gradient w.r.t. the source map: scatter-add of the four taps (fp32 atomics)
__global__ void warp_backward_kernel(const float* __restrict__ gout, const float* __restrict__ proj,
                                     const float* __restrict__ depth, int B, int C, int N, int H, int W, int H1,
                                     int W1, float* __restrict__ gsrc) {
    const int64_t total = (int64_t)B * N * H * W;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % W);
    const int y = (int)((t / W) % H);
    const int n = (int)((t / ((int64_t)W * H)) % N);
    const int b = (int)(t / ((int64_t)W * H * N));
    const WarpGeom g = make_geom(W, H, W1, H1);
    const float* m = proj + (size_t)b * 12;
    float rx, ry, rz, ix, iy;
    ray_dir(m, (float)x * g.xr, (float)y * g.yr, rx, ry, rz);
    project(g, m, rx, ry, rz, depth[t], ix, iy, nullptr);
    const Taps tp = make_taps(ix, iy, W1, H1);
    const int64_t plane = (int64_t)N * H * W;
    const float* go = gout + ((int64_t)b * C * N + n) * H * W + (int64_t)y * W + x;
    float* gs = gsrc + (int64_t)b * C * H1 * W1;
    for (int c = 0; c < C; ++c) {
        const float gv = go[c * plane];
        float* f = gs + (int64_t)c * H1 * W1;
        if (tp.nw != 0.0f) atomicAdd(f + tp.y0 * W1 + tp.x0, gv * tp.nw);
        if (tp.ne != 0.0f) atomicAdd(f + tp.y0 * W1 + tp.x1, gv * tp.ne);
        if (tp.sw != 0.0f) atomicAdd(f + tp.y1 * W1 + tp.x0, gv * tp.sw);
        if (tp.se != 0.0f) atomicAdd(f + tp.y1 * W1 + tp.x1, gv * tp.se);
    }
}

// ---------------------------------------------------------------------------------------------
// ref_quarter: out[b, y, x, 0:C1 | C1:C1+C2 | C1+C2:] on the level-2 grid   (itermvs.py:95-98)
// thread per (b, y, x, channel quad); writes float4 (C1, C2, C3 are multiples of 4).
// ---------------------------------------------------------------------------------------------
template <int FT>
__device__ __forceinline__ float ld(const itermvs_fmap& f, int b, int c, int y, int x) {
    return ld_feat<FT>((const float*)f.data, b * f.sb + c * f.sc + y * f.sy + x * f.sx);
}

// F.interpolate(scale_factor=2, bilinear, align_corners=False) source index / weight
__device__ __forceinline__ void up2_axis(int d, int n_in, int& i0, int& i1, float& l0, float& l1) {
    float s = ((float)d + 0.5f) * 0.5f - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

// four consecutive channels of one position (16 bytes in fp32, 8 in 16-bit storage when the map is channels-last)
template <int FT>
__device__ __forceinline__ void ld4(const itermvs_fmap& f, int b, int c, int y, int x, float (&v)[4]) {
    const int64_t off = b * f.sb + c * f.sc + y * f.sy + x * f.sx;
    if (f.sc == 1) {
        load_feat<4, FT>((const float*)f.data, (uint32_t)off, v);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld_feat<FT>((const float*)f.data, off + k * f.sc);
    }
}

// grid = (x chunks, B * H rows, 3 parts [+ 1]): blockIdx.z selects the pyramid level a block resamples, so a wave runs ONE of the
// three forms (the first version took a thread per (pixel, channel quad) of all 96 channels: every wave executed the x0.5, the
// copy and the x2 branch one after the other and spent most of its instructions on three runtime integer divisions -- 11.7 us
// for 17 MB at cfg 1); the row and the batch item come from blockIdx.y, the only division left is by the part's constant number
// of channel quads.  Same arithmetic per element as before (bit-identical results).
// z == 3 (itermvs_ref_quarter_compose only): compose_proj, an independent piece of work of a few threads.
template <int FT>
__global__ void __launch_bounds__(256) ref_quarter_kernel(itermvs_fmap r1, itermvs_fmap r2, itermvs_fmap r3, int B, float* __restrict__ out,
                                                          ComposeArgs comp) {
    const int part = blockIdx.z;
    if (part == 3) {
        if (blockIdx.y == 0) compose_proj_body(comp, (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x);
        return;
    }
    const int H = r2.H, W = r2.W;
    const int CQ = r1.C + r2.C + r3.C;
    const int b = (int)blockIdx.y / H, y = (int)blockIdx.y - b * H;
    const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
    float v[4];
    int x, c, coff;
    if (part == 0) {
        const int quads = r1.C >> 2;
        x = t / quads; c = (t - x * quads) * 4; coff = 0;
        if (x >= W) return;
        // x0.5 bilinear == weights 0.5/0.5 on rows 2y,2y+1 and columns 2x,2x+1
        float a00[4], a01[4], a10[4], a11[4];
        ld4<FT>(r1, b, c, 2 * y, 2 * x, a00); ld4<FT>(r1, b, c, 2 * y, 2 * x + 1, a01);
        ld4<FT>(r1, b, c, 2 * y + 1, 2 * x, a10); ld4<FT>(r1, b, c, 2 * y + 1, 2 * x + 1, a11);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float top = a00[k] * 0.5f + a01[k] * 0.5f;
            const float bot = a10[k] * 0.5f + a11[k] * 0.5f;
            v[k] = top * 0.5f + bot * 0.5f;
        }
    } else if (part == 1) {
        const int quads = r2.C >> 2;
        x = t / quads; c = (t - x * quads) * 4; coff = r1.C;
        if (x >= W) return;
        ld4<FT>(r2, b, c, y, x, v);
    } else {
        const int quads = r3.C >> 2;
        x = t / quads; c = (t - x * quads) * 4; coff = r1.C + r2.C;
        if (x >= W) return;
        int y0, y1, x0, x1;
        float hy0, hy1, hx0, hx1;
        up2_axis(y, r3.H, y0, y1, hy0, hy1);
        up2_axis(x, r3.W, x0, x1, hx0, hx1);
        float a00[4], a01[4], a10[4], a11[4];
        ld4<FT>(r3, b, c, y0, x0, a00); ld4<FT>(r3, b, c, y0, x1, a01);
        ld4<FT>(r3, b, c, y1, x0, a10); ld4<FT>(r3, b, c, y1, x1, a11);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float top = a00[k] * hx0 + a01[k] * hx1;
            const float bot = a10[k] * hx0 + a11[k] * hx1;
            v[k] = top * hy0 + bot * hy1;
        }
    }
    *reinterpret_cast<float4*>(out + ((size_t)((size_t)b * H + y) * W + x) * CQ + coff + c) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace itermvs

using namespace itermvs;

extern "C" int itermvs_compose_proj(const float* mats, int32_t n_sets, int32_t V, float* out, int32_t* nan_flag,
                                    const float* depth_min, const float* depth_max, int32_t B, float* inv_min,
                                    float* inv_max, void* stream) {
    ITERMVS_RETURN_IF(!mats || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(inv_min && (!depth_min || !depth_max || !inv_max || B < 1), ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(n_sets < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(V < 2 || V - 1 > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    int total = n_sets * (V - 1);
    if (inv_min && B > total) total = B;
    const ComposeArgs c{mats, out, nan_flag, depth_min, depth_max, inv_min, inv_max, n_sets, V, B};
    hipLaunchKernelGGL(compose_proj_kernel, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, c);
    return itermvs_launch_status();
}

extern "C" int itermvs_warp(const itermvs_fmap* src, const float* proj, const float* depth, int32_t B, int32_t N,
                            int32_t H, int32_t W, float* out, uint8_t* mask, void* stream) {
    ITERMVS_RETURN_IF(src && src->dtype != ITERMVS_F32, ITERMVS_ERR_DTYPE);
    ITERMVS_RETURN_IF(!src || !src->data || !proj || !depth || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || N < 1 || H < 1 || W < 1 || src->C < 1 || src->H < 1 || src->W < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * N * H * W;
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *src, proj,
                       depth, B, N, H, W, out, mask);
    return itermvs_launch_status();
}

extern "C" int itermvs_warp_backward(const float* grad_out, const float* proj, const float* depth, int32_t B, int32_t C,
                                     int32_t N, int32_t H, int32_t W, int32_t H1, int32_t W1, float* grad_src,
                                     void* stream) {
    ITERMVS_RETURN_IF(!grad_out || !proj || !depth || !grad_src, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || C < 1 || N < 1 || H < 1 || W < 1 || H1 < 1 || W1 < 1, ITERMVS_ERR_DIMS);
    const int64_t total = (int64_t)B * N * H * W;
    hipLaunchKernelGGL(warp_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, proj, depth, B, C, N, H, W, H1, W1, grad_src);
    return itermvs_launch_status();
}

static int launch_ref_quarter(const itermvs_fmap* r1, const itermvs_fmap* r2, const itermvs_fmap* r3, int32_t B, float* out,
                              const ComposeArgs* comp, void* stream) {
    ITERMVS_RETURN_IF(!r1 || !r2 || !r3 || !r1->data || !r2->data || !r3->data || !out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(B < 1 || r2->H < 1 || r2->W < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(r1->H != 2 * r2->H || r1->W != 2 * r2->W || r2->H != 2 * r3->H || r2->W != 2 * r3->W,
                      ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF((r1->C % 4) || (r2->C % 4) || (r3->C % 4), ITERMVS_ERR_CHANNELS);
    ITERMVS_RETURN_IF(((uintptr_t)out) % 16, ITERMVS_ERR_ALIGN);
    // channels-last maps (sc == 1) take the vector path of ld4: four channels per 16-byte (fp32) / 8-byte (16-bit) load at a
    // 32-bit element offset -> base pointer, pixel / row / batch strides and extent must allow it (other layouts: scalar loads)
    for (const itermvs_fmap* f : {r1, r2, r3}) {
        if (f->sc != 1) continue;
        const uintptr_t need = f->dtype == ITERMVS_F32 ? 16 : 8;
        ITERMVS_RETURN_IF(((uintptr_t)f->data) % need || (f->sx % 4) || (f->sy % 4) || (f->sb % 4), ITERMVS_ERR_ALIGN);
        ITERMVS_RETURN_IF((int64_t)(B - 1) * f->sb + (int64_t)(f->H - 1) * f->sy + (int64_t)(f->W - 1) * f->sx + f->C > 0x7fffffffLL,
                          ITERMVS_ERR_DIMS);
    }
    const int64_t total = (int64_t)B * r2->H * r2->W * ((r1->C + r2->C + r3->C) / 4);
    ITERMVS_RETURN_IF(total >= ((int64_t)1 << 31) - 512, ITERMVS_ERR_DIMS);        // 32-bit offsets in the kernel
    ITERMVS_RETURN_IF((int64_t)B * r2->H > 65535, ITERMVS_ERR_DIMS);               // grid.y = B * H rows
    ITERMVS_RETURN_IF(r1->dtype != r2->dtype || r1->dtype != r3->dtype, ITERMVS_ERR_DTYPE);
    int cmax = r1->C > r2->C ? r1->C : r2->C;
    cmax = cmax > r3->C ? cmax : r3->C;
    int gx = (r2->W * (cmax / 4) + 255) / 256;
    ComposeArgs c{};
    int parts = 3;
    if (comp) {
        c = *comp;
        int ct = c.n_sets * (c.V - 1);
        if (c.inv_min && c.B > ct) ct = c.B;
        const int need = (ct + 255) / 256;
        gx = gx > need ? gx : need;
        parts = 4;
    }
    const dim3 grid((unsigned)gx, (unsigned)(B * r2->H), (unsigned)parts);
    switch (r1->dtype) {
        case ITERMVS_F32: hipLaunchKernelGGL(ref_quarter_kernel<ITERMVS_F32>, grid, dim3(256), 0, (hipStream_t)stream, *r1, *r2, *r3, B, out, c); break;
        case ITERMVS_F16: hipLaunchKernelGGL(ref_quarter_kernel<ITERMVS_F16>, grid, dim3(256), 0, (hipStream_t)stream, *r1, *r2, *r3, B, out, c); break;
        case ITERMVS_BF16: hipLaunchKernelGGL(ref_quarter_kernel<ITERMVS_BF16>, grid, dim3(256), 0, (hipStream_t)stream, *r1, *r2, *r3, B, out, c); break;
        default: return ITERMVS_ERR_DTYPE;
    }
    return itermvs_launch_status();
}

extern "C" int itermvs_ref_quarter(const itermvs_fmap* r1, const itermvs_fmap* r2, const itermvs_fmap* r3, int32_t B,
                                   float* out, void* stream) {
    return launch_ref_quarter(r1, r2, r3, B, out, nullptr, stream);
}

extern "C" int itermvs_ref_quarter_compose(const itermvs_fmap* r1, const itermvs_fmap* r2, const itermvs_fmap* r3, int32_t B,
                                           float* out, const float* mats, int32_t n_sets, int32_t V, float* proj_out,
                                           int32_t* nan_flag, const float* depth_min, const float* depth_max, int32_t Bd,
                                           float* inv_min, float* inv_max, void* stream) {
    ITERMVS_RETURN_IF(!mats || !proj_out, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(inv_min && (!depth_min || !depth_max || !inv_max || Bd < 1), ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(n_sets < 1, ITERMVS_ERR_DIMS);
    ITERMVS_RETURN_IF(V < 2 || V - 1 > ITERMVS_MAX_SRC, ITERMVS_ERR_VIEWS);
    const ComposeArgs c{mats, proj_out, nan_flag, depth_min, depth_max, inv_min, inv_max, n_sets, V, Bd};
    return launch_ref_quarter(r1, r2, r3, B, out, &c, stream);
}
