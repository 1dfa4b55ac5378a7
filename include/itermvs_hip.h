/*
 * itermvs_hip.h -- C ABI of libitermvs_hip.so (hand-written gfx950 HIP kernels for the
 * IterMVS matching hot path).
 *
 * The reference (FangjinhuaWang/IterMVS) has NO native / FFI interface: its hot path is
 * Python calling stock PyTorch ops.  The entry points below are therefore the seams of the
 * reference's Python functions, lowered to plain pointers + sizes; each one cites the
 * reference code it replaces (paths relative to the reference repository).  The ctypes
 * binding a maintainer of the reference would add is shown in INTEGRATION.md and lives in
 * itermvs_amd/_lib.py.
 *
 * Conventions:
 *   - every function ENQUEUES work on `stream` and returns immediately: no allocation, no
 *     synchronisation, no global mutable state (re-entrant across streams / threads);
 *   - all tensors are caller-owned DEVICE buffers of fp32 unless stated, borrowed for the
 *     duration of the enqueue; inputs are never written;
 *   - return value 0 = ok, negative = itermvs_status (bad dims, null pointer, unsupported
 *     channel count, misalignment); nothing is thrown;
 *   - `void* stream` is a hipStream_t (NULL = default stream).
 */
#ifndef ITERMVS_HIP_H
#define ITERMVS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the prototypes of this header are its ONLY dynamic symbols
 * (tests/test_host_cpu.py holds `nm -D` to that). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define ITERMVS_ABI_VERSION 16
#define ITERMVS_MAX_SRC 16     /* source views per reference view (pair.txt holds 10) */
#define ITERMVS_MAX_HYP 8      /* hypotheses per level in the iteration branch (4,4,2) */
#define ITERMVS_GROUPS 8       /* models/itermvs.py:28  */
#define ITERMVS_PROB_BINS 256  /* models/itermvs.py:134 */
#define ITERMVS_WINDOW_RADIUS 4 /* models/itermvs.py:135 */

typedef enum itermvs_status {
    ITERMVS_OK = 0,
    ITERMVS_ERR_NULL = -1,        /* required pointer is NULL                     */
    ITERMVS_ERR_DIMS = -2,        /* non-positive / inconsistent dimension        */
    ITERMVS_ERR_CHANNELS = -3,    /* C not in {16,32,48} (C/G must be 2,4,6)      */
    ITERMVS_ERR_VIEWS = -4,       /* S < 1 or S > ITERMVS_MAX_SRC                 */
    ITERMVS_ERR_ALIGN = -5,       /* pointer / stride not aligned for vector path */
    ITERMVS_ERR_LAYOUT = -6,      /* fused kernels need channels-last features    */
    ITERMVS_ERR_LAUNCH = -7,      /* hipLaunchKernel failed (see hipGetLastError) */
    ITERMVS_ERR_DTYPE = -8        /* feature storage type not supported by this entry point */
} itermvs_status;

/* Storage type of FEATURE maps (the pyramids the correlation kernels gather from).  Arithmetic is always fp32: 16-bit
 * storage halves the gathered bytes (BASELINE cfg 4 "bf16", cfg 5 "fp16"); coordinates, depths, view weights, correlations
 * and every other tensor stay fp32. */
typedef enum itermvs_dtype {
    ITERMVS_F32 = 0,
    ITERMVS_F16 = 1,              /* IEEE binary16 */
    ITERMVS_BF16 = 2              /* bfloat16      */
} itermvs_dtype;

/* library / ABI identification */
int itermvs_version(void);
const char* itermvs_error_string(int status);

/* A 4-D feature map addressed with ELEMENT strides, so NCHW and channels-last (NHWC) views
 * of torch tensors are both accepted.  The fused kernels require sc == 1 (channels-last). */
typedef struct itermvs_fmap {
    const void* data;         /* elements of `dtype` */
    int64_t sb, sc, sy, sx;   /* strides of batch, channel, row, column (elements) */
    int32_t C, H, W;
    int32_t dtype;            /* itermvs_dtype; 16-bit storage is accepted by the fused correlation entry points and
                                 itermvs_ref_quarter, everything else takes ITERMVS_F32 */
} itermvs_fmap;

/* The S source-view feature maps of one pyramid level: per-view base pointers sharing one
 * set of element strides (views of a [B,V,...] tensor or separately allocated maps). */
typedef struct itermvs_level_src {
    const void* view[ITERMVS_MAX_SRC];   /* elements of `dtype` */
    int64_t sb, sc, sy, sx;
    int32_t C, H, W;
    int32_t dtype;                       /* itermvs_dtype (all levels of one call share it) */
} itermvs_level_src;

/* ------------------------------------------------------------------------------------------
 * itermvs_compose_proj -- models/module.py:77-90
 *   proj = src_proj @ inverse(ref_proj); rot = proj[:3,:3]; trans = proj[:3,3]
 * `mats` is [n_sets, V, 4, 4] row-major (view 0 = reference); `out` is [n_sets, V-1, 12]
 * holding rows of [rot | trans].  Inverse and product are evaluated in fp64 and rounded
 * once to fp32.  `nan_flag` (device int32, may be NULL) is OR-ed with 1 when a result is
 * NaN -- the deferred form of the reference's host-side asserts (module.py:83,87).
 * When `inv_min` != NULL the same launch also writes the inverse depth range of the batch,
 * inv_min[b] = 1 / depth_min[b], inv_max[b] = 1 / depth_max[b] (models/itermvs.py:240-241), b < B.
 * ------------------------------------------------------------------------------------------ */
int itermvs_compose_proj(const float* mats, int32_t n_sets, int32_t V, float* out,
                         int32_t* nan_flag, const float* depth_min, const float* depth_max,
                         int32_t B, float* inv_min, float* inv_max, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_warp -- models/module.py:68-125  differentiable_warping(src_fea, src_proj,
 *   ref_proj, depth_samples, return_mask)
 * Plain (un-fused) seam: writes the warped volume out[B,C,N,H,W] (contiguous) and, when
 * `mask` != NULL, the validity mask [B,N,H,W] as uint8.  `proj` = [B,12] from
 * itermvs_compose_proj.  Any strides / any C.  Kept for API completeness and unit tests;
 * the engine itself never materialises the warped volume.
 * ------------------------------------------------------------------------------------------ */
int itermvs_warp(const itermvs_fmap* src, const float* proj, const float* depth,
                 int32_t B, int32_t N, int32_t H, int32_t W, float* out, uint8_t* mask,
                 void* stream);

/* gradient of itermvs_warp w.r.t. src (grid math is no_grad in the reference, module.py:77):
 * grad_src[B,C,H1,W1] (contiguous NCHW, must be zero-filled) += scatter of grad_out[B,C,N,H,W] */
int itermvs_warp_backward(const float* grad_out, const float* proj, const float* depth,
                          int32_t B, int32_t C, int32_t N, int32_t H, int32_t W,
                          int32_t H1, int32_t W1, float* grad_src, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_ref_quarter -- models/itermvs.py:95-98
 * Reference-view features of the three pyramid levels resampled onto the 1/4 grid and packed
 * channels-last: out[B,H,W,C1+C2+C3] (level 1: x0.5 bilinear == 2x2 mean; level 2: copy;
 * level 3: x2 bilinear, align_corners=False).  H,W = size of level 2.
 * Any element strides; a channels-last map (sc == 1) is read four channels per load and must then have its data
 * pointer 16-byte (fp32) / 8-byte (16-bit) aligned and sx, sy, sb multiples of 4 (ITERMVS_ERR_ALIGN otherwise).
 * ------------------------------------------------------------------------------------------ */
int itermvs_ref_quarter(const itermvs_fmap* ref_l1, const itermvs_fmap* ref_l2,
                        const itermvs_fmap* ref_l3, int32_t B, float* out, void* stream);
/* itermvs_ref_quarter_compose -- itermvs_ref_quarter and, in the SAME launch (a few extra threads), itermvs_compose_proj with
 * its arguments (mats .. inv_max; Bd = batch size of the depth range): the two are independent and both precede the
 * correlation kernels. */
int itermvs_ref_quarter_compose(const itermvs_fmap* r1, const itermvs_fmap* r2, const itermvs_fmap* r3, int32_t B, float* out,
                                const float* mats, int32_t n_sets, int32_t V, float* proj_out, int32_t* nan_flag,
                                const float* depth_min, const float* depth_max, int32_t Bd, float* inv_min, float* inv_max,
                                void* stream);

/* itermvs_copy_multi -- n (<= 8) device-to-device copies of bytes[i] bytes in ONE launch: the staging of a sample (images +
 * cameras + depth range) into the static input buffers a captured hipGraph reads (engine.GraphedRunner).  src / dst / bytes
 * are HOST arrays. */
int itermvs_copy_multi(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);

/* itermvs_box_probe -- a fixed micro-kernel for bench.py's box calibration (no reference counterpart: the reference prints
 * wall-clock only, eval.py:130-137): `blocks` workgroups of 4 waves issue `iters` x 4 independent v_mfma_f32_16x16x4_f32 per
 * wave (2*16*16*4 FLOP each) and store one float per thread to sink[blocks*256].  clocks (device, 2 x uint64, may be NULL):
 * workgroup 0 writes its shader-clock ticks (s_memtime) and its constant-rate 100 MHz ticks (s_memrealtime) across the loop
 * -> the sustained shader clock of THIS chip under matrix load.  The companion bandwidth probe is itermvs_copy_multi on a
 * 256 MB buffer.  Timed by the caller with events on `stream`. */
int itermvs_box_probe(float* sink, int32_t blocks, int32_t iters, uint64_t* clocks, void* stream);
/* itermvs_box_chase -- the latency probe of the same calibration: ONE lane follows i -> ring[i] from `start` for `steps` dependent
 * loads (ring: device uint32 indices forming one cycle, built by the caller; its size selects L2 / memory-side cache / HBM);
 * out[0] = the final index (start of the next call: untouched lines), clocks[0] = elapsed 100 MHz ticks (s_memrealtime),
 * clocks[1] = elapsed shader-clock ticks (the clock of a nearly idle chip).  clocks: 2 x uint64. */
int itermvs_box_chase(const uint32_t* ring, uint32_t start, int32_t steps, uint32_t* out, uint64_t* clocks, void* stream);
/* itermvs_clock_stamp -- out (device, uint64[16][2], zeroed by the caller): every XCD writes { its 100 MHz counter, its shader-clock
 * counter } to slot XCC_ID.  Two stamps around other work on the stream: per XCD, delta(shader) / delta(100 MHz) x 100 = the clock
 * in MHz the chip sustained under THAT work (bench.py `box.sclk_workload_MHz`). */
int itermvs_clock_stamp(uint64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_corr_iter -- models/itermvs.py:84-120 (Evaluation.forward, iteration branch, up to
 * but excluding CorrNet) fused with module.py:68-125 and, optionally, the hypothesis
 * construction of itermvs.py:290-293.
 * For every 1/4-res pixel, level l in {1,2,3}, hypothesis n, group g:
 *   out_l[b,n,g,y,x] = sum_s w[b,s,y,x] * corr_s / (1e-5 + sum_s w[b,s,y,x]),
 *   corr_s = mean_{c in group g} warp(src_l[s])[c] * ref_q[c]
 * No warped volume / cost volume is written to memory.
 * ------------------------------------------------------------------------------------------ */
typedef struct itermvs_corr_iter_params {
    int32_t B, S, H, W;                        /* sample grid = level-2 size              */
    int32_t N[3];                              /* hypotheses per level (reference: 4,4,2) */
    int32_t impl;                              /* reserved, must be 0 (one kernel form)   */
    itermvs_level_src src[3];                  /* [level-1]; channels-last (sc == 1)      */
    const float* ref_q;                        /* [B,H,W,C1+C2+C3] from itermvs_ref_quarter */
    const float* proj;                         /* [3,B,S,12] from itermvs_compose_proj    */
    const float* view_w;                       /* view weights, element (b,s,y,x) at view_w[b*view_w_sb + s*view_w_ss + (y*W+x)*view_w_sp] */
    int64_t view_w_sb, view_w_ss, view_w_sp;   /* all 0 = contiguous [B,S,H,W]; the engine stores them INTERLEAVED [B,H,W,S]
                                                * (ss = 1, sp = S): the S weights of a pixel are one vector load of a lane
                                                * quad.  itermvs_corr_iter_backward needs the contiguous form. */
    const float* depth[3];                     /* explicit hypotheses [B,N_l,H,W] or NULL */
    const float* norm_depth;                   /* normalised depth at norm_depth[b*norm_depth_sb + y*W + x]; used where depth[l] == NULL */
    int64_t norm_depth_sb;                     /* batch stride (elements) of norm_depth   */
    float offsets[3][ITERMVS_MAX_HYP];         /* normalised offsets corr_interval*interval_scale (itermvs.py:229-235) */
    const float* inv_depth_min;                /* device [B]  1/depth_min                 */
    const float* inv_depth_max;                /* device [B]  1/depth_max                 */
    float* out[3];                             /* [B,N_l,8,H,W] contiguous (CorrNet input [B*N,8,H,W]) */
} itermvs_corr_iter_params;

int itermvs_corr_iter(const itermvs_corr_iter_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_corr_init -- models/itermvs.py:48-51 (+ :11-19 DepthInitialization when depth==NULL)
 * Per-view group correlation at level 3 for all N hypotheses:
 *   out[b,s,n,g,y,x]  (== PixelViewWeight input [B*S*N, 8, H, W])
 * ------------------------------------------------------------------------------------------ */
typedef struct itermvs_corr_init_params {
    int32_t B, S, H, W, N;                     /* sample grid = level-3 size; N = 32      */
    int32_t out_layout;                        /* 0 = out [B,S,N,8,H,W]; 1 = groups last [B,S,N,H,W,8] (out 16-byte aligned): the layout
                                                * itermvs_conv2d (in_layout 1) stages with two 16-byte loads per pixel.  The backward takes 0 */
    itermvs_level_src src;                     /* level-3 source features, channels-last  */
    itermvs_fmap ref;                          /* level-3 reference features (any strides) */
    const float* proj;                         /* [B,S,12]                                */
    const float* depth;                        /* [B,N,H,W] or NULL (generate uniform inverse depth) */
    const float* inv_depth_min;                /* device [B] */
    const float* inv_depth_max;                /* device [B] */
    float* out;                                /* [B,S,N,8,H,W] (see out_layout) */
} itermvs_corr_init_params;

int itermvs_corr_init(const itermvs_corr_init_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_tap_indices -- diagnostic: the bilinear tap indices the FUSED kernels use (models/module.py:99-115 followed by
 * grid_sample's un-normalisation and floor, ATen GridSampler.h:31,205-207).  itermvs_corr_iter / itermvs_corr_init and
 * their gradients never store the sampling position; this entry evaluates it with the same device functions in the same
 * order (hypothesis construction, ray, projection with the library's division form, floor, bounds) and writes, for every
 * (b, view s, hypothesis n, pixel p = y*W + x) of the sample grid:
 *   out[b,s,n,0,p] = floor(ix)   out[b,s,n,1,p] = floor(iy)     as int32 (NaN -> INT32_MIN, saturated at +-2^30)
 *   out[b,s,n,2,p] = bit 0: column x0 inside the source map, bit 1: x0+1, bit 2: row y0, bit 3: y0+1
 *   coords[b,s,n,0,p] = ix, coords[b,s,n,1,p] = iy  (optional, may be NULL)
 * Hypotheses: `depth` [B,N,H,W] when given; else, with init != 0, the N initial planes (models/itermvs.py:13-17); else
 * norm_depth + offsets[n] clamped and un-normalised (models/itermvs.py:291-293, N <= ITERMVS_MAX_HYP).
 * tests/test_tap_indices_gpu.py holds it bit for bit to the reference's own sampling grids (tests/golden/tap_cases.npz).
 * ------------------------------------------------------------------------------------------ */
typedef struct itermvs_tap_params {
    int32_t B, S, H, W;                        /* sample grid                              */
    int32_t N;                                 /* hypotheses                               */
    int32_t H1, W1;                            /* source map size                          */
    int32_t init;                              /* != 0: generate the initial planes        */
    const float* proj;                         /* [B,S,12]                                 */
    const float* depth;                        /* [B,N,H,W] or NULL                        */
    const float* norm_depth;                   /* element (b,y,x) at norm_depth[b*norm_depth_sb + y*W + x] */
    int64_t norm_depth_sb;
    float offsets[ITERMVS_MAX_HYP];
    const float* inv_depth_min;                /* device [B] */
    const float* inv_depth_max;                /* device [B] */
    int32_t* out;                              /* [B,S,N,3,H,W] */
    float* coords;                             /* [B,S,N,2,H,W] or NULL */
} itermvs_tap_params;

int itermvs_tap_indices(const itermvs_tap_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gradients of the two fused correlation entry points for training (train.py:194-243; the training branches of
 * models/itermvs.py:59-61, 111-113).  The sampling grid carries no gradient (models/module.py:77 builds it under
 * torch.no_grad), the view weights of the iteration branch are detached (models/itermvs.py:295); so the gradient flows to
 * the source features (scatter-add of the four taps, fp32 hardware atomics) and to the reference features (gather; the
 * warped values are recomputed, no [B,C,N,H,W] volume is ever stored).
 *
 * itermvs_corr_iter_backward: `p` = the forward's parameter block (its `out` pointers are ignored);
 *   grad_out[l]   [B,N_l,8,H,W]  dL/d out_l
 *   grad_src[l]   host array of S device pointers: dL/d src_l[s], addressed with the strides of p->src[l]
 *                 (channels-last), ZERO-FILLED by the caller, accumulated into;
 *   grad_ref_q    [B,H,W,C1+C2+C3] dL/d ref_q, written completely.
 * itermvs_corr_init_backward: `p` = the forward's parameter block (p->ref must be channels-last here, N <= 32);
 *   grad_out      [B,S,N,8,H,W]  dL/d out;   grad_src: S device pointers as above (zero-filled);
 *   grad_ref      dL/d ref, addressed with the strides of p->ref: ZERO-FILLED by the caller, accumulated into (several
 *                 workgroups per pixel: one per view and chunk of 8 hypotheses).
 * ------------------------------------------------------------------------------------------ */
int itermvs_corr_iter_backward(const itermvs_corr_iter_params* p, const float* const grad_out[3],
                               float* const* const grad_src[3], float* grad_ref_q, void* stream);
int itermvs_corr_init_backward(const itermvs_corr_init_params* p, const float* grad_out, float* const* grad_src,
                               float* grad_ref, void* stream);

/* itermvs_view_aggregate -- models/itermvs.py:59-69
 *   out[b,n,g,p] = sum_s corr[b,s,n,g,p] * w[b,s,p] / (1e-5 + sum_s w[b,s,p])
 * corr [B,S,N,8,P], w [B,S,P] (PixelViewWeight output at 1/8 res), out [B,N,8,P]. */
int itermvs_view_aggregate(const float* corr, const float* w, int32_t S, int32_t B, int32_t N,
                           int32_t P, float* out, void* stream);
/* itermvs_view_aggregate_up -- the same, and in the SAME launch (extra blocks: two independent pieces of work that only
 * read w) the x2 bilinear up-sampling of the view weights the iterations use (models/itermvs.py:56-57,71):
 *   w [B,S,H3,W3] -> w_up [B,S,2*H3,2*W3], or, with w_up_interleaved != 0, the same values stored [B,2*H3,2*W3,S] (the
 *   layout itermvs_corr_iter reads fastest: view_w_ss = 1, view_w_sp = S).
 *   corr_layout: 0 = corr [B,S,N,8,P]; 1 = groups last [B,S,N,P,8] (itermvs_corr_init's out_layout 1; 16-byte aligned). */
int itermvs_view_aggregate_up(const float* corr, int32_t corr_layout, const float* w, int32_t S, int32_t B, int32_t N, int32_t H3,
                              int32_t W3, float* out, float* w_up, int32_t w_up_interleaved, void* stream);

/* itermvs_softmax_max -- models/itermvs.py:347-348 (PixelViewWeight tail)
 *   out[m,p] = max_n softmax_n(x[m,n,p]);  x [M,N,P] contiguous, out [M,P]. */
int itermvs_softmax_max(const float* x, int32_t M, int32_t N, int32_t P, float* out, void* stream);

/* itermvs_pvw_tail -- models/itermvs.py:343-348 (PixelViewWeight after its 3x3 layer), fused:
 *   out[m,p] = max_n softmax_n( sum_c w[c] * x[m*N+n, c, p] + bias )
 * x [M*N, C, P] planes (C = 16, N <= 32), w [C], bias [1] or NULL, out [M,P]. */
int itermvs_pvw_tail(const float* x, const float* w, const float* bias, int32_t M, int32_t N, int32_t C,
                     int32_t P, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_prob_regress -- models/itermvs.py:171-190 and :201-219
 *   p = softmax(logits over 256 bins); k* = first argmax p; window k*-4..k*+4 clamped to
 *   [0,255] (duplicates kept); nd = (sum k p_k / (1e-6 + sum p_k)) / 255
 * logits [B,256,H,W] addressed with element strides (sb, sc, sp) where p = y*W+x must be
 * linear (sp = stride between neighbouring pixels).  Outputs (each may be NULL):
 *   nd_out0 / nd_out1: two destinations for the normalised depth, written at
 *     nd_outX[b*nd_sbX + p] (lets the GRU input buffers be filled in place);
 *   prob [B,256,P] contiguous (training / API parity);  best [B,P] int64 arg-max index.
 * ------------------------------------------------------------------------------------------ */
int itermvs_prob_regress(const float* logits, int64_t sb, int64_t sc, int64_t sp, int32_t B,
                         int32_t P, float* nd_out0, int64_t nd_sb0, float* nd_out1,
                         int64_t nd_sb1, float* prob, int64_t* best, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_head_regress -- the two 1x1 layers of the depth head fused with itermvs_prob_regress
 * (models/itermvs.py:121-126 depth_head[2:5], then :171-190 / :201-219):
 *   y = relu(W1 x); logits = W2 y + b2; nd = window regression of softmax(logits)
 * x [B,32,P] planes (batch stride x_sb) is the output of the head's first (3x3, dilated) layer.
 * Packed weights (fp32):  w1_packed [4][2][4][16][4] with element (mb,u,q,i,s) = W1[mb*16+i][u*16+q*4+s],
 *                         w2_packed [16][4][4][16][4] with element (mb,m,q,i,r) = W2[mb*16+i][m*16+q*4+r];
 * bias2 [256].  Outputs as itermvs_prob_regress (nd_out0 / nd_out1 / best may be NULL); the 256-bin
 * logits are never written to memory.
 * ------------------------------------------------------------------------------------------ */
int itermvs_head_regress(const float* x, int64_t x_sb, int32_t B, int32_t P, const float* w1_packed,
                         const float* w2_packed, const float* bias2, float* nd_out0, int64_t nd_sb0,
                         float* nd_out1, int64_t nd_sb1, int64_t* best, void* stream);

/* itermvs_head_fused -- the whole depth head in one launch: its first layer (3x3, dilation 2, 32 -> 32, ReLU; weights in
 * itermvs_conv2d's weight_format 2 = [9][2][4][32][4]) evaluated from an LDS tile of `hidden` [B,32,H,W] (planes, batch
 * stride hidden_sb) and chained in registers into itermvs_head_regress (models/itermvs.py:121-126, 171-190, 201-219).
 * w2_format: 0 = w2_packed is itermvs_head_regress's fp32 operand layout (exact fp32 matrix instruction);
 *            3 = the 64 -> 256 layer in the bf16x3 form of itermvs_conv2d's weight_format 3 (both operands split exactly into three
 *                bf16 terms, the six largest cross products, fp32 accumulation): w2_packed = bf16 [16][2][3][64][8] whose element
 *                (ob, g, p, lane = 16 q + i, j) is term p of W2[ob*16 + i][(2g + j/4)*16 + 4q + j%4]
 *                (itermvs_amd.ops.pack_head_w2_split3).  w2_packed 16-byte aligned in both forms.
 * w0_format: 0 = w0_tile is the fp32 tile format above; 3 (only together with w2_format 3) = the dilated 3x3 layer in bf16x3 as well:
 *                w0_tile = bf16 [block 2][tap 9][term 3][lane 64][8] whose element (mb, tap, p, 16 q + i, j) is term p of
 *                W0[16 mb + i][(j/4)*16 + 4 q + j%4][tap] (itermvs_amd.ops.pack_head_w0_split3); 16-byte aligned.
 *                itermvs_head_fused_conf takes the fp32 3x3 weights only. */
int itermvs_head_fused(const float* hidden, int64_t hidden_sb, int32_t B, int32_t H, int32_t W,
                       const void* w0_tile, int32_t w0_format, const float* w1_packed, const void* w2_packed, int32_t w2_format,
                       const float* bias2, float* nd_out0, int64_t nd_sb0, float* nd_out1, int64_t nd_sb1, int64_t* best, void* stream);
/* itermvs_head_fused_conf -- itermvs_head_fused and, in the SAME launch on the same staged tile of `hidden`, the confidence
 * head (models/itermvs.py:147-151 with the sigmoid of :198, run on the last GRU iteration :197-199): wc_tile = its dilated
 * 3x3 layer 32 -> 32 in weight_format 2 ([9][2][4][32][4], 16-byte aligned), conf_dot = the 32 weights of its 1x1 layer + bias,
 * conf [B,1,H,W] planes at batch stride conf_sb receives sigmoid(conv1x1(relu(conv3x3(hidden)))). */
int itermvs_head_fused_conf(const float* hidden, int64_t hidden_sb, int32_t B, int32_t H, int32_t W,
                            const float* w0_tile, const float* w1_packed, const void* w2_packed, int32_t w2_format, const float* bias2,
                            float* nd_out0, int64_t nd_sb0, float* nd_out1, int64_t nd_sb1, int64_t* best,
                            const float* wc_tile, const float* conf_dot, float* conf, int64_t conf_sb, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_conv3x3_conv1x1 -- conv3x3 (pad 1) 32 -> 64 + ReLU followed by conv1x1 64 -> NO (+ bias1) in one launch, the
 * 64-channel tensor never stored: IterMVS.upsample (models/itermvs.py:243-247, applied at :262-263; NO = 144, no bias) and
 * Update.hidden_init_head (models/itermvs.py:153-157, applied at :159-160; NO = 32, bias).
 *   x [B,32,H,W] planes (batch stride x_sb); w0_tile = the 3x3 weight in itermvs_conv2d's weight_format 2 = [9][2][4][64][4];
 *   w1_packed [NOB][4][4][16][4] (NOB = ceil(NO/16), zero padded) with element (ob,m,q,i,r) = W1[ob*16+i][m*16+q*4+r];
 *   bias1 [NOB*16] or NULL; out [B,NO,H,W] planes (batch stride out_sb).  All weight pointers 16-byte aligned; NO <= 192.
 *   weight_format 0: the fp32 layouts above (exact fp32 matrix instruction).  weight_format 3: both layers in the bf16x3 form of
 *   itermvs_conv2d's weight_format 3 (operands split exactly into three bf16 terms, six cross products, fp32 accumulation):
 *   w0_tile = bf16 [block 4][tap 9][term 3][lane 64][8] with element (w, tap, p, 16 q + i, j) = term p of
 *   W0[16 w + i][(j/4)*16 + 4 q + j%4][tap]; w1_packed = bf16 [NOB][k group 2][term 3][lane 64][8] with element (ob, g, p, 16 q + i, j) =
 *   term p of W1[ob*16 + i][(2g + j/4)*16 + 4 q + j%4] (itermvs_amd.ops.pack_conv3x3_conv1x1_split3).
 * ------------------------------------------------------------------------------------------ */
int itermvs_conv3x3_conv1x1(const float* x, int64_t x_sb, int32_t B, int32_t H, int32_t W, const void* w0_tile,
                            const void* w1_packed, int32_t weight_format, const float* bias1, int32_t NO, float* out, int64_t out_sb,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_res_chain16 -- the three 3x3 16 -> 16 convolutions that follow the stem in FeatureNet's half-resolution stage
 * (models/net.py:13,40 `layer1` = two ResidualBlocks, models/module.py:33-50; BatchNorm folded) in ONE launch:
 *     a = relu(conv(y1; W0) + b0 + shortcut)     layer1[0].conv2 + layer1[0].downsample's result
 *     b = relu(conv(a;  W1) + b1)                layer1[1].conv1
 *     out = relu(conv(b; W2) + b2 + a)           layer1[1].conv2 + skip
 * a and b never leave LDS.  Arithmetic: the bf16x3 form of itermvs_conv2d's weight_format 3 (both operands split exactly
 * into three bf16 terms, six cross products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
 *   y1, shortcut [N,16,H,W] (image strides y1_sn / shortcut_sn, elements): itermvs_stem's two results, in_layout = the
 *   out_layout they were written with (0 planes, 1 channel quads [N,4,H,W,4]: a quarter of the load instructions);
 *   weights[3]: each layer's weight in weight_format 3 = bf16 [9][1][3][16][16] (itermvs_amd.ops.MfmaWeight(...).tile3), 16-byte
 *   aligned; bias[3]: 16 floats each (NULL = none); out [N,16,H,W] planes.  H, W <= 4095.
 * ------------------------------------------------------------------------------------------ */
int itermvs_res_chain16(const float* y1, int64_t y1_sn, const float* shortcut, int64_t shortcut_sn, int32_t in_layout, int32_t N,
                        int32_t H, int32_t W, const void* const* weights, const float* const* bias, float* out, int64_t out_sn, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_lateral_conv3x3 -- one level of FeatureNet's top-down path in ONE launch (models/net.py:48-50; test mode :62-63):
 *     intra = F.interpolate(coarse, scale_factor=2, mode="bilinear") + inner(fine)       1x1 layer Cf -> 48, bias
 *     out   = output(intra)                                                               3x3 layer 48 -> Cout, bias, padding 1
 * `intra` ([N,48,H,W]; 78 MB at level 1 of cfg 1) never reaches memory: only the output convolution reads it (net.py:50).
 * Arithmetic: the 1x1 layer on the exact fp32 matrix instruction, the up-sampling operation for operation that of
 * itermvs_bilinear_up (align_corners=False), the 3x3 layer in the bf16x3 form of itermvs_conv2d's weight_format 3.
 *   fine [N,Cf,H,W] planes (image stride fine_sn), coarse [N,48,H/2,W/2] planes (coarse_sn); H, W even; Cf = 16, Cout = 16
 *   (ITERMVS_ERR_CHANNELS otherwise: the level-1 layer pair of the path);
 *   w_lat: the 1x1 weight in weight_format 1 = fp32 [Cf][48] (itermvs_amd.ops.MfmaWeight(...).data), b_lat [48] or NULL;
 *   w_out: the 3x3 weight in weight_format 3 = bf16 [9][3][3][16][16] (MfmaWeight(...).tile3), 16-byte aligned; b_out [Cout] or NULL;
 *   out: out_layout 0 = fp32 planes [N,Cout,H,W]; 1 / 2 / 3 = channels-last [N,H,W,Cout] in fp32 / fp16 / bf16 storage (what the
 *   correlation kernels gather from; out_sn in ELEMENTS of the storage type); out2: optional dense fp32 planar copy or NULL.
 * ------------------------------------------------------------------------------------------ */
int itermvs_lateral_conv3x3(const float* fine, int64_t fine_sn, int32_t Cf, const float* coarse, int64_t coarse_sn, int32_t N,
                            int32_t H, int32_t W, const float* w_lat, const float* b_lat, const void* w_out, const float* b_out,
                            int32_t Cout, void* out, int64_t out_sn, int32_t out_layout, float* out2, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_gru_conv -- the dilated 3x3 convolutions of the ConvGRU with their gate math, as cooperative bf16x3 kernels
 * (models/module.py:53-66; the GRU of models/itermvs.py:131-137: 32 hidden + 1 + 10 input channels, dilation 2, padding 2).
 *   mode 0: x = [h | inputs] [B,43,H,W];  z = sigmoid(convz(x)) -> out [B,32,H,W];  r = sigmoid(convr(x)), r * h -> out2
 *           (module.py:61-63).  w_packed / bias: the two layers stacked along the output channels (64).
 *   mode 1: x = [r*h | inputs];  q = tanh(convq(x));  h' = (1 - z) h + z q -> out (and out2 when not NULL)  (module.py:64-65).
 *           h may alias out (every element is read before it is written, by the same thread).
 * x, h, z, out, out2: fp32 channel planes of H*W elements (row stride W), batch strides *_sb in elements.
 * w_packed (16-byte aligned): bf16 [output block of 16][tap 9][operand 6][lane 64][8]; with lane = 16 q + i and (h, m, l) the exact
 * three-term bf16 split of W[16 ob + i][c][tap]:  operands 0..2 = h, m, l of channel c = (j / 4) * 16 + 4 q + j % 4;  operands
 * 3..5 of channel c = 32 + 8 (q % 2) + j (zero from 43 on) = h, m and (q < 2 ? l : h).
 * Arithmetic: fp32 accumulation of the six largest cross products (a few 1e-7 of the fp32 form's range; tests/test_kernels_gpu.py).
 * Replaces, in the inference step, two itermvs_conv2d launches per GRU iteration.
 * ------------------------------------------------------------------------------------------ */
int itermvs_gru_conv(const float* x, int64_t x_sb, int32_t B, int32_t H, int32_t W, int32_t mode, const void* w_packed,
                     const float* bias, const float* h, int64_t h_sb, const float* z, int64_t z_sb, float* out, int64_t out_sb,
                     float* out2, int64_t out2_sb, void* stream);



/* ------------------------------------------------------------------------------------------
 * ConvGRU gates -- models/module.py:59-66 (element-wise form for the traced / training path; in the inference
 * engine the gate math rides in the epilogue of the dilated 3x3 convolutions, itermvs_conv2d act 4/5)
 * itermvs_gru_rh :  rh[b,c,p] = sigmoid(zr[b,32+c,p]) * h[b,c,p]           (r * h, :63-64)
 * itermvs_gru_out:  h[b,c,p]  = (1-z) * h + z * tanh(q),  z = sigmoid(zr[b,c,p])   (:62,64,65)
 * zr = [B,2*hid,P] (z pre-activations then r pre-activations), q = [B,hid,P];
 * h / rh are addressed as base + b*sb + c*P + p so they can live inside the [B,43,P]
 * concatenated GRU input buffers.  itermvs_gru_out updates h in place and, when h_copy != NULL,
 * also stores the new state contiguously at h_copy[B,hid,P] (input of the depth / confidence head convolutions).
 * ------------------------------------------------------------------------------------------ */
int itermvs_gru_rh(const float* zr, const float* h, int64_t h_sb, float* rh, int64_t rh_sb,
                   int32_t B, int32_t hid, int32_t P, void* stream);
int itermvs_gru_out(const float* zr, const float* q, float* h, int64_t h_sb, float* h_copy,
                    int32_t B, int32_t hid, int32_t P, void* stream);

/* itermvs_pack_scores -- models/itermvs.py:122-124,193: concatenates the three CorrNet outputs
 * ([B,N_l,P] each) into channels [ch0, ch0+N0+N1+N2) of up to two [B,Ctot,P] buffers. */
int itermvs_pack_scores(const float* s0, const float* s1, const float* s2, const int32_t N[3],
                        int32_t B, int32_t P, float* dst0, float* dst1, int64_t dst_sb,
                        int32_t ch0, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_convex_upsample -- models/itermvs.py:262-264 (softmax over the 9 taps) +
 * models/module.py:127-140 (upsample) + models/module.py:148-152 (depth_unnormalization)
 *   logits [B,144,H,W] (element strides sb,sc,sy,sx; channel = k*16 + i*4 + j)
 *   nd     [B,1,H,W] at nd + b*nd_sb + y*W + x
 *   depth  [B,1,4H,4W] contiguous = 1/(inv_max + up*(inv_min-inv_max));
 *   when `norm_out` != NULL the un-normalised convex combination is also stored there.
 * ------------------------------------------------------------------------------------------ */
int itermvs_convex_upsample(const float* logits, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                            const float* nd, int64_t nd_sb, const float* inv_depth_min,
                            const float* inv_depth_max, int32_t B, int32_t H, int32_t W,
                            float* depth, float* norm_out, void* stream);
/* itermvs_final_upsample -- itermvs_convex_upsample (depth, models/itermvs.py:321-322) and, in the SAME launch, the x4 bilinear
 * up-sampling of the confidence (models/itermvs.py:323-324): conf [M,H,W] -> conf_up [M,4H,4W]. */
int itermvs_final_upsample(const float* logits, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* nd, int64_t nd_sb,
                           const float* inv_depth_min, const float* inv_depth_max, int32_t B, int32_t H, int32_t W, float* depth,
                           const float* conf, int32_t M, float* conf_up, void* stream);

/* itermvs_bilinear_up -- F.interpolate(x, scale_factor=s, mode='bilinear') for integer s
 * (models/itermvs.py:56,161,323): x [M,H,W] -> out [M,s*H,s*W]; `act`: 0 none, 1 tanh
 * (hidden_init, itermvs.py:162). */
int itermvs_bilinear_up(const float* x, int32_t M, int32_t H, int32_t W, int32_t scale,
                        int32_t act, float* out, void* stream);
/* same for x [B,C,H,W] with up to two destinations given as [C,sH,sW] planes at batch strides out_sb / out2_sb
 * (elements): lets the initial hidden state (itermvs.py:161-163) land in `hidden` and in channels 0..31 of the
 * GRU input buffer in one launch.  out2 may be NULL. */
int itermvs_bilinear_up2(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, int32_t scale, int32_t act,
                         float* out, int64_t out_sb, float* out2, int64_t out2_sb, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_conv2d -- the small-channel 2-D convolutions of the path, with fused epilogues:
 *   nn.Conv2d / ConvBnReLU (BN folded) / ConvReLU      models/module.py:6-30
 *   ResidualBlock's relu(x + y)                        models/module.py:33-50
 *   nn.ConvTranspose2d(3, stride 2, pad 1, out_pad 1)  models/itermvs.py:359-363 (CorrNet)
 *   ConvGRU gates                                      models/module.py:59-66
 * fp32, contiguous [C,H,W] planes; `in_sn` / `out_sn` / ... are BATCH strides in elements, so
 * inputs and outputs may be channel slices of wider buffers.  `weight` is PACKED:
 *   weight_format 0: [Cin][k][k][Cout] (conv weight.permute(1,2,3,0); transposed-conv
 *     weight.permute(0,2,3,1)) -- VALU kernels, required for `transposed`;
 *   weight_format 1: [k*k][Cin_pad4][Cout_pad16], zero padded -- matrix-core (MFMA f32 16x16x4)
 *     implicit-GEMM kernel (any 1x1 / 3x3 shape);
 *   weight_format 2: [9][chunks][4][Cout_pad16][S] with element (tap, ch, q, co, s) = weight of input
 *     channel ch*4*S + q*S + s (S = 1 / 2 / 4 for Cin <= 4 / <= 8 / larger), zero padded -- the LDS-tiled
 *     persistent matrix-core kernel for 3x3, stride 1|2, dilation 1|2;
 *   weight_format 3: bfloat16 [9][chunks][3][Cout_pad16][16] with element (tap, ch, p, co, c) = term p of the exact
 *     three-term bf16 split (h = w truncated to 8 significant bits, m = (w - h) truncated, l = w - h - m) of the weight of
 *     input channel ch*16 + c, zero padded -- the same LDS-tiled kernel on v_mfma_f32_16x16x32_bf16 for 3x3 layers with
 *     Cin > 8: activations are split the same way when staged, the six largest cross terms are accumulated in fp32
 *     (error ~2^-23 per product: fp32-rounding class, NOT bit-identical to the fp32 forms).  Up to three weight sets
 * per launch: batch items [0,seg_end[0]) use set 0, [seg_end[0],seg_end[1]) set 1, the rest set 2
 * (the three CorrNets of one iteration in one launch).
 * Epilogue `act`: 0 v+add | 1 relu(v+add) | 2 sigmoid | 3 tanh | 4 sigmoid(v)*aux1 (r*h) |
 *                 5 (1-aux2)*aux1 + aux2*tanh(v)  (GRU state update; aux1 = h, aux2 = z) |
 *                 6 sum_co relu(v[co]) * aux1[co] + aux1[Cout]: a following 1x1 convolution to ONE channel folded into the
 *                   epilogue (PixelViewWeight, models/itermvs.py:337-346); Cout = 16 or 32, 3x3, weight_format 2, `out` is the
 *                   single plane [N,1,H,W] (batch stride out_sn), aux1 = Cout + 1 floats shared by all batch items
 *                   (aux1_sn = 0) |
 *                 7 sigmoid of 6 (the confidence head, models/itermvs.py:147-151,198).
 * `add_mode` 0: `add` has the output's shape; 1: `add` is [N,Cout,Hout/2,Wout/2] and its x2 bilinear
 *   up-sampling (F.interpolate(scale_factor=2, mode='bilinear'), models/net.py:46,49) is evaluated
 *   in the epilogue (matrix-core formats only, Hout and Wout even).
 * `out_layout` 0: `out` is [Cout,H,W] planes per batch item (batch stride out_sn); 1: `out` is a dense
 *   channels-last [N,H,W,Cout] tensor (the layout the correlation kernels read; act 0, no `add`,
 *   Cout % 4 == 0, matrix-core formats only); 2 / 3: the same in fp16 / bf16 storage (round to nearest even of the
 *   fp32 result; `out` points to 16-bit elements, out_sn counts them).
 * `out2` (optional) receives a second, contiguous [N,Cout,H,W] copy of the result.
 * `split_cout` > 0 (matrix-core formats, multiple of 16): output channels [split_cout, Cout) form a SECOND
 *   result with its own activation `act_b` and destination `out_b` (planes, batch stride out_b_sn, channel
 *   index rebased to 0); channels [0, split_cout) go to `out` with `act`.  One launch then evaluates two
 *   convolutions of the same input -- the ConvGRU update and reset gates (models/module.py:61-63).
 * ------------------------------------------------------------------------------------------ */
typedef struct itermvs_conv_params {
    const float* in;
    float* out;
    float* out2;
    const float* add;
    const float* aux1;
    const float* aux2;
    int64_t in_sn, out_sn, add_sn, aux1_sn, aux2_sn;
    const float* weight[3];
    const float* bias[3];                      /* [Cout] or NULL */
    int32_t seg_end[3];
    int32_t n_seg;
    int32_t N, Cin, Hin, Win, Cout;
    int32_t ksize, stride, pad, dilation;      /* ksize 1 or 3 */
    int32_t transposed;                        /* 1: ConvTranspose2d(3, stride 2, pad 1, output_padding 1) */
    int32_t act;
    int32_t weight_format;
    int32_t add_mode;
    int32_t out_layout;
    int32_t split_cout;
    int32_t act_b;
    int32_t in_layout;                         /* 0 = `in` [N,Cin,Hin,Win] planes; 1 = channels last [N,Hin,Win,8] -- only the 3x3 layers with
                                                * exactly 8 input channels in weight_format 3 (stride 1, no dilation: PixelViewWeight's layer
                                                * reading itermvs_corr_init's out_layout 1); `in` 16-byte aligned, in_sn a multiple of 4 */
    float* out_b;
    int64_t out_b_sn;
} itermvs_conv_params;

int itermvs_conv2d(const itermvs_conv_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_corrnet -- the whole CorrNet (models/itermvs.py:352-381: conv0..conv2, the two transposed convolutions with
 * their skip additions, conv5 + bias) in ONE launch, every intermediate in LDS (32 x 32 output tiles, halos recomputed).
 *   x [M,8,H,W] planes (batch stride x_sn), H and W multiples of 4;  out / out2 (optional) [M,1,H,W] at batch strides
 *   out_sn / out2_sn (the GRU input buffers take the scores in place);
 *   weights: host array of n_seg (1..3) device pointers to PACKED weight sets, batch items [0,seg_end[0]) use set 0,
 *   [seg_end[0],seg_end[1]) set 1, the rest set 2 (the three CorrNets of one GRU iteration in one launch).
 *   Packed set (fp32, 14288 floats, 16-byte aligned): the five matrix-core layers in operand order
 *   [tap = ky*3+kx][k-step = ci/4][q = ci%4][co, zero padded] -- conv0 in its two-rows-per-tile form ([window row 4][kx 3]
 *   [k-step][q][16]: columns 0..7 = channel co with tap row = window row, columns 8..15 = channel co - 8 with tap row =
 *   window row - 1, zero where that is outside 0..2) | conv1 (16) | conv2 (32) | conv3, transposed (16) | conv4, transposed
 *   (16) -- then conv5 as [ci 8][tap 9], the bias, 7 pad (itermvs_amd.ops.pack_corrnet_weights).
 * ------------------------------------------------------------------------------------------ */
int itermvs_corrnet(const float* x, int64_t x_sn, const float* const* weights, const int32_t* seg_end, int32_t n_seg,
                    int32_t M, int32_t H, int32_t W, float* out, int64_t out_sn, float* out2, int64_t out2_sn, void* stream);
/* itermvs_corrnet_bf16x3 -- the same launch with conv0, conv2 and the two transposed convolutions on v_mfma_f32_16x16x32_bf16:
 * both operands split EXACTLY into three bf16 terms, the six largest cross products accumulated in fp32 (error of the size of one
 * fp32 rounding per product, like itermvs_conv2d's weight_format 3); conv1 stays on the fp32 instruction (its input is kept in fp32
 * for the skip connection and the last layer).  Packed set: 23120 floats = conv0 as bf16 A operands [18 MFMAs][64 lanes][8 bf16]
 * (4608 floats: MFMA 2v / 2v+1 = terms h / m of the window-position pair v, 12 + v = [l l | h h]; lane (q, m): window position
 * 2v + (q & 1), output row-channel m as in the fp32 form) | conv1 as in itermvs_corrnet (1152 floats) | conv2, conv3, conv4 in
 * weight_format 3 = bf16 [tap 9][chunk ci/16][term h, m, l][row co][16 ci] with 32 / 16 / 16 rows (6912 + 6912 + 3456 floats; a
 * transposed convolution as the convolution weight [co][ci][ky][kx] = w[ci][co][ky][kx]) | conv5 [ci 8][tap 9], the bias, 7 pad
 * (itermvs_amd.ops.pack_corrnet_weights(split3=True)). */
int itermvs_corrnet_bf16x3(const float* x, int64_t x_sn, const float* const* weights, const int32_t* seg_end, int32_t n_seg,
                           int32_t M, int32_t H, int32_t W, float* out, int64_t out_sn, float* out2, int64_t out2_sn, void* stream);


/* ------------------------------------------------------------------------------------------
 * itermvs_stem -- the first two layers of FeatureNet in ONE launch (models/net.py:13-14,39-40; models/module.py:33-50):
 * FeatureNet.conv1 (ConvBnReLU 3 -> 8) and, from its result without a round trip through HBM, layer1[0].conv1
 * (ConvBnReLU 8 -> 16, stride 2) and layer1[0].downsample (ConvBn 8 -> 16, stride 2).  BatchNorm folded by the caller.
 *   x [M,3,H,W] planes (batch stride x_sn);  y = relu(conv1 branch), sc = downsample branch: [M,16,H2,W2] planes at batch
 *   stride out_sn, H2 = (H-1)/2+1;
 *   w0: 224 floats = conv1 weight as [ci][ky][kx][co 8] then its 8 biases;
 *   w1: 2336 floats = the two stride-2 layers' weights, output channels concatenated (conv1 branch first), in matrix-core
 *   operand order [tap = ky*3+kx][k-step = ci/4][q = ci%4][co 32], then the 32 biases (itermvs_amd.ops.pack_stem_weights).
 *   out_layout: 0 = planes [M,16,H2,W2]; 1 = channel quads [M,4,H2,W2,4] (channel 4*cq + c of pixel p at ((cq*H2*W2 + p)*4 + c):
 *   the layout itermvs_res_chain16 fetches with 16-byte loads; y, sc 16-byte aligned, out_sn a multiple of 4).
 * ------------------------------------------------------------------------------------------ */
int itermvs_stem(const float* x, int64_t x_sn, int32_t M, int32_t H, int32_t W, const float* w0, const float* w1,
                 float* y, float* sc, int64_t out_sn, int32_t out_layout, void* stream);
/* itermvs_stem_compose -- itermvs_stem and, in the SAME launch (its first workgroup), itermvs_compose_proj with its arguments
 * (mats .. inv_max; Bd = batch size of the depth range; module.py:77-90, itermvs.py:267-268): the composition depends on the
 * cameras only, so its ~10 us fp64 chain hides behind the first, longest launch of FeatureNet. */
int itermvs_stem_compose(const float* x, int64_t x_sn, int32_t M, int32_t H, int32_t W, const float* w0, const float* w1,
                         float* y, float* sc, int64_t out_sn, int32_t out_layout, const float* mats, int32_t n_sets, int32_t V, float* proj_out,
                         int32_t* nan_flag, const float* depth_min, const float* depth_max, int32_t Bd, float* inv_min,
                         float* inv_max, void* stream);



/* ------------------------------------------------------------------------------------------
 * itermvs_fuse_depth -- the filter that follows the depth-inference path:
 *   reproject_with_depth (eval.py:154-194), check_geometric_consistency (eval.py:197-212) and the per-reference
 *   arithmetic of filter_depth (eval.py:238-269) for ONE reference view against its S source views, one pass.
 * depth_ref / conf_ref [H,W] fp32; depth_src: HOST array of S device pointers to [H,W] fp32 depth maps;
 * mats: device [S][60] fp32 = per pair  inv(K_ref)(9) | (E_src inv(E_ref)) rows 0..2 (12) | K_src(9) | inv(K_src)(9) |
 *   (E_ref inv(E_src)) rows 0..2 (12) | K_ref(9), inverted / composed on the host in float32 like eval.py does.
 * Outputs: depth_avg [H,W] fp64 = (sum of consistent reprojected depths + depth_ref) / (count + 1); optional uint8
 *   masks photo (conf > photo_thres), geo (count >= geo_mask_thres), final (both); optional int32 count.
 * ------------------------------------------------------------------------------------------ */
int itermvs_fuse_depth(const float* depth_ref, const float* conf_ref, const float* const* depth_src,
                       const float* mats, int32_t S, int32_t H, int32_t W, double geo_pixel_thres,
                       float geo_depth_thres, float photo_thres, int32_t geo_mask_thres, double* depth_avg,
                       uint8_t* photo_mask, uint8_t* geo_mask, uint8_t* final_mask, int32_t* geo_sum,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * itermvs_image_pyramid -- the input side of the path: datasets/dtu_yao_eval.py:61-74
 * (read_img) on the GPU.  src [V,Hs,Ws,3] uint8 interleaved RGB (the decoded images of one sample, same size) ->
 *   level0 [V,3,H,W]       = cv2.resize(2 * src / 255. - 1, (W, H), INTER_LINEAR)   (float32)
 *   level1..3 [V,3,H>>l,W>>l] = cv2.resize(level0, ..., INTER_LINEAR)               (may be NULL: the network reads level 0 only)
 * H and W multiples of 8 when the lower levels are requested.  cv2's algorithm is restated from its published form.
 * ------------------------------------------------------------------------------------------ */
int itermvs_image_pyramid(const uint8_t* src, int32_t V, int32_t Hs, int32_t Ws, int32_t H, int32_t W, float* level0,
                          float* level1, float* level2, float* level3, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm (+ ReLU) of FeatureNet's ConvBnReLU / ConvBn layers (models/module.py:33-50 under
 * train.py:194-243; torch.nn.BatchNorm2d in train() mode): x, y, dy, dx are dense NCHW fp32 [N,C,H*W].
 *   forward : batch mean / biased variance per channel -> y = [relu]((x - mean) / sqrt(var + eps) * gamma + beta);
 *             save_mean, save_invstd [C] are written for the backward; running_mean / running_var (may be NULL) are
 *             updated in place with `momentum` (running_var with the unbiased variance, like torch).
 *   backward: dx, dgamma, dbeta from dy (the gradient w.r.t. y; the ReLU mask is recomputed from x).
 * `workspace`: itermvs_bn_workspace_floats(N, C, HW) floats of scratch (partials; same size for both directions).
 * ------------------------------------------------------------------------------------------ */
int itermvs_bn_workspace_floats(int32_t N, int32_t C, int32_t HW);
int itermvs_bn_train_forward(const float* x, float* y, int32_t N, int32_t C, int32_t HW, const float* gamma, const float* beta,
                             float eps, float momentum, int32_t relu, float* running_mean, float* running_var,
                             float* save_mean, float* save_invstd, float* workspace, void* stream);
int itermvs_bn_train_backward(const float* x, const float* dy, float* dx, int32_t N, int32_t C, int32_t HW, const float* gamma,
                              const float* beta, const float* save_mean, const float* save_invstd, int32_t relu,
                              float* dgamma, float* dbeta, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optional per-launch timing (HIP events recorded on the launch stream around the kernels of
 * itermvs_corr_iter / itermvs_corr_init).  Used by bench.py for the roofline figure.
 * itermvs_profile_enable(n) allocates n event pairs (n = 0 disables and frees);
 * itermvs_profile_collect synchronises the events and returns the number of samples copied:
 * kind[i] (1 = corr_iter, 2 = corr_init) and ms[i].
 * ------------------------------------------------------------------------------------------ */
int itermvs_profile_enable(int32_t capacity);
/* which launches are timed: bit 0 = corr_iter (kind 1), bit 1 = corr_init (kind 2), bit 2 = itermvs_conv2d
 * (kind 3).  Default 0x3. */
int itermvs_profile_set_mask(int32_t mask);
int itermvs_profile_collect(int32_t* kind, float* ms, int32_t max_samples);
/* Launches captured into a hipGraph while profiling is enabled are bracketed by external event-record nodes:
 * itermvs_profile_graph_count() = number of such pairs so far (capture order);
 * itermvs_profile_graph_read(first, count, kind, ms) waits for pairs [first, first+count) of the LATEST replay of
 * their graph and returns how many were read. */
int itermvs_profile_graph_count(void);
int itermvs_profile_graph_read(int32_t first, int32_t count, int32_t* kind, float* ms);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* ITERMVS_HIP_H */
