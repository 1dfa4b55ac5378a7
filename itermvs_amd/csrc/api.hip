// Library identification, error strings and the optional per-launch timing hooks.
#include <stdio.h>
#include <mutex>
#include <vector>

#include "common.hpp"

extern "C" int itermvs_version(void) { return ITERMVS_ABI_VERSION; }

extern "C" const char* itermvs_error_string(int status) {
    switch (status) {
        case ITERMVS_OK: return "ok";
        case ITERMVS_ERR_NULL: return "required pointer is NULL";
        case ITERMVS_ERR_DIMS: return "non-positive or inconsistent dimension";
        case ITERMVS_ERR_CHANNELS: return "unsupported channel count (C must be 16, 32 or 48; C/8 in {2,4,6})";
        case ITERMVS_ERR_VIEWS: return "number of source views out of range [1, ITERMVS_MAX_SRC]";
        case ITERMVS_ERR_ALIGN: return "pointer or stride not aligned for the vector path (16 bytes)";
        case ITERMVS_ERR_LAYOUT: return "fused kernels need channels-last feature maps (channel stride 1)";
        case ITERMVS_ERR_LAUNCH: return "HIP kernel launch failed";
        case ITERMVS_ERR_DTYPE: return "feature storage type not supported by this entry point";
        default: return "unknown itermvs status";
    }
}

// ---------------------------------------------------------------------------------------------
// Timing hooks.  Off by default (zero cost: one relaxed flag test per launch).  This is the only
// global state in the library and exists for bench.py's roofline measurement.
// ---------------------------------------------------------------------------------------------
namespace {
struct Sample {
    hipEvent_t t0, t1;
    int kind;
};
std::mutex g_mu;
std::vector<Sample> g_pool;
int g_used = 0;
bool g_enabled = false;
int g_mask = 0x3;   // kinds 1 (corr_iter) and 2 (corr_init) by default; bit 2 = itermvs_conv2d launches
}  // namespace

static bool capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// Launches captured into a hipGraph are bracketed by EXTERNAL event-record nodes (hipEventRecordExternal):
// every replay of the graph re-records the same pair, which itermvs_profile_graph_read turns into that replay's
// kernel duration (tools/ubench/graph_event_timing.hip: identical to eager event timing).  The pairs live as long
// as the library; they are created only while profiling is enabled at capture time.
namespace {
std::vector<Sample> g_graph;      // pairs embedded in captured graphs, in capture order
std::vector<Sample> g_graph_free; // created by itermvs_profile_enable (events cannot be created while capturing)
Sample g_open;                    // pair whose begin was captured and whose end is pending
bool g_open_valid = false;
}  // namespace

// An event-record NODE spliced into the stream capture: the graph being captured and its current frontier come
// from hipStreamGetCaptureInfo_v2, the node is added by hand and made the new frontier.  (hipEventRecordWithFlags(
// ..., hipEventRecordExternal) does the same in one call on ROCm 7.2 but returns hipErrorInvalidValue with the
// HIP runtime bundled in the PyTorch 2.10+rocm7.0 wheel, which is the one loaded in a torch process.)
static bool capture_event_record(hipEvent_t ev, hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    hipGraphNode_t node = nullptr;
    bool ok = hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps) == hipSuccess &&
              st == hipStreamCaptureStatusActive && graph != nullptr &&
              hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev) == hipSuccess &&
              hipStreamUpdateCaptureDependencies(stream, &node, 1, hipStreamSetCaptureDependencies) == hipSuccess;
    if (!ok) {
        fprintf(stderr, "[itermvs] could not add an event-record node to the captured graph: %s\n",
                hipGetErrorString(hipGetLastError()));
    }
    return ok;
}

void itermvs_profile_begin(int kind, hipStream_t stream) {
    if (!g_enabled || !((g_mask >> (kind - 1)) & 1)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (capturing(stream)) {
        if (g_graph_free.empty()) return;
        Sample s = g_graph_free.back();
        g_graph_free.pop_back();
        s.kind = kind;
        if (!capture_event_record(s.t0, stream)) return;
        g_open = s;
        g_open_valid = true;
        return;
    }
    if (g_used >= (int)g_pool.size()) return;
    g_pool[g_used].kind = kind;
    (void)hipEventRecord(g_pool[g_used].t0, stream);
}

void itermvs_profile_cancel() {}  // a begun sample without an end is simply overwritten by the next begin

void itermvs_profile_end(int kind, hipStream_t stream) {
    if (!g_enabled || !((g_mask >> (kind - 1)) & 1)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (capturing(stream)) {
        if (!g_open_valid || g_open.kind != kind) return;
        g_open_valid = false;
        if (!capture_event_record(g_open.t1, stream)) return;
        g_graph.push_back(g_open);
        return;
    }
    if (g_used >= (int)g_pool.size() || g_pool[g_used].kind != kind) return;
    (void)hipEventRecord(g_pool[g_used].t1, stream);
    ++g_used;
}

extern "C" int itermvs_profile_graph_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_graph.size();
}

extern "C" int itermvs_profile_graph_read(int32_t first, int32_t count, int32_t* kind, float* ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (int i = first; i < first + count && i < (int)g_graph.size() && i >= 0; ++i) {
        if (hipEventSynchronize(g_graph[i].t1) != hipSuccess) break;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, g_graph[i].t0, g_graph[i].t1) != hipSuccess) break;
        if (kind) kind[n] = g_graph[i].kind;
        if (ms) ms[n] = t;
        ++n;
    }
    return n;
}

extern "C" int itermvs_profile_set_mask(int32_t mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = mask;
    return ITERMVS_OK;
}

extern "C" int itermvs_profile_enable(int32_t capacity) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& s : g_pool) {
        (void)hipEventDestroy(s.t0);
        (void)hipEventDestroy(s.t1);
    }
    g_pool.clear();
    g_used = 0;
    g_enabled = capacity > 0;
    // spare pairs for launches captured into hipGraphs while profiling is on (kept by the graphs that use them)
    while (g_enabled && (int)g_graph_free.size() < 64) {
        Sample s;
        s.kind = 0;
        if (hipEventCreate(&s.t0) != hipSuccess || hipEventCreate(&s.t1) != hipSuccess) return ITERMVS_ERR_LAUNCH;
        g_graph_free.push_back(s);
    }
    for (int i = 0; i < capacity; ++i) {
        Sample s;
        s.kind = 0;
        if (hipEventCreate(&s.t0) != hipSuccess || hipEventCreate(&s.t1) != hipSuccess) return ITERMVS_ERR_LAUNCH;
        g_pool.push_back(s);
    }
    return ITERMVS_OK;
}

extern "C" int itermvs_profile_collect(int32_t* kind, float* ms, int32_t max_samples) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (int i = 0; i < g_used && n < max_samples; ++i) {
        if (hipEventSynchronize(g_pool[i].t1) != hipSuccess) break;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, g_pool[i].t0, g_pool[i].t1) != hipSuccess) break;
        if (kind) kind[n] = g_pool[i].kind;
        if (ms) ms[n] = t;
        ++n;
    }
    g_used = 0;
    return n;
}


// ---------------------------------------------------------------------------------------------
// itermvs_copy_multi: up to 8 device-to-device copies in ONE launch (the staging of a sample into the static inputs of a
// captured graph: 20 MB of images + five small tensors would otherwise be a memcpy and a multi-tensor kernel)
// ---------------------------------------------------------------------------------------------
namespace itermvs {
struct CopyMultiArgs {
    const char* src[8];
    char* dst[8];
    int64_t bytes[8];
};
__global__ void __launch_bounds__(256) copy_multi_kernel(CopyMultiArgs a) {
    const int k = blockIdx.y;
    const int64_t n = a.bytes[k];
    const char* __restrict__ s = a.src[k];
    char* __restrict__ d = a.dst[k];
    const int64_t n16 = (((uintptr_t)s | (uintptr_t)d) & 15) == 0 ? n / 16 : 0;       // 16-byte pieces when both are aligned
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (int64_t i = n16 * 16 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
}  // namespace itermvs

extern "C" int itermvs_copy_multi(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream) {
    using namespace itermvs;
    ITERMVS_RETURN_IF(!src || !dst || !bytes, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(n < 1 || n > 8, ITERMVS_ERR_DIMS);
    CopyMultiArgs a{};
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        ITERMVS_RETURN_IF(bytes[i] < 0, ITERMVS_ERR_DIMS);
        ITERMVS_RETURN_IF(bytes[i] > 0 && (!src[i] || !dst[i]), ITERMVS_ERR_NULL);
        a.src[i] = (const char*)src[i]; a.dst[i] = (char*)dst[i]; a.bytes[i] = bytes[i];
        most = bytes[i] > most ? bytes[i] : most;
    }
    if (most == 0) return ITERMVS_OK;
    int64_t blocks = (most / 16 + 256 * 4 - 1) / (256 * 4);          // four 16-byte pieces per thread of the largest tensor
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    return itermvs_launch_status();
}


// ---------------------------------------------------------------------------------------------
// itermvs_box_probe: bench.py's box calibration (a fixed fp32-MFMA issue loop; clocks of workgroup 0)
// ---------------------------------------------------------------------------------------------
namespace itermvs {
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void __launch_bounds__(256) box_probe_kernel(float* __restrict__ sink, int iters, unsigned long long* __restrict__ clocks) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (clocks && blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = r1 - r0;
    }
}
}  // namespace itermvs

namespace itermvs {
// one lane follows ring[i] -> ring[ring[i]] ...: the latency of a dependent load (L2 / memory-side cache / HBM by the ring's size)
__global__ void __launch_bounds__(64) box_chase_kernel(const uint32_t* __restrict__ ring, uint32_t start, int steps,
                                                       uint32_t* __restrict__ out, unsigned long long* __restrict__ clocks) {
    if (threadIdx.x != 0) return;
    uint32_t i = start;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) i = __builtin_nontemporal_load(ring + i);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = i;                      // where the walk ended: the next call starts there (lines not touched before)
    clocks[0] = r1 - r0;
    clocks[1] = c1 - c0;             // shader-clock ticks of a nearly idle chip over the same interval
}

// The 100 MHz counter and the shader-clock counter of every XCD (the counters are per XCD and not aligned with each other): 64
// one-wave workgroups, each writes { s_memrealtime, s_memtime } to slot XCC_ID of `out` (uint64[16][2]).  Two launches bracket a
// stretch of other work on the stream; per XCD, delta(shader ticks) / delta(100 MHz ticks) = the clock sustained under that work.
__global__ void __launch_bounds__(64) clock_stamp_kernel(unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;        // HW_REG_XCC_ID[3:0]
    out[2 * xcc] = __builtin_amdgcn_s_memrealtime();
    out[2 * xcc + 1] = __builtin_amdgcn_s_memtime();
}
}  // namespace itermvs

extern "C" int itermvs_box_chase(const uint32_t* ring, uint32_t start, int32_t steps, uint32_t* out, uint64_t* clocks, void* stream) {
    ITERMVS_RETURN_IF(!ring || !out || !clocks, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(steps < 1, ITERMVS_ERR_DIMS);
    hipLaunchKernelGGL(itermvs::box_chase_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ring, start, steps, out,
                       reinterpret_cast<unsigned long long*>(clocks));
    return itermvs_launch_status();
}

extern "C" int itermvs_clock_stamp(uint64_t* out, void* stream) {
    ITERMVS_RETURN_IF(!out, ITERMVS_ERR_NULL);
    hipLaunchKernelGGL(itermvs::clock_stamp_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(out));
    return itermvs_launch_status();
}

extern "C" int itermvs_box_probe(float* sink, int32_t blocks, int32_t iters, uint64_t* clocks, void* stream) {
    ITERMVS_RETURN_IF(!sink, ITERMVS_ERR_NULL);
    ITERMVS_RETURN_IF(blocks < 1 || iters < 1, ITERMVS_ERR_DIMS);
    hipLaunchKernelGGL(itermvs::box_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sink, iters,
                       reinterpret_cast<unsigned long long*>(clocks));
    return itermvs_launch_status();
}
