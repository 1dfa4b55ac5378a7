// Library identification, error strings and the optional per-launch timing hooks.
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

static bool capturing(hipStream_t stream) {   // launches recorded into a hipGraph are not timed
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

void itermvs_profile_begin(int kind, hipStream_t stream) {
    if (!g_enabled || !((g_mask >> (kind - 1)) & 1) || capturing(stream)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used >= (int)g_pool.size()) return;
    g_pool[g_used].kind = kind;
    (void)hipEventRecord(g_pool[g_used].t0, stream);
}

void itermvs_profile_cancel() {}  // a begun sample without an end is simply overwritten by the next begin

void itermvs_profile_end(int kind, hipStream_t stream) {
    if (!g_enabled || !((g_mask >> (kind - 1)) & 1) || capturing(stream)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used >= (int)g_pool.size() || g_pool[g_used].kind != kind) return;
    (void)hipEventRecord(g_pool[g_used].t1, stream);
    ++g_used;
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
