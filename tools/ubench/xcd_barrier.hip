// Device-wide barrier inside one persistent kernel on MI355X, priced against a kernel boundary: the decision input for
// fusing the ConvGRU gates + depth head of one GRU iteration into one launch.
//
// Forms (all placement-independent: groups are blockIdx % 8, which lands on XCD b % 8 in practice -- for speed only):
//   flat    one monotonic counter, every workgroup: lane-0 release fence -> atomic arrive -> relaxed sc1 poll -> acquire fence
//   xcd     hierarchical: per-group counter; the group's last arriver bumps a top counter; the top's last arriver stores
//           the generation word of every group; every workgroup polls its own group's generation word
// Payload per phase (what the fused GRU tail would exchange): each workgroup writes `per_block` floats and, after the
// barrier, reads the slices of two neighbouring workgroups (other CUs, other XCDs) and checks every word.
//   plain   plain stores + lane-0 agent release fence
//   wt      write-through (sc1) stores, every wave drains vmcnt, no release fence
// Timing is host-paired: (kernel with N barriers - kernel with 0 barriers) / N, best of 7.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_barrier xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned __attribute__((address_space(1))) gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct BarrierState {            // every polled word on its own 128-byte line; zeroed by a memset before each launch
    unsigned top[32];
    unsigned grp[8][32];
    unsigned gen[8][32];
    unsigned flat[32];
    unsigned timeout[32];
};

__device__ __forceinline__ bool spin_until(unsigned* word, unsigned want, unsigned* tmo) {
    unsigned spins = 0;
    while (__hip_atomic_load(word, RLX_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, RLX_AGENT); return false; }
    }
    return true;
}

template <bool HIER, bool RELEASE>
__device__ __forceinline__ void grid_barrier(BarrierState* st, unsigned epoch, unsigned n_blocks) {
    __syncthreads();                                       // every wave's stores are issued and waited for (vmcnt(0) + s_barrier)
    if (threadIdx.x == 0) {
        if (RELEASE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (HIER) {
            const unsigned g = blockIdx.x & 7u;
            const unsigned gsize = (n_blocks + 7u - g) / 8u;   // blocks b with b % 8 == g
            const unsigned old = __hip_atomic_fetch_add(&st->grp[g][0], 1u, RLX_AGENT);
            if (old == gsize * epoch - 1u) {
                const unsigned ngroups = n_blocks < 8u ? n_blocks : 8u;
                const unsigned old2 = __hip_atomic_fetch_add(&st->top[0], 1u, RLX_AGENT);
                if (old2 == ngroups * epoch - 1u)
                    for (unsigned k = 0; k < ngroups; ++k) __hip_atomic_store(&st->gen[k][0], epoch, RLX_AGENT);
            }
            spin_until(&st->gen[g][0], epoch, &st->timeout[0]);
        } else {
            __hip_atomic_fetch_add(&st->flat[0], 1u, RLX_AGENT);
            spin_until(&st->flat[0], n_blocks * epoch, &st->timeout[0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// MODE 0: no payload; 1: plain stores + lane-0 release fence
template <bool HIER, int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) phases(float* buf, BarrierState* st, int n_phase, int per_block, unsigned* errors) {
    const unsigned nb = gridDim.x;
    unsigned bad = 0;
    for (int ph = 0; ph < n_phase; ++ph) {
        if (MODE) {
            float* dst = buf + (size_t)(ph & 1) * nb * per_block + (size_t)blockIdx.x * per_block;
            const float* src = buf + (size_t)((ph + 1) & 1) * nb * per_block;
            const unsigned o1 = (blockIdx.x + 1) % nb, o2 = (blockIdx.x + nb / 2 + 3) % nb;
            for (int i = threadIdx.x; i < per_block; i += THREADS) {
                float v = 0.0f;
                if (ph) {
                    const float a = src[(size_t)o1 * per_block + i], b = src[(size_t)o2 * per_block + i];
                    bad += (a != (float)ph) + (b != (float)ph);
                    v = a;
                }
                dst[i] = v + 1.0f;
            }
        }
        grid_barrier<HIER, MODE == 1>(st, (unsigned)ph + 1u, nb);
    }
    if (bad) atomicAdd(errors, bad);
}

// write-through form: global_store_dword ... sc1, every wave drains vmcnt, no release fence
template <int THREADS, bool HIER>
__global__ void __launch_bounds__(THREADS) phases_wt(float* buf, BarrierState* st, int n_phase, int per_block, unsigned* errors) {
    const unsigned nb = gridDim.x;
    unsigned bad = 0;
    for (int ph = 0; ph < n_phase; ++ph) {
        float* dst = buf + (size_t)(ph & 1) * nb * per_block + (size_t)blockIdx.x * per_block;
        const float* src = buf + (size_t)((ph + 1) & 1) * nb * per_block;
        const unsigned o1 = (blockIdx.x + 1) % nb, o2 = (blockIdx.x + nb / 2 + 3) % nb;
        for (int i = threadIdx.x; i < per_block; i += THREADS) {
            float v = 0.0f;
            if (ph) {
                const float a = src[(size_t)o1 * per_block + i], b = src[(size_t)o2 * per_block + i];
                bad += (a != (float)ph) + (b != (float)ph);
                v = a;
            }
            const float w = v + 1.0f;
            float* p = dst + i;
            asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_barrier<HIER, false>(st, (unsigned)ph + 1u, nb);
    }
    if (bad) atomicAdd(errors, bad);
}

template <typename F>
static float best_ms(F launch, BarrierState* st, unsigned* err, int reps = 7) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        (void)hipMemsetAsync(st, 0, sizeof(BarrierState), 0);
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    (void)err;
    return best;
}

template <bool HIER, int MODE, int THREADS>
static void run_case(const char* name, int nb, int per_block, float* buf, BarrierState* st, unsigned* err) {
    const int n = 40;
    (void)hipMemset(err, 0, 4);
    auto l0 = [&] { hipLaunchKernelGGL((phases<HIER, MODE, THREADS>), dim3(nb), dim3(THREADS), 0, 0, buf, st, 1, per_block, err); };
    auto ln = [&] { hipLaunchKernelGGL((phases<HIER, MODE, THREADS>), dim3(nb), dim3(THREADS), 0, 0, buf, st, n + 1, per_block, err); };
    const float t0 = best_ms(l0, st, err), tn = best_ms(ln, st, err);
    unsigned e = 0, tmo = 0;
    (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&tmo, &st->timeout[0], 4, hipMemcpyDeviceToHost);
    printf("%-34s %4d WGs x %3d thr, %5.1f KB/WG: %6.2f us per phase+barrier  (1 phase %.1f us; stale words %u; timeout %u)\n",
           name, nb, THREADS, MODE ? per_block * 4 / 1024.0f : 0.0f, (tn - t0) * 1e3f / n, t0 * 1e3f, e, tmo);
}

template <bool HIER, int THREADS>
static void run_wt(const char* name, int nb, int per_block, float* buf, BarrierState* st, unsigned* err) {
    const int n = 40;
    (void)hipMemset(err, 0, 4);
    auto l0 = [&] { hipLaunchKernelGGL((phases_wt<THREADS, HIER>), dim3(nb), dim3(THREADS), 0, 0, buf, st, 1, per_block, err); };
    auto ln = [&] { hipLaunchKernelGGL((phases_wt<THREADS, HIER>), dim3(nb), dim3(THREADS), 0, 0, buf, st, n + 1, per_block, err); };
    const float t0 = best_ms(l0, st, err), tn = best_ms(ln, st, err);
    unsigned e = 0, tmo = 0;
    (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&tmo, &st->timeout[0], 4, hipMemcpyDeviceToHost);
    printf("%-34s %4d WGs x %3d thr, %5.1f KB/WG: %6.2f us per phase+barrier  (1 phase %.1f us; stale words %u; timeout %u)\n",
           name, nb, THREADS, per_block * 4 / 1024.0f, (tn - t0) * 1e3f / n, t0 * 1e3f, e, tmo);
}

__global__ void __launch_bounds__(256) one_phase(float* buf, int per_block, int ph) {
    const unsigned nb = gridDim.x;
    float* dst = buf + (size_t)(ph & 1) * nb * per_block + (size_t)blockIdx.x * per_block;
    const float* src = buf + (size_t)((ph + 1) & 1) * nb * per_block;
    const unsigned o1 = (blockIdx.x + 1) % nb;
    for (int i = threadIdx.x; i < per_block; i += 256) dst[i] = src[(size_t)o1 * per_block + i] + 1.0f;
}

int main() {
    const int max_nb = 1024, max_pb = 8192;
    float* buf; BarrierState* st; unsigned* err;
    (void)hipMalloc(&buf, (size_t)2 * max_nb * max_pb * 4);
    (void)hipMalloc(&st, sizeof(BarrierState));
    (void)hipMalloc(&err, 4);
    (void)hipMemset(buf, 0, (size_t)2 * max_nb * max_pb * 4);
    printf("== no payload ==\n");
    for (int nb : {256, 512, 1024}) {
        run_case<false, 0, 256>("flat counter", nb, 0, buf, st, err);
        run_case<true, 0, 256>("xcd-hierarchical", nb, 0, buf, st, err);
    }
    run_case<true, 0, 512>("xcd-hierarchical", 256, 0, buf, st, err);
    printf("== 10 KB per workgroup written, two neighbours' slices read and checked after the barrier ==\n");
    for (int nb : {256, 512}) {
        run_case<false, 1, 256>("flat, plain stores + release", nb, 2560, buf, st, err);
        run_case<true, 1, 256>("xcd, plain stores + release", nb, 2560, buf, st, err);
        run_wt<true, 256>("xcd, sc1 write-through stores", nb, 2560, buf, st, err);
    }
    run_case<true, 1, 512>("xcd, plain stores + release", 256, 2560, buf, st, err);
    run_wt<true, 512>("xcd, sc1 write-through stores", 256, 2560, buf, st, err);
    printf("== 32 KB per workgroup ==\n");
    run_case<true, 1, 256>("xcd, plain stores + release", 256, 8192, buf, st, err);
    run_wt<true, 256>("xcd, sc1 write-through stores", 256, 8192, buf, st, err);
    // the same 10 KB phases as dependent kernel launches (stream order = the barrier)
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int nb : {256, 512}) {
            const int n = 40;
            float best = 1e9f;
            for (int r = 0; r < 7; ++r) {
                (void)hipEventRecord(e0, 0);
                for (int i = 0; i < n; ++i) hipLaunchKernelGGL(one_phase, dim3(nb), dim3(256), 0, 0, buf, 2560, i);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("%-34s %4d WGs x 256 thr,  10.0 KB/WG: %6.2f us per phase+boundary (eager launches, host-enqueue bound below ~3 us)\n",
                   "separate launches", nb, best * 1e3f / n);
        }
        // the same chain replayed from a hipGraph
        for (int nb : {256, 512}) {
            const int n = 40;
            hipStream_t s; (void)hipStreamCreate(&s);
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(one_phase, dim3(nb), dim3(256), 0, s, buf, 2560, i);
            (void)hipStreamEndCapture(s, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            float best = 1e9f;
            for (int r = 0; r < 7; ++r) {
                (void)hipEventRecord(e0, s);
                (void)hipGraphLaunch(ge, s);
                (void)hipEventRecord(e1, s);
                (void)hipStreamSynchronize(s);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("%-34s %4d WGs x 256 thr,  10.0 KB/WG: %6.2f us per phase+boundary (hipGraph replay)\n", "separate launches", nb, best * 1e3f / n);
        }
    }
    return 0;
}
