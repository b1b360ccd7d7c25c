// Internal helpers shared by the HIP translation units of libboa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <unordered_map>
#include <mutex>
#include <vector>

#include "boa_hip.h"

void boa_set_error(const char* fmt, ...);

#define BOA_HIP_TRY(expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            boa_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (_e == hipErrorOutOfMemory) ? BOA_ENOMEM : BOA_EHIP;                         \
        }                                                                                       \
    } while (0)

#define BOA_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            boa_set_error(__VA_ARGS__);       \
            return BOA_EINVAL;                \
        }                                     \
    } while (0)

#define BOA_TRY(expr)               \
    do {                            \
        int _r = (expr);            \
        if (_r != BOA_OK) return _r; \
    } while (0)

// Per-launch timing with ONE event per launch: launches on the context's stream are back to back, so the end event of
// launch i is the start event of launch i + 1 (chain entry = {event, class of the interval that ENDS at it, -1 for a
// fresh start}).  Anything else put on the stream (API copies / memsets) breaks the chain so that it is not billed
// to the next kernel.  (Two events per launch cost 3.5 % of a 512^3 volume: 9 000 launches.)
struct ProfRec {
    int kclass;
    hipEvent_t ev;
};

struct boa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 256;
    hipEvent_t t0[8] = {}, t1[8] = {};
    // per-kernel-class event profiling
    bool prof = false;
    bool prof_break = true;  // the next timed launch needs its own start event
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[BOA_K_COUNT] = {};
    long long prof_launches[BOA_K_COUNT] = {};
    double prof_flops[BOA_K_COUNT] = {};
    double prof_bytes[BOA_K_COUNT] = {};
    long long counters[BOA_CNT_COUNT] = {};  // launches per kernel variant (boa_debug_counter)
    // k_conv_ws run tables (first tile + length of every virtual workgroup's run), one per distinct layer geometry,
    // built on first use and kept for the life of the context: (key, device pointer)
    std::vector<std::pair<std::vector<int>, void*>> ws_runs;
    std::vector<std::vector<int>> ws_runs_host;   // parallel to ws_runs: the host copy of a run table (empty for descriptor tables)
    size_t ws_desc_bytes = 0;                     // bytes of the per-launch descriptor tables (key[0] == -1); boa_trim releases them
    // boa_malloc / boa_free: stream-ordered caching allocator for the transient volume-sized buffers of the host code
    // (a freed block goes back to `pool_free` without a device synchronisation and is handed out again for a request of
    // about its size; every use of a block is enqueued on `stream`, so reuse is ordered).  Network activations and weight
    // arenas use boa_malloc_raw / hipFree and never enter the pool.
    std::recursive_mutex pool_mu;             // the pool may be entered from several host threads (Python finalisers, two-lane runs)
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    size_t pool_bytes = 0;          // bytes parked in pool_free
    size_t pool_cap = 0;            // 0 = not initialised (BOA_POOL_GB, default 48; 0 disables pooling)
    // stash of the fused sliding-window head (boa_net_predict_labels_fold): the last decoder activation of every tile of the
    // current volume; grow-only, shared by the context's networks (they run one after the other on the stream)
    void* stash = nullptr;
    size_t stash_bytes = 0;
    bool stash_busy = false;        // a tile loop is filling it (boa_trim leaves it alone)
    // activation arena shared by the context's networks (net.hip: net_bind_arena); act_gen counts re-allocations
    void* act_arena = nullptr;
    size_t act_bytes = 0;
    unsigned long long act_gen = 0;
};
int boa_malloc_raw(boa_ctx* c, size_t bytes, void** dev_out);

// RAII-less explicit bracket: KernelTimer t(ctx, klass, flops, bytes); <launch>; t.stop();
struct KernelTimer {
    boa_ctx* ctx;
    int k;
    bool on = false;
    KernelTimer(boa_ctx* c, int kclass, double flops, double bytes);
    void stop();
};
int boa_prof_flush(boa_ctx* ctx);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// linear index -> (o0, o1, o2) of a [.][d1][d2] array.  Volumes here stay below 2^32 voxels, where two 32-bit divisions do
// what three 64-bit ones (each ~10x the instructions) did in the element-wise kernels.
__device__ __forceinline__ void idx3(size_t i, int d1, int d2, int& o0, int& o1, int& o2) {
    if (i <= 0xffffffffull) {
        const unsigned u = (unsigned)i, r = u / (unsigned)d2;
        o2 = (int)(u - r * (unsigned)d2);
        o0 = (int)(r / (unsigned)d1);
        o1 = (int)(r - (unsigned)o0 * (unsigned)d1);
    } else {
        const size_t r = i / (size_t)d2;
        o2 = (int)(i - r * (size_t)d2);
        o0 = (int)(r / (size_t)d1);
        o1 = (int)(r - (size_t)o0 * (size_t)d1);
    }
}

// ---- device helpers -------------------------------------------------------------------------------
__device__ __forceinline__ float h2f(__half h) { return __half2float(h); }
__device__ __forceinline__ __half f2h(float f) { return __float2half_rn(f); }
__device__ __forceinline__ float us2f(unsigned short u) { return __half2float(__ushort_as_half(u)); }
__device__ __forceinline__ unsigned short f2us(float f) { return __half_as_ushort(__float2half_rn(f)); }
