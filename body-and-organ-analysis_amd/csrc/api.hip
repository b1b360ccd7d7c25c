// Context, memory, timing and error plumbing of libboa_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <algorithm>
#include <iterator>
#include <string.h>

#include "common.h"

static thread_local char g_err[1024] = "";

void boa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* boa_last_error(void) { return g_err; }
extern "C" int boa_version(void) { return 100; }

extern "C" int boa_init(int device, void* stream, boa_ctx** out) {
    BOA_REQUIRE(out != nullptr, "boa_init: out is NULL");
    int n = 0;
    BOA_HIP_TRY(hipGetDeviceCount(&n));
    BOA_REQUIRE(device >= 0 && device < n, "boa_init: device %d not available (%d visible)", device, n);
    BOA_HIP_TRY(hipSetDevice(device));
    boa_ctx* c = new boa_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        BOA_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    hipDeviceProp_t p;
    BOA_HIP_TRY(hipGetDeviceProperties(&p, device));
    c->cu_count = p.multiProcessorCount;
    for (int i = 0; i < 8; ++i) {
        BOA_HIP_TRY(hipEventCreate(&c->t0[i]));
        BOA_HIP_TRY(hipEventCreate(&c->t1[i]));
    }
    *out = c;
    return BOA_OK;
}

extern "C" void boa_destroy(boa_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (int i = 0; i < 8; ++i) {
        hipEventDestroy(c->t0[i]);
        hipEventDestroy(c->t1[i]);
    }
    for (auto& r : c->prof_pending) hipEventDestroy(r.ev);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    for (auto& r : c->ws_runs) hipFree(r.second);
    for (auto& b : c->pool_free) hipFree(b.second);
    for (auto& b : c->pool_live) hipFree(b.first);  // buffers the caller never freed
    if (c->stash) hipFree(c->stash);
    if (c->act_arena) hipFree(c->act_arena);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int boa_device_info(boa_ctx* c, char* name, int name_len, int* cu_count, size_t* total_mem,
                               size_t* free_mem) {
    BOA_REQUIRE(c, "ctx is NULL");
    hipDeviceProp_t p;
    BOA_HIP_TRY(hipGetDeviceProperties(&p, c->device));
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    size_t f = 0, t = 0;
    BOA_HIP_TRY(hipMemGetInfo(&f, &t));
    if (total_mem) *total_mem = t;
    if (free_mem) *free_mem = f;
    return BOA_OK;
}

int boa_malloc_raw(boa_ctx* c, size_t bytes, void** dev_out) {
    BOA_REQUIRE(c && dev_out, "boa_malloc: NULL argument");
    BOA_HIP_TRY(hipSetDevice(c->device));
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess && (c->pool_bytes || (c->stash && !c->stash_busy))) {  // give the parked blocks (and an idle tile stash) back and try once more
        (void)hipGetLastError();
        boa_trim(c);
        e = hipMalloc(&p, bytes ? bytes : 1);
    }
    if (e != hipSuccess) {
        boa_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return BOA_ENOMEM;
    }
    *dev_out = p;
    return BOA_OK;
}

static size_t pool_round(size_t bytes) {
    if (bytes <= (1u << 20)) return (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
    return (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
}

extern "C" int boa_malloc(boa_ctx* c, size_t bytes, void** dev_out) {
    BOA_REQUIRE(c && dev_out, "boa_malloc: NULL argument");
    if (c->pool_cap == 0) {
        // cap of the parked bytes: $BOA_POOL_GB, else 48 GB but never more than a sixth of the device's memory (torch / RCCL
        // allocations of the same process cannot reclaim parked blocks: the sharded paths call boa_trim before collectives)
        const char* e = getenv("BOA_POOL_GB");
        double gb = e ? atof(e) : 48.0;
        if (!e) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) gb = std::min(gb, (double)tot / 6.0 / (double)(1ull << 30));
        }
        c->pool_cap = gb > 0 ? (size_t)(gb * (double)(1ull << 30)) : 1;  // 1 byte: nothing is ever parked
    }
    const size_t sz = pool_round(bytes);
    std::lock_guard<std::recursive_mutex> lock(c->pool_mu);
    auto it = c->pool_free.lower_bound(sz);
    if (it != c->pool_free.end() && it->first - sz <= std::max<size_t>(sz / 4, 2u << 20)) {
        *dev_out = it->second;
        c->pool_bytes -= it->first;
        c->pool_live[it->second] = it->first;
        c->pool_free.erase(it);
        return BOA_OK;
    }
    BOA_TRY(boa_malloc_raw(c, sz, dev_out));
    c->pool_live[*dev_out] = sz;
    return BOA_OK;
}

extern "C" int boa_free(boa_ctx* c, void* dev) {
    BOA_REQUIRE(c, "ctx is NULL");
    if (!dev) return BOA_OK;
    std::lock_guard<std::recursive_mutex> lock(c->pool_mu);
    auto it = c->pool_live.find(dev);
    if (it == c->pool_live.end()) {  // not one of ours (or allocated raw): synchronise, then release
        BOA_HIP_TRY(hipStreamSynchronize(c->stream));
        BOA_HIP_TRY(hipFree(dev));
        return BOA_OK;
    }
    const size_t sz = it->second;
    c->pool_live.erase(it);
    c->pool_free.emplace(sz, dev);
    c->pool_bytes += sz;
    if (c->pool_bytes > c->pool_cap) {  // over the cap: release the largest parked blocks
        BOA_HIP_TRY(hipStreamSynchronize(c->stream));
        while (c->pool_bytes > c->pool_cap && !c->pool_free.empty()) {
            auto last = std::prev(c->pool_free.end());
            BOA_HIP_TRY(hipFree(last->second));
            c->pool_bytes -= last->first;
            c->pool_free.erase(last);
        }
    }
    return BOA_OK;
}

static int trim_locked(boa_ctx* c);
extern "C" int boa_trim(boa_ctx* c) {
    BOA_REQUIRE(c, "ctx is NULL");
    std::lock_guard<std::recursive_mutex> lock(c->pool_mu);
    return trim_locked(c);
}

extern "C" int boa_bind_thread(boa_ctx* c) {
    BOA_REQUIRE(c, "ctx is NULL");
    BOA_HIP_TRY(hipSetDevice(c->device));
    return BOA_OK;
}

static int trim_locked(boa_ctx* c) {
    // the gather head's tile stash (up to tens of GB, grow-only between trims) goes back too: a trim means somebody needs room --
    // the other lane's context, torch / RCCL buffers, a post-processing volume whose hipMalloc failed -- and the next volume's
    // tile loop re-allocates it (or falls back to the scatter form if it no longer fits)
    const bool drop_stash = c->stash && !c->stash_busy;
    if (c->pool_free.empty() && !drop_stash && c->ws_desc_bytes == 0) return BOA_OK;
    BOA_HIP_TRY(hipStreamSynchronize(c->stream));
    // the conv kernels' tile descriptor tables (one per layer geometry, batch and grid: the tail batches of differently sized volumes keep
    // adding some) are rebuilt on their next use -- one small kernel each; the run tables (per layer geometry only) stay
    for (size_t i = c->ws_runs.size(); i-- > 0;)
        if (!c->ws_runs[i].first.empty() && c->ws_runs[i].first[0] == -1) {
            hipFree(c->ws_runs[i].second);
            c->ws_runs.erase(c->ws_runs.begin() + i);
            c->ws_runs_host.erase(c->ws_runs_host.begin() + i);
        }
    c->ws_desc_bytes = 0;
    for (auto& b : c->pool_free) hipFree(b.second);
    c->pool_free.clear();
    c->pool_bytes = 0;
    if (drop_stash) {
        hipFree(c->stash);
        c->stash = nullptr;
        c->stash_bytes = 0;
    }
    return BOA_OK;
}

extern "C" int boa_memset(boa_ctx* c, void* dev, int value, size_t bytes) {
    BOA_REQUIRE(c && dev, "boa_memset: NULL argument");
    KernelTimer t(c, BOA_K_OTHER, 0, (double)bytes);  // accumulator zeroing is part of a volume's kernel time
    BOA_HIP_TRY(hipMemsetAsync(dev, value, bytes, c->stream));
    t.stop();
    return BOA_OK;
}

extern "C" int boa_h2d(boa_ctx* c, void* dev_dst, const void* host_src, size_t bytes) {
    BOA_REQUIRE(c && dev_dst && host_src, "boa_h2d: NULL argument");
    c->prof_break = true;
    BOA_HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    BOA_HIP_TRY(hipStreamSynchronize(c->stream));
    return BOA_OK;
}

extern "C" int boa_d2h(boa_ctx* c, void* host_dst, const void* dev_src, size_t bytes) {
    BOA_REQUIRE(c && host_dst && dev_src, "boa_d2h: NULL argument");
    c->prof_break = true;
    BOA_HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
    BOA_HIP_TRY(hipStreamSynchronize(c->stream));
    return BOA_OK;
}

extern "C" int boa_host_alloc(boa_ctx* c, size_t bytes, void** host_out) {
    BOA_REQUIRE(c && host_out, "boa_host_alloc: NULL argument");
    BOA_HIP_TRY(hipSetDevice(c->device));
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        boa_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return BOA_ENOMEM;
    }
    *host_out = p;
    return BOA_OK;
}

extern "C" int boa_host_free(boa_ctx*, void* host) {
    if (!host) return BOA_OK;
    BOA_HIP_TRY(hipHostFree(host));
    return BOA_OK;
}

extern "C" int boa_sync(boa_ctx* c) {
    BOA_REQUIRE(c, "ctx is NULL");
    c->prof_break = true;
    BOA_HIP_TRY(hipStreamSynchronize(c->stream));
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

extern "C" int boa_timer_start(boa_ctx* c, int slot) {
    BOA_REQUIRE(c && slot >= 0 && slot < 8, "boa_timer_start: bad slot");
    BOA_HIP_TRY(hipEventRecord(c->t0[slot], c->stream));
    return BOA_OK;
}

extern "C" int boa_timer_stop(boa_ctx* c, int slot, float* ms_out) {
    BOA_REQUIRE(c && slot >= 0 && slot < 8 && ms_out, "boa_timer_stop: bad argument");
    BOA_HIP_TRY(hipEventRecord(c->t1[slot], c->stream));
    BOA_HIP_TRY(hipEventSynchronize(c->t1[slot]));
    BOA_HIP_TRY(hipEventElapsedTime(ms_out, c->t0[slot], c->t1[slot]));
    return BOA_OK;
}

// ---- per-kernel-class profiling with HIP events on the launch stream --------------------------------
static hipEvent_t take_event(boa_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

KernelTimer::KernelTimer(boa_ctx* c, int kclass, double flops, double bytes) : ctx(c), k(kclass) {
    if (!c->prof) return;
    if (c->prof_pending.size() >= 32768) boa_prof_flush(c);
    on = true;
    c->prof_flops[k] += flops;
    c->prof_bytes[k] += bytes;
    c->prof_launches[k] += 1;
    if (c->prof_break || c->prof_pending.empty()) {
        hipEvent_t e = take_event(c);
        hipEventRecord(e, c->stream);
        c->prof_pending.push_back({-1, e});
        c->prof_break = false;
    }
}

void KernelTimer::stop() {
    if (!on) return;
    hipEvent_t e = take_event(ctx);
    hipEventRecord(e, ctx->stream);
    ctx->prof_pending.push_back({k, e});
}

int boa_prof_flush(boa_ctx* c) {
    if (!c->prof_pending.empty()) hipEventSynchronize(c->prof_pending.back().ev);
    for (size_t i = 0; i < c->prof_pending.size(); ++i) {
        const ProfRec& r = c->prof_pending[i];
        float ms = 0.f;
        if (r.kclass >= 0 && i > 0 && hipEventElapsedTime(&ms, c->prof_pending[i - 1].ev, r.ev) == hipSuccess)
            c->prof_ms[r.kclass] += ms;
        c->ev_pool.push_back(r.ev);
    }
    c->prof_pending.clear();
    c->prof_break = true;
    return BOA_OK;
}

extern "C" long long boa_debug_counter(boa_ctx* c, int which, int reset) {
    if (!c || which < 0 || which >= BOA_CNT_COUNT) return -1;
    const long long v = c->counters[which];
    if (reset) c->counters[which] = 0;
    return v;
}

extern "C" int boa_prof_enable(boa_ctx* c, int on) {
    BOA_REQUIRE(c, "ctx is NULL");
    if (!on) boa_prof_flush(c);
    c->prof = on != 0;
    return BOA_OK;
}

extern "C" int boa_prof_reset(boa_ctx* c) {
    BOA_REQUIRE(c, "ctx is NULL");
    boa_prof_flush(c);
    for (int i = 0; i < BOA_K_COUNT; ++i) {
        c->prof_ms[i] = 0;
        c->prof_launches[i] = 0;
        c->prof_flops[i] = 0;
        c->prof_bytes[i] = 0;
    }
    return BOA_OK;
}

extern "C" int boa_prof_get(boa_ctx* c, int kclass, double* total_ms, long long* launches, double* flops,
                            double* bytes) {
    BOA_REQUIRE(c && kclass >= 0 && kclass < BOA_K_COUNT, "boa_prof_get: bad class");
    boa_prof_flush(c);
    if (total_ms) *total_ms = c->prof_ms[kclass];
    if (launches) *launches = c->prof_launches[kclass];
    if (flops) *flops = c->prof_flops[kclass];
    if (bytes) *bytes = c->prof_bytes[kclass];
    return BOA_OK;
}
