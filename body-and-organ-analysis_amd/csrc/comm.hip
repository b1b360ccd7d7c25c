// RCCL transport of the tile-sharded sliding window (several GPUs share one volume): the collectives are issued from
// the C ABI on HIP streams of the engine -- a communication stream per boa_comm, ordered against the context's compute stream by
// events, so that the slab exchange of model k runs under the tile loop of model k + 1 and nothing synchronises the host.
// The overlap slabs travel straight out of / into the fp16 accumulator planes: in the planar [C][X][Y][Z] layout the planes
// [lo, hi) of axis 0 are one contiguous run per class, so a boundary is (C + 1) ncclSend / ncclRecv calls inside one group --
// no packing kernel, no staging buffer.  (Reference: the accumulate loop NN/inference/predict_from_raw_data.py:611-614 is
// sequential on one device; what is exchanged here are its partial sums, see boa_hip/tile_shard.py for the ordering argument.)
//
// librccl is resolved at run time (dlopen): the library has no link-time dependency on it, single-GPU processes never load it,
// and a process that already carries a librccl (PyTorch bundles one) keeps using that copy.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <string>

#include "common.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("BOA_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so"};
        if (env && *env) r.h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        for (const char* n : names)      // a copy this process already loaded (e.g. PyTorch's)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.h) return;
#define BOA_SYM(field, name) r.field = (decltype(r.field))dlsym(r.h, name)
        BOA_SYM(GetUniqueId, "ncclGetUniqueId");
        BOA_SYM(CommInitRank, "ncclCommInitRank");
        BOA_SYM(CommDestroy, "ncclCommDestroy");
        BOA_SYM(Send, "ncclSend");
        BOA_SYM(Recv, "ncclRecv");
        BOA_SYM(AllReduce, "ncclAllReduce");
        BOA_SYM(GroupStart, "ncclGroupStart");
        BOA_SYM(GroupEnd, "ncclGroupEnd");
        BOA_SYM(GetErrorString, "ncclGetErrorString");
#undef BOA_SYM
        Dl_info info;
        if (r.GetUniqueId && dladdr((void*)r.GetUniqueId, &info) && info.dli_fname) r.path = info.dli_fname;
        if (!(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.AllReduce && r.GroupStart && r.GroupEnd)) {
            dlclose(r.h);
            r.h = nullptr;
        }
    });
    return r.h ? &r : nullptr;
}

}  // namespace

struct boa_comm {
    boa_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t stream = nullptr;   // communication stream
    hipEvent_t ev_in = nullptr;     // recorded on the compute stream: "everything the exchange reads / overwrites is ready"
    hipEvent_t ev_out = nullptr;    // recorded on the communication stream after the last queued exchange
    bool pending = false;           // ev_out has been recorded and not yet waited for
    long long calls = 0, bytes = 0;
};

#define BOA_NCCL_TRY(expr)                                                                                      \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            boa_set_error("%s failed: %s (%s:%d)", #expr, R->GetErrorString ? R->GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return BOA_EHIP;                                                                                    \
        }                                                                                                       \
    } while (0)

// Inside an open ncclGroupStart / ncclGroupEnd pair a failing call must close the group before returning: an open group on this
// thread would swallow every later RCCL call of the process (the other ranks' matching collectives then hang).
#define BOA_NCCL_GROUP_TRY(expr)                                                                                \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            (void)R->GroupEnd();                                                                                \
            (void)fence_out(c);   /* ev_out stays recordable: a later boa_comm_wait must not wait on a stale event */ \
            boa_set_error("%s failed: %s (%s:%d)", #expr, R->GetErrorString ? R->GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return BOA_EHIP;                                                                                    \
        }                                                                                                       \
    } while (0)

// Messages per direction in one ncclGroup of the point-to-point exchanges ($BOA_COMM_GROUP, default 8; 0 = everything in one group).
// Every sub-group holds the sends AND the receives of the same piece indices, so the pairs of a sub-group match on both sides of every
// boundary and the ranks walk the sub-groups in the same order: a line of neighbour exchanges has no cycle, whatever the group size.
static int comm_group_msgs() {
    static const int g = getenv("BOA_COMM_GROUP") ? atoi(getenv("BOA_COMM_GROUP")) : 8;
    return g > 0 ? g : (1 << 30);
}

extern "C" int boa_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" const char* boa_comm_library(void) {
    Rccl* R = rccl();
    return R ? R->path.c_str() : "";
}

extern "C" int boa_comm_unique_id(unsigned char id_out[128]) {
    BOA_REQUIRE(id_out, "boa_comm_unique_id: NULL argument");
    Rccl* R = rccl();
    BOA_REQUIRE(R, "boa_comm: librccl.so not found (set BOA_RCCL_LIB)");
    ncclUniqueId id;
    BOA_NCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, 128);
    return BOA_OK;
}

extern "C" int boa_comm_create(boa_ctx* ctx, int world, int rank, const unsigned char id_in[128], boa_comm** out) {
    BOA_REQUIRE(ctx && id_in && out && world >= 1 && rank >= 0 && rank < world, "boa_comm_create: bad argument (world %d, rank %d)", world, rank);
    Rccl* R = rccl();
    BOA_REQUIRE(R, "boa_comm: librccl.so not found (set BOA_RCCL_LIB)");
    BOA_HIP_TRY(hipSetDevice(ctx->device));
    boa_comm* c = new boa_comm;
    c->ctx = ctx;
    c->world = world;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id_in, 128);
    ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        boa_set_error("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, R->GetErrorString ? R->GetErrorString(r) : "?");
        delete c;
        return BOA_EHIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
        boa_set_error("boa_comm_create: stream / event creation failed");
        R->CommDestroy(c->comm);
        delete c;
        return BOA_EHIP;
    }
    *out = c;
    return BOA_OK;
}

extern "C" void boa_comm_destroy(boa_comm* c) {
    if (!c) return;
    Rccl* R = rccl();
    if (c->stream) hipStreamSynchronize(c->stream);
    if (R && c->comm) R->CommDestroy(c->comm);
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_out) hipEventDestroy(c->ev_out);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// the communication stream may start once everything queued on the compute stream so far has finished
static int fence_in(boa_comm* c) {
    BOA_HIP_TRY(hipEventRecord(c->ev_in, c->ctx->stream));
    BOA_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    return BOA_OK;
}

static int fence_out(boa_comm* c) {
    BOA_HIP_TRY(hipEventRecord(c->ev_out, c->stream));
    c->pending = true;
    c->ctx->prof_break = true;
    return BOA_OK;
}

// Later work on the compute stream waits for every exchange queued so far (a stream-side wait: the host does not block).
extern "C" int boa_comm_wait(boa_comm* c) {
    BOA_REQUIRE(c, "boa_comm_wait: NULL argument");
    if (c->pending) {
        BOA_HIP_TRY(hipStreamWaitEvent(c->ctx->stream, c->ev_out, 0));
        c->pending = false;
        c->ctx->prof_break = true;
    }
    return BOA_OK;
}

static ncclDataType_t nccl_type(int dtype) {
    switch (dtype) {
        case 0: return ncclUint8;
        case 1: return ncclFloat16;
        case 2: return ncclInt32;
        default: return ncclFloat32;
    }
}
static size_t type_bytes(int dtype) { return dtype == 0 ? 1 : (dtype == 1 ? 2 : 4); }

// One grouped point-to-point exchange: n_send contiguous pieces to rank `dst` (skipped when dst < 0), n_recv pieces from rank
// `src` (skipped when src < 0); pieces are matched in order.  Queued on the communication stream after everything queued on the
// compute stream so far; boa_comm_wait orders later compute work behind it.
extern "C" int boa_comm_exchange(boa_comm* c, int dst, const void* const* send_ptrs, const size_t* send_bytes, int n_send, int src,
                                 void* const* recv_ptrs, const size_t* recv_bytes, int n_recv) {
    BOA_REQUIRE(c && n_send >= 0 && n_recv >= 0 && dst < c->world && src < c->world,
                "boa_comm_exchange: bad argument");
    BOA_REQUIRE(dst < 0 || n_send == 0 || (send_ptrs && send_bytes), "boa_comm_exchange: NULL send arrays");
    BOA_REQUIRE(src < 0 || n_recv == 0 || (recv_ptrs && recv_bytes), "boa_comm_exchange: NULL receive arrays");
    for (int i = 0; dst >= 0 && i < n_send; ++i) BOA_REQUIRE(send_ptrs[i] || send_bytes[i] == 0, "boa_comm_exchange: send piece %d is NULL", i);
    for (int i = 0; src >= 0 && i < n_recv; ++i) BOA_REQUIRE(recv_ptrs[i] || recv_bytes[i] == 0, "boa_comm_exchange: receive piece %d is NULL", i);
    Rccl* R = rccl();
    BOA_TRY(fence_in(c));
    const int ns = dst >= 0 ? n_send : 0, nr = src >= 0 ? n_recv : 0, G = comm_group_msgs();
    for (int g0 = 0; g0 < std::max(ns, nr); g0 += std::min(G, std::max(ns, nr))) {
        const int g1 = g0 + std::min(G, std::max(ns, nr) - g0);
        BOA_NCCL_TRY(R->GroupStart());
        for (int i = g0; i < std::min(g1, ns); ++i) {
            BOA_NCCL_GROUP_TRY(R->Send(send_ptrs[i], send_bytes[i], ncclUint8, dst, c->comm, c->stream));
            c->bytes += (long long)send_bytes[i];
        }
        for (int i = g0; i < std::min(g1, nr); ++i) BOA_NCCL_GROUP_TRY(R->Recv(recv_ptrs[i], recv_bytes[i], ncclUint8, src, c->comm, c->stream));
        BOA_NCCL_TRY(R->GroupEnd());
    }
    c->calls++;
    return fence_out(c);
}

// Plane ranges of fp16 logits [C][PV0][PV1][PV2] to / from several peers, in place (the reduce-scatter of plane-disjoint fold logits:
// every plane travels once, to the rank that finalises it).  Sub-groups of whole classes; their size is a function of $BOA_COMM_GROUP
// and the world size only, so every rank opens and closes the same groups in the same order whatever its own message lists hold.
extern "C" int boa_comm_planes_to_owner(boa_comm* c, uint16_t* logits, int C, const int PV[3], int n_send, const int* send_peer,
                                        const int* send_lo, const int* send_hi, int n_recv, const int* recv_peer, const int* recv_lo,
                                        const int* recv_hi) {
    BOA_REQUIRE(c && logits && PV && C >= 1 && n_send >= 0 && n_recv >= 0, "boa_comm_planes_to_owner: bad argument");
    BOA_REQUIRE(n_send == 0 || (send_peer && send_lo && send_hi), "boa_comm_planes_to_owner: NULL send arrays");
    BOA_REQUIRE(n_recv == 0 || (recv_peer && recv_lo && recv_hi), "boa_comm_planes_to_owner: NULL receive arrays");
    for (int i = 0; i < n_send; ++i)
        BOA_REQUIRE(send_peer[i] >= 0 && send_peer[i] < c->world && send_peer[i] != c->rank && send_lo[i] >= 0 && send_lo[i] < send_hi[i] && send_hi[i] <= PV[0],
                    "boa_comm_planes_to_owner: send %d: planes [%d, %d) to rank %d", i, send_lo[i], send_hi[i], send_peer[i]);
    for (int i = 0; i < n_recv; ++i)
        BOA_REQUIRE(recv_peer[i] >= 0 && recv_peer[i] < c->world && recv_peer[i] != c->rank && recv_lo[i] >= 0 && recv_lo[i] < recv_hi[i] && recv_hi[i] <= PV[0],
                    "boa_comm_planes_to_owner: receive %d: planes [%d, %d) from rank %d", i, recv_lo[i], recv_hi[i], recv_peer[i]);
    Rccl* R = rccl();
    const size_t plane = (size_t)PV[1] * PV[2], vv = (size_t)PV[0] * plane;
    BOA_TRY(fence_in(c));
    const int per = std::max(1, comm_group_msgs() / std::max(1, c->world));   // classes per RCCL group
    for (int k0 = 0; k0 < C; k0 += per) {
        const int k1 = std::min(C, k0 + per);
        BOA_NCCL_TRY(R->GroupStart());
        for (int k = k0; k < k1; ++k) {
            for (int i = 0; i < n_send; ++i) {
                const size_t n = (size_t)(send_hi[i] - send_lo[i]) * plane;
                BOA_NCCL_GROUP_TRY(R->Send(logits + (size_t)k * vv + (size_t)send_lo[i] * plane, n, ncclFloat16, send_peer[i], c->comm, c->stream));
                c->bytes += (long long)(n * 2);
            }
            for (int i = 0; i < n_recv; ++i)
                BOA_NCCL_GROUP_TRY(R->Recv(logits + (size_t)k * vv + (size_t)recv_lo[i] * plane, (size_t)(recv_hi[i] - recv_lo[i]) * plane, ncclFloat16,
                                           recv_peer[i], c->comm, c->stream));
        }
        BOA_NCCL_TRY(R->GroupEnd());
    }
    c->calls++;
    return fence_out(c);
}

// The overlap slab of a tile-row boundary: planes [lo, hi) of the C class planes of `acc` and of `nacc` (fp16, planar
// [.][PV0][PV1][PV2]).  send: this rank's finished partial sums -> rank dst; recv: the lower rank's -> straight into this rank's
// planes (exact mode: nothing of this rank's has been added there yet) or into `recv_stage` ((C + 1) x planes x PV1 PV2 halves,
// all-reduce mode: boa_add_f16_planes adds them afterwards).  Either direction may be absent (dst / src < 0).
extern "C" int boa_comm_shift_slab(boa_comm* c, int dst, int send_lo, int send_hi, int src, int recv_lo, int recv_hi, uint16_t* acc,
                                   uint16_t* nacc, int C, const int PV[3], uint16_t* recv_stage) {
    BOA_REQUIRE(c && acc && nacc && PV && C >= 1, "boa_comm_shift_slab: NULL argument");
    Rccl* R = rccl();
    const size_t plane = (size_t)PV[1] * PV[2], vv = (size_t)PV[0] * plane;
    BOA_REQUIRE(dst < 0 || (send_lo >= 0 && send_lo < send_hi && send_hi <= PV[0]), "boa_comm_shift_slab: send planes [%d, %d)", send_lo, send_hi);
    BOA_REQUIRE(src < 0 || (recv_lo >= 0 && recv_lo < recv_hi && recv_hi <= PV[0]), "boa_comm_shift_slab: recv planes [%d, %d)", recv_lo, recv_hi);
    BOA_TRY(fence_in(c));
    const size_t n_s = dst >= 0 ? (size_t)(send_hi - send_lo) * plane : 0, n_r = src >= 0 ? (size_t)(recv_hi - recv_lo) * plane : 0;
    const int G = comm_group_msgs();
    for (int k0 = 0; k0 <= C; k0 += std::min(G, C + 1)) {   // plane k = class k, plane C = the weight plane n
        const int k1 = std::min(C + 1, k0 + std::min(G, C + 1));
        BOA_NCCL_TRY(R->GroupStart());
        for (int k = k0; dst >= 0 && k < k1; ++k) {
            const uint16_t* p = k < C ? acc + (size_t)k * vv + (size_t)send_lo * plane : nacc + (size_t)send_lo * plane;
            BOA_NCCL_GROUP_TRY(R->Send(p, n_s, ncclFloat16, dst, c->comm, c->stream));
        }
        for (int k = k0; src >= 0 && k < k1; ++k) {
            uint16_t* p = recv_stage ? recv_stage + (size_t)k * n_r : (k < C ? acc + (size_t)k * vv + (size_t)recv_lo * plane : nacc + (size_t)recv_lo * plane);
            BOA_NCCL_GROUP_TRY(R->Recv(p, n_r, ncclFloat16, src, c->comm, c->stream));
        }
        BOA_NCCL_TRY(R->GroupEnd());
    }
    c->bytes += (long long)(n_s * 2 * (C + 1));
    c->calls++;
    return fence_out(c);
}

// In-place sum over all ranks (label volumes with disjoint supports, flags, plane-disjoint logits).  dtype: 0 uint8, 1 fp16,
// 2 int32, 3 fp32.
extern "C" int boa_comm_all_reduce(boa_comm* c, void* dev, size_t count, int dtype) {
    BOA_REQUIRE(c && dev && dtype >= 0 && dtype <= 3, "boa_comm_all_reduce: bad argument");
    Rccl* R = rccl();
    BOA_TRY(fence_in(c));
    BOA_NCCL_TRY(R->AllReduce(dev, dev, count, nccl_type(dtype), ncclSum, c->comm, c->stream));
    c->calls++;
    c->bytes += (long long)(count * type_bytes(dtype));
    return fence_out(c);
}

extern "C" int boa_comm_stats(boa_comm* c, long long* calls, long long* bytes) {
    BOA_REQUIRE(c, "boa_comm_stats: NULL argument");
    if (calls) *calls = c->calls;
    if (bytes) *bytes = c->bytes;
    return BOA_OK;
}

// acc[k][lo:hi] = half(float(acc[k][lo:hi]) + float(stage[k])) for the C class planes and n: the "allreduce" mode's P + Q on the
// rank that owns the slab (one RTNE rounding per element, what a two-rank fp16 sum all-reduce computes)
__global__ __launch_bounds__(256) void k_add_f16_planes(uint16_t* __restrict__ acc, uint16_t* __restrict__ nacc, const uint16_t* __restrict__ stage, int C,
                                                        size_t vv, size_t off, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = blockIdx.y;
    uint16_t* p = (k < C ? acc + (size_t)k * vv : nacc) + off + i;
    *p = f2us(us2f(*p) + us2f(stage[(size_t)k * n + i]));
}

extern "C" int boa_add_f16_planes(boa_ctx* ctx, uint16_t* acc, uint16_t* nacc, const uint16_t* stage, int C, const int PV[3], int lo, int hi) {
    BOA_REQUIRE(ctx && acc && nacc && stage && PV && lo >= 0 && lo < hi && hi <= PV[0], "boa_add_f16_planes: bad argument");
    const size_t plane = (size_t)PV[1] * PV[2], n = (size_t)(hi - lo) * plane;
    KernelTimer tm(ctx, BOA_K_OTHER, 0, 6.0 * (double)n * (C + 1));
    hipLaunchKernelGGL(k_add_f16_planes, dim3((unsigned)((n + 255) / 256), C + 1), dim3(256), 0, ctx->stream, acc, nacc, stage, C, (size_t)PV[0] * plane,
                       (size_t)lo * plane, n);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
