// Device-side pieces shared by the wave-specialised conv kernels (k_conv_ws in conv_ws.hip, k_conv_ns in conv_ns.hip): tile
// sequence (virtual workgroups / run tables), the producers' halo staging with the deferred InstanceNorm + LeakyReLU, scalar-base
// global access helpers.  See conv_ws.hip for the design notes.
#pragma once
#include "conv.h"
#ifndef WS_THREADS
#define WS_THREADS 512  // 256 consumer threads (4 waves) + WS_PROD producer threads
#endif
#define WS_PROD (WS_THREADS - 256)
#ifndef WS_DEFER_EPILOGUE
#define WS_DEFER_EPILOGUE 1
#endif
#define WS_WB ((27 * 64 + WS_PROD - 1) / WS_PROD)  // weight items per producer thread (taps * 64 items, <= 27 taps)
#define WS_TRACE_SLOTS 4096
// timeline stamps (debug only): slot = event counter of the calling wave; wave 0 (consumer) and wave 4 (producer) of block tr_blk
// (BOA_WS_TRACE=<block>: block 0 only sees tiles on the tensor's y = z = 0 edge, i.e. the producers' slow bounds-checked path)
// The production build compiles the stamps OUT (each site is ~6 instructions of a single-wave role loop that runs at ~12 cycles per
// instruction: A/B -1.9 % per forward without them); `tools/build_alt.sh trace -DWS_WITH_TRACE` + BOA_HIP_LIB builds the traced copy.
#ifndef WS_WITH_TRACE
#define WS_STAMP(code) do { } while (0)
#define WS_TRACING false
#else
#define WS_TRACING (p.trace != nullptr)
#define WS_STAMP(code)                                                                          \
    do {                                                                                        \
        if (p.trace && (int)blockIdx.x == tr_blk && lane == 0 && ((wave & 3) == 0) && tr_n < WS_TRACE_SLOTS / 2) { \
            p.trace[(producer ? WS_TRACE_SLOTS / 2 : 0) + tr_n] = ((unsigned long long)(code) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); \
            ++tr_n;                                                                             \
        }                                                                                       \
    } while (0)
#endif

// one k-octet plane of the halo: whole producer rounds of WS_PROD / 2 voxels (lanes past the halo store into the padding
// instead of being masked: no per-item exec juggling), + 64 bytes of bank skew between the two planes
__device__ __forceinline__ int ws_plane_bytes(int HV) { return ((HV + WS_PROD / 2 - 1) / (WS_PROD / 2)) * (WS_PROD / 2) * 16 + 64; }

struct TileCoord {
    int n, cy, ox0, oy0, oz0;
    int sp;  // spatial tile index
};

__device__ __forceinline__ TileCoord decode_tile(const ConvArgs& p, int t) {
    // default order: spatial tiles fastest (x fastest, then z, then y), then cout chunk, then n.  With the dominant halo
    // overlap along x (thin tiles in x), consecutive tiles share their x-halo planes.
    // p.cy_fast (2-chunk inputs with several cout chunks): cout chunk fastest -- consecutive tiles of a workgroup then
    // need the SAME staged halo (both 16-channel chunks stay in the two LDS buffers), only the weights change.
    TileCoord c;
    const int nsp = p.t0 * p.t1 * p.t2;
    const int ncy = p.ncy;
    if (p.cy_fast) {
        c.cy = t % ncy;
        const int r = t / ncy;
        c.sp = r % nsp;
        c.n = r / nsp;
    } else {
        c.sp = t % nsp;
        const int r = t / nsp;
        c.cy = r % ncy;
        c.n = r / ncy;
    }
    int bt = c.sp;
    const int tx = bt % p.t0;
    bt /= p.t0;
    const int tz = bt % p.t2;
    const int ty = bt / p.t2;
    c.ox0 = tx * p.b0 * p.w0;
    c.oy0 = ty * p.b1 * p.w1;
    c.oz0 = tz * p.b2 * p.w2;
    return c;
}

// next tile of a contiguous walk (tile index + 1) without divisions
__device__ __forceinline__ void next_tile(const ConvArgs& p, TileCoord& c) {
    const int e0 = p.b0 * p.w0, e1 = p.b1 * p.w1, e2 = p.b2 * p.w2;
    const int ncy = p.ncy;
    if (p.cy_fast) {
        if (++c.cy < ncy) return;
        c.cy = 0;
    }
    ++c.sp;
    c.ox0 += e0;
    if (c.ox0 >= p.t0 * e0) {
        c.ox0 = 0;
        c.oz0 += e2;
        if (c.oz0 >= p.t2 * e2) {
            c.oz0 = 0;
            c.oy0 += e1;
            if (c.oy0 >= p.t1 * e1) {
                c.oy0 = 0;
                c.sp = 0;
                if (p.cy_fast) {
                    ++c.n;
                } else if (++c.cy >= ncy) {
                    c.cy = 0;
                    ++c.n;
                }
            }
        }
    }
}

// Tile sequence of a workgroup.  The tiles of ONE sample (spatial tiles x cout chunks, in decode_tile's order) are cut
// into `vw` contiguous runs, one per VIRTUAL workgroup j of that sample (vw = min(tiles per sample, CUs): a function of the
// layer geometry only).  Virtual workgroup (n, j) accumulates the InstanceNorm partial sums of its run and writes them to
// statistics slot 4 j + wave of sample n, so the set of partial sums of a sample -- and with it the sample's (scale, shift)
// and every later activation -- does not depend on how many other samples share the launch (batch-invariant results: the
// same tile gives the same logits at tile batch 1, 4 or 8 and in a tile-sharded run).  A physical workgroup b executes the
// virtual workgroups b, b + G, b + 2G, ... one after the other; which physical workgroup runs a virtual one only affects
// speed.  Workgroup b is observed to run on XCD b % 8 (each XCD has a private 4 MiB L2): a sample's tile list is cut into 8
// contiguous ranges, one per XCD (j % 8 == b % 8), and the virtual workgroups of an XCD take contiguous runs of that range,
// so the halos an XCD re-reads are the ones its own L2 just fetched.  The runs (first tile's coordinates + length) are a
// host-built table per layer geometry (`ws_run_table`): starting a run costs a few scalar loads, no divisions.
struct TileSeq {
    int n, j;  // current virtual workgroup: sample, index within the sample (statistics slot = 4 j + wave)
    int left;  // tiles left in its run, the current one included
    TileCoord tc;
};

// position at the first tile of virtual workgroup (n, j): a table lookup (wave-uniform -> scalar loads), no divisions
__device__ __forceinline__ void seq_run(const ConvArgs& p, TileSeq& s) {
    const int* r = p.runs + s.j * 8;
    s.left = r[0];
    s.tc.n = s.n;
    s.tc.cy = r[1];
    s.tc.sp = r[2];
    s.tc.ox0 = r[3];
    s.tc.oy0 = r[4];
    s.tc.oz0 = r[5];
}

__device__ __forceinline__ void seq_first_of(const ConvArgs& p, TileSeq& s, int blk) {
    s.n = blk / p.vw;
    s.j = blk - s.n * p.vw;
    seq_run(p, s);
}
__device__ __forceinline__ void seq_first(const ConvArgs& p, TileSeq& s) { seq_first_of(p, s, (int)blockIdx.x); }

// next tile of this workgroup's sequence; true when it is the first tile of a new virtual workgroup
__device__ __forceinline__ bool seq_next(const ConvArgs& p, TileSeq& s) {
    if (--s.left > 0) {
        next_tile(p, s.tc);
        return false;
    }
    s.n += p.vstep_n;  // virtual workgroup v + G
    s.j += p.vstep_j;
    if (s.j >= p.vw) {
        s.j -= p.vw;
        ++s.n;
    }
    seq_run(p, s);
    return true;
}

// ---- tile descriptors (WS_DESC) -------------------------------------------------------------------------
// The walk above costs each role ~700 .. 1 100 cycles per tile of dependent scalar work (kernel-argument and run-table loads with
// their waits, SGPR spills through v_readlane: round-5 trace of an interior workgroup), on the consumers' critical path in a tile's
// first interval.  Its result is a pure function of (layer geometry, batch, grid), so it is run ONCE per such key by k_ws_build_desc
// (the same seq_first / seq_next code, one thread per physical workgroup) into a table the roles read with one scalar load per tile:
//   row b (desc_row entries of 8 ints): entry 0 = {tiles of workgroup b, ...}, entry 1 + k = descriptor of its k-th tile
//   descriptor = {n, cy, ox0, oy0, oz0, flags, vo, ibase}
//     flags  bit 0 first tile of a virtual workgroup (statistics slot changes), bit 1 output tile inside the tensor, bit 2 same
//            spatial tile as the previous tile (halo reuse, cy_fast), bits 4-5 halo class (0 inside the tensor, 1 out by one
//            voxel layer on the faces of bits 8-13, 2 general), bits 16.. virtual workgroup index j
//     vo     (ox0 * Ho + oy0) * Wo + oz0;  ibase  (ix0 * Hi + iy0) * Wi + iz0 of the halo origin
#define WS_DF_NEWRUN 1
#define WS_DF_FULL 2
#define WS_DF_REUSE 4

struct TileDesc {
    TileCoord tc;
    int flags, vo, ibase;
};

__device__ __forceinline__ TileDesc make_desc(const ConvArgs& p, const TileSeq& s, bool new_run) {
    TileDesc d;
    d.tc = s.tc;
    const TileCoord& tc = s.tc;
    int fl = new_run ? WS_DF_NEWRUN : 0;
    if (tc.ox0 + p.b0 * p.w0 <= p.Do && tc.oy0 + p.b1 * p.w1 <= p.Ho && tc.oz0 + p.b2 * p.w2 <= p.Wo) fl |= WS_DF_FULL;
    if (p.cy_fast && tc.cy != 0 && !new_run) fl |= WS_DF_REUSE;
    const int ix0 = tc.ox0 * p.s0 - p.p0, iy0 = tc.oy0 * p.s1 - p.p1, iz0 = tc.oz0 * p.s2 - p.p2;
    if (ix0 >= 0 && iy0 >= 0 && iz0 >= 0 && ix0 + p.h0 <= p.Di && iy0 + p.h1 <= p.Hi && iz0 + p.h2 <= p.Wi) {
    } else if (ix0 >= -1 && iy0 >= -1 && iz0 >= -1 && ix0 + p.h0 <= p.Di + 1 && iy0 + p.h1 <= p.Hi + 1 && iz0 + p.h2 <= p.Wi + 1) {
        fl |= 1 << 4;
        fl |= (ix0 < 0 ? 1 : 0) << 8 | (ix0 + p.h0 > p.Di ? 1 : 0) << 9 | (iy0 < 0 ? 1 : 0) << 10 | (iy0 + p.h1 > p.Hi ? 1 : 0) << 11 |
              (iz0 < 0 ? 1 : 0) << 12 | (iz0 + p.h2 > p.Wi ? 1 : 0) << 13;
    } else {
        fl |= 2 << 4;
    }
    fl |= s.j << 16;
    d.flags = fl;
    d.vo = (tc.ox0 * p.Ho + tc.oy0) * p.Wo + tc.oz0;
    d.ibase = (ix0 * p.Hi + iy0) * p.Wi + iz0;
    return d;
}

// one descriptor = two 16-byte scalar loads (desc is a __restrict__ const kernel argument and the address is wave-uniform)
__device__ __forceinline__ TileDesc load_desc(const int* __restrict__ row, int k) {
    const int4 a = *(const int4*)(row + (size_t)(k + 1) * 8), b = *(const int4*)(row + (size_t)(k + 1) * 8 + 4);
    TileDesc d;
#define WS_U(x) __builtin_amdgcn_readfirstlane(x)   // (no instruction when the load was scalar; keeps the "s" asm operands legal otherwise)
    d.tc.n = WS_U(a.x); d.tc.cy = WS_U(a.y); d.tc.ox0 = WS_U(a.z); d.tc.oy0 = WS_U(a.w); d.tc.oz0 = WS_U(b.x); d.tc.sp = 0;
    d.flags = WS_U(b.y); d.vo = WS_U(b.z); d.ibase = WS_U(b.w);
#undef WS_U
    return d;
}

// ---- producer ----------------------------------------------------------------------------------------
// Lane pairing: producer thread q stages channel octet (q & 1) of the halo voxels v = (q >> 1) + (WS_PROD / 2) * j.
// Two neighbouring lanes read the two 16-byte octets of the same voxel record (32 contiguous bytes) and neighbouring
// lane pairs read neighbouring voxels, so a wave-wide load touches each cache line it needs once.  (One lane per voxel
// with separate "low octet" / "high octet" instructions touches every line twice: the L1 tag pipeline, not HBM, was
// the producers' limit.)
#define WS_MAXV (2 * 1536 / WS_PROD)  // (voxel, octet) items per producer thread (HV <= 1536)

struct ProdConst {
    int rel[WS_MAXV];  // input voxel index of halo voxel j relative to the tile's halo origin
    int hc[WS_MAXV];   // packed halo coordinates hx | hy << 10 | hz << 20
    unsigned in_halo;  // bit j: v < HV
    unsigned face[6];  // bit j: halo voxel j lies on the halo's face x = 0, x = h0 - 1, y = 0, y = h1 - 1, z = 0, z = h2 - 1
};

__device__ __forceinline__ ProdConst prod_const(const ConvArgs& p, int q, int HV) {
    ProdConst k;
    k.in_halo = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f) k.face[f] = 0;
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        // (LDS index v = hx * xs + hy * h2 + hz; xs >= h1 * h2: the slots of a padded plane's tail belong to no halo voxel)
        const int v = (q >> 1) + (WS_PROD / 2) * j;
        const int hx = v / p.xs, rem = v - hx * p.xs;
        const int hz = rem % p.h2, hy = rem / p.h2;
        const bool in = v < HV && hy < p.h1;
        k.rel[j] = in ? (hx * p.Hi + hy) * p.Wi + hz : 0;
        k.hc[j] = in ? (hx | (hy << 10) | (hz << 20)) : 0;
        const unsigned bit = in ? (1u << j) : 0u;
        k.in_halo |= bit;
        k.face[0] |= hx == 0 ? bit : 0u;
        k.face[1] |= hx == p.h0 - 1 ? bit : 0u;
        k.face[2] |= hy == 0 ? bit : 0u;
        k.face[3] |= hy == p.h1 - 1 ? bit : 0u;
        k.face[4] |= hz == 0 ? bit : 0u;
        k.face[5] |= hz == p.h2 - 1 ? bit : 0u;
    }
    return k;
}

struct ProdItems {
    int gi[WS_MAXV];  // input voxel index (0 when padding / beyond the halo)
    unsigned ok;      // bit j: voxel j is inside the input tensor
};

__device__ __forceinline__ void prod_setup(const ConvArgs& p, const TileCoord& tc, const ProdConst& k, ProdItems& it) {
    const int ix0 = tc.ox0 * p.s0 - p.p0, iy0 = tc.oy0 * p.s1 - p.p1, iz0 = tc.oz0 * p.s2 - p.p2;
    const int base = (ix0 * p.Hi + iy0) * p.Wi + iz0;
    const bool interior = ix0 >= 0 && iy0 >= 0 && iz0 >= 0 && ix0 + p.h0 <= p.Di && iy0 + p.h1 <= p.Hi && iz0 + p.h2 <= p.Wi;
    if (interior) {  // wave-uniform: the whole halo lies inside the tensor (lanes beyond the halo have rel == 0)
        it.ok = k.in_halo;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) it.gi[j] = base + k.rel[j];
    } else if (ix0 >= -1 && iy0 >= -1 && iz0 >= -1 && ix0 + p.h0 <= p.Di + 1 && iy0 + p.h1 <= p.Hi + 1 && iz0 + p.h2 <= p.Wi + 1) {
        // wave-uniform: the halo sticks out of the tensor by exactly one voxel layer on some sides (the conv padding of a
        // tile at the tensor's border -- with 4 x 4 x 32 tiles on 128^3 more than half of all tiles): the voxels outside are
        // whole faces of the halo, whose per-lane item masks are kernel constants.  ~35 instructions instead of ~165.
        unsigned out = 0;
        out |= ix0 < 0 ? k.face[0] : 0u;
        out |= ix0 + p.h0 > p.Di ? k.face[1] : 0u;
        out |= iy0 < 0 ? k.face[2] : 0u;
        out |= iy0 + p.h1 > p.Hi ? k.face[3] : 0u;
        out |= iz0 < 0 ? k.face[4] : 0u;
        out |= iz0 + p.h2 > p.Wi ? k.face[5] : 0u;
        const unsigned okm = k.in_halo & ~out;
        it.ok = okm;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) it.gi[j] = ((okm >> j) & 1u) ? base + k.rel[j] : 0;
    } else {
        // straight-line (no short-circuit branches): unsigned compares fold the two-sided range checks
        unsigned okm = 0;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) {
            const unsigned ix = (unsigned)(ix0 + (k.hc[j] & 1023)), iy = (unsigned)(iy0 + ((k.hc[j] >> 10) & 1023)),
                           iz = (unsigned)(iz0 + (k.hc[j] >> 20));
            const unsigned ok = (unsigned)(ix < (unsigned)p.Di) & (unsigned)(iy < (unsigned)p.Hi) & (unsigned)(iz < (unsigned)p.Wi) &
                                ((k.in_halo >> j) & 1u);
            it.gi[j] = ok ? base + k.rel[j] : 0;
            okm |= ok << j;
        }
        it.ok = okm;
    }
}

// the same with the tile's descriptor: the halo class and the faces that stick out were decided when the table was built
__device__ __forceinline__ void prod_setup_desc(const ConvArgs& p, const TileDesc& d, const ProdConst& k, ProdItems& it) {
    const int cls = (d.flags >> 4) & 3;
    const int base = d.ibase;
    if (cls == 0) {
        it.ok = k.in_halo;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) it.gi[j] = base + k.rel[j];
    } else if (cls == 1) {
        unsigned out = 0;
#pragma unroll
        for (int f = 0; f < 6; ++f) out |= (d.flags >> (8 + f)) & 1 ? k.face[f] : 0u;
        const unsigned okm = k.in_halo & ~out;
        it.ok = okm;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) it.gi[j] = ((okm >> j) & 1u) ? base + k.rel[j] : 0;
    } else {
        prod_setup(p, d.tc, k, it);
    }
}

// Global-memory accesses in the scalar-base form `global_{load,store} v_off, ..., s[base:base+1]`: a wave-uniform pointer
// pinned in an SGPR pair (the empty asm also keeps hipcc from folding the lane offset into one 64-bit VGPR address, which
// selects the slow VGPR-pair form again) + a 32-bit lane offset.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define WS_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ WS_GLOBAL unsigned char* sgpr_ptr(const void* p) {
    // (readfirstlane: free when the value already lives in SGPRs; when hipcc has moved a wave-uniform chain to the VALU -- it does
    //  with the descriptor fields -- the "+s" operand below is otherwise an "illegal VGPR to SGPR copy")
    const size_t v = (size_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    WS_GLOBAL unsigned char* g = (WS_GLOBAL unsigned char*)(((size_t)hi << 32) | lo);  // explicit global address space: the asm hides the provenance
    asm volatile("" : "+s"(g));
    return g;
}

#define OPAQUE4(a) "+v"((a).x), "+v"((a).y), "+v"((a).z), "+v"((a).w)

// a "use" of a uint4 as ONE 128-bit register tuple (OPAQUE4 names the four components separately, and hipcc then splits a
// global_load_dwordx4 destination into copies -- with a vmcnt(0) right behind the load)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void touch128(uint4& a) {
    u32x4_t t = {a.x, a.y, a.z, a.w};
    asm volatile("" : "+v"(t));
    a = make_uint4(t.x, t.y, t.z, t.w);
}

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_h2(float a, float b) {  // {half(a), half(b)}, round to nearest even
    union {
        h2_t v;
        unsigned u;
    } c;
    c.v = __builtin_convertvector(f2_t{a, b}, h2_t);
    return c.u;
}

// Deferred InstanceNorm + LeakyReLU on 8 channels in packed fp16: y = fma(x, s, t); y = max(y, slope * y).
// (scale, shift) are the fp16 roundings produced by k_norm_finalize; one rounding per element (fp16 fma), i.e. the
// result differs from the fp32-evaluated transform by the rounding of s and t only (numerics).
__device__ __forceinline__ uint4 norm_act8_pk(uint4 raw, const unsigned* w /* 4 x {scales, shifts} */, unsigned slope2) {
    union {
        uint4 u;
        h2_t v[4];
    } x;
    union {
        unsigned u;
        h2_t v;
    } s, t, sl;
    x.u = raw;
    sl.u = slope2;
    // staged over the four channel pairs (all fmas, then all muls, then all maxes): a packed-fp16 result feeding the next
    // packed op back to back costs an s_nop each time -- emitted pair by pair that was 2 nops per 3 useful instructions
    h2_t y[4], z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.u = w[2 * i];
        t.u = w[2 * i + 1];
        y[i] = __builtin_elementwise_fma(x.v[i], s.v, t.v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = y[i] * sl.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) x.v[i] = __builtin_elementwise_max(y[i], z[i]);
    return x.u;
}

// Registers of one chunk in flight between its global loads (prod_issue) and its LDS stores (prod_commit).  The two
// halves run one chunk-barrier apart, so the HBM/L2 round trip overlaps the consumers' work on the previous chunk
// instead of sitting on the producer's own critical path.
struct ChunkRegs {
    uint4 d[WS_MAXV];     // this lane's channel octet of its halo voxels
    unsigned ssw[8];      // this lane's octet: 4 channel pairs x {packed scales, packed shifts}
    unsigned ok;          // bit j: voxel j is inside the input tensor
    unsigned live;        // bit j: voxel j belongs to the halo (v < HV)
    int has_ss;
    int skip_halo;        // the LDS buffer already holds this chunk's halo (previous tile = same spatial tile, other couts)
    int cc, cy;           // k_conv_ws: (chunk, cout chunk) the set was issued for -- the weights follow by DMA at commit time
};

__device__ __forceinline__ void prod_issue(const ConvArgs& p, const TileCoord& tc, const ProdItems& it, unsigned live, int cc,
                                           bool skip_halo, int q, int dbg, ChunkRegs& rg) {
    int cg = cc * 16;
    const size_t in_vox = (size_t)p.Di * p.Hi * p.Wi;
    const __half* base;
    const unsigned* ss;  // 16 words per chunk: 8 channel pairs x {packed scales, packed shifts}
    int C;
    // chunk-planar activations [N][C/16][voxel][16]: the chunk's plane starts at ((n * C + cg) * in_vox) halves
    if (cg < p.C0) {
        base = p.src0 + ((size_t)tc.n * p.C0 + cg) * in_vox;
        ss = p.ss16_0 ? p.ss16_0 + ((size_t)tc.n * p.C0 + cg) : nullptr;
        C = p.C0;
    } else {
        cg -= p.C0;
        base = p.src1 + ((size_t)tc.n * p.C1 + cg) * in_vox;
        ss = p.ss16_1 ? p.ss16_1 + ((size_t)tc.n * p.C1 + cg) : nullptr;
        C = p.C1;
    }
    (void)C;
    if (dbg & 64) ss = nullptr;
    rg.skip_halo = skip_halo;
    if (skip_halo) return;
    rg.ok = it.ok;
    rg.live = live;
    rg.has_ss = ss != nullptr;
    const unsigned cb2 = 32u;  // bytes per voxel in a 16-channel plane: a wave load reads 1 KiB of consecutive bytes per 32 voxels of a row
    // wave-uniform 64-bit base (SGPR pair) + 32-bit lane offset: the `global_load v, v_off, s[base]` form.  The form with a
    // 64-bit VGPR address is starved next to a wave that keeps the matrix pipe busy (tools/valu_under_mfma.hip: 660
    // instead of 64 cycles per load instruction), the scalar-base form is not.
    const unsigned char* sbase = (const unsigned char*)base;
    const unsigned lane_off = (unsigned)(q & 1) * 16u;
    const int nrounds = (p.h0 * p.xs + WS_PROD / 2 - 1) / (WS_PROD / 2);  // rounds that carry halo voxels (wave-uniform)
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        // no per-lane test: voxels beyond the halo / outside the tensor have gi == 0 (a valid address, data discarded)
        if (j < nrounds && !(dbg & 128))  // (dbg 128: ablation without the halo loads)
        // v_mad_u32_u24 (full rate; the plain 32-bit form compiled to the quarter-rate v_mad_u64_u32): voxel index and
        // record size are below 2^24 (checked on the host)
        rg.d[j] = *(const uint4*)(sbase + (__umul24((unsigned)it.gi[j], cb2) + lane_off));
    }
    if (ss) {
        unsigned so = (unsigned)(q & 1) * 32u;
        asm volatile("" : "+v"(so));  // keep the 32 -> 64 bit extension in this block: scalar-base load form
        const uint4 s0 = *(const uint4*)((const unsigned char*)ss + so);
        const uint4 s1 = *(const uint4*)((const unsigned char*)ss + so + 16);
        rg.ssw[0] = s0.x; rg.ssw[1] = s0.y; rg.ssw[2] = s0.z; rg.ssw[3] = s0.w;
        rg.ssw[4] = s1.x; rg.ssw[5] = s1.y; rg.ssw[6] = s1.z; rg.ssw[7] = s1.w;
    }
}

// The chunk's weights (taps KiB) straight from L2 into the LDS buffer with LDS-DMA: no staging VGPRs, no ds_write pass.
// One global_load_lds_dwordx4 moves 64 lanes x 16 B to [M0 + lane * 16]; item i = q + b * WS_PROD lands at dst_w + i * 16,
// i.e. every wave writes whole 1 KiB blocks.  Scalar-base source form (see prod_issue).  The caller waits (vmcnt) before
// the chunk barrier: hipcc does not count these.
__device__ __forceinline__ void dma_weights(const ConvArgs& p, int cc, int cy, unsigned char* dst_w, int q, int taps) {
    const int nw = taps * 64;
    const __half* wsrc = p.wpk + ((size_t)cc * taps * 2) * p.Cout * 8 + (size_t)cy * 32 * 8;
    const unsigned lds0 = (unsigned)(size_t)dst_w + (unsigned)((q >> 6) * 64) * 16u;   // this wave's first block
#pragma unroll
    for (int b = 0; b < WS_WB; ++b) {
        const int i = q + b * WS_PROD;
        if (i < nw) {  // whole waves pass or fail together except in the last block (nw is a multiple of 64)
            const unsigned voff = ((unsigned)(i >> 5) * (unsigned)p.Cout + (unsigned)(i & 31)) * 16u;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(b * WS_PROD) * 16u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(wsrc), "s"(m0v)
                         : "memory");
        }
    }
}

template <bool SS, bool EDGE>
__device__ __forceinline__ void commit_items(const ChunkRegs& rg, unsigned char* d0, int nv, unsigned slope2) {
    // distinct markers keep hipcc from merging the copies back into one path with per-item masks
    if (EDGE)
        asm volatile("; commit: edge tile");
    else
        asm volatile("; commit: interior tile");
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        if (j < nv) {  // wave-uniform: whole rounds, lanes past the halo store into the plane's padding
            uint4 o = rg.d[j];
            if (SS) o = norm_act8_pk(o, rg.ssw, slope2);
            if (EDGE && !((rg.ok >> j) & 1u)) o = make_uint4(0, 0, 0, 0);
            *(uint4*)(d0 + j * (WS_PROD / 2 * 16)) = o;
        }
    }
}

__device__ __forceinline__ void prod_commit(const ConvArgs& p, ChunkRegs& rg, unsigned char* dst_in, int q, int HV, int plane,
                                            int dbg) {
    const int nv = (dbg & 32) ? 0 : (HV + WS_PROD / 2 - 1) / (WS_PROD / 2);
    if (rg.skip_halo) return;
    // the (scale, shift) words stay in VGPRs: a packed fp16 fma takes at most one scalar operand, so SGPR copies cost
    // a v_mov (+ hazard nops) per use -- more instructions than the transform itself
    union {
        unsigned u;
        h2_t v;
    } sl2;
    sl2.v = h2_t{(_Float16)p.slope, (_Float16)p.slope};
    unsigned char* d0 = dst_in + (q & 1) * plane + (q >> 1) * 16;
    // padding voxels (outside the tensor) must read as zero AFTER the transform; tiles whose halo lies inside the tensor
    // (rg.ok == rg.live for every lane, decided per wave) skip the per-voxel selects.  The four (transform, edge)
    // combinations are separate straight-line copies: 13 instructions per item in the common one (12 packed ops +
    // ds_write_b128 with an immediate offset) instead of ~35 with per-item masks and selects.
    const bool edge = __builtin_amdgcn_ballot_w64(rg.ok != rg.live) != 0;
    if (rg.has_ss) {
        if (edge)
            commit_items<true, true>(rg, d0, nv, sl2.u);
        else
            commit_items<true, false>(rg, d0, nv, sl2.u);
    } else {
        if (edge)
            commit_items<false, true>(rg, d0, nv, sl2.u);
        else
            commit_items<false, false>(rg, d0, nv, sl2.u);
    }
}

// ---- split-precision ("x3") producers ------------------------------------------------------------------
// The chunk is 8 fp32 channels of the raw activation ([N][C/8][voxel][8] floats: the same 32 bytes per voxel and plane as a
// 16-channel fp16 chunk, so prod_issue / prod_setup / the run tables are shared byte for byte: lane q holds channels 4 (q & 1) ..
// + 3 of its halo voxels and their (scale, shift) as 8 fp32 words).  Commit: y = lrelu(fma(x, scale, shift)) in fp32 -- the
// reference's CPU arithmetic -- then y = hi + lo with hi = half(y), lo = half(y - hi) (y - hi is exact in fp32; |lo| <= ulp(hi) / 2,
// so hi + lo carries 22 significant bits + sign of lo).  hi goes to LDS plane 0, lo to plane 1 (the consumers' k-half 0 / 1).
__device__ __forceinline__ void x3_split4(const float (&y)[4], uint2& hi, uint2& lo) {
    union {
        h2_t v;
        unsigned u;
    } h01, h23;
    h01.v = __builtin_convertvector(f2_t{y[0], y[1]}, h2_t);
    h23.v = __builtin_convertvector(f2_t{y[2], y[3]}, h2_t);
    const float r0 = y[0] - (float)h01.v[0], r1 = y[1] - (float)h01.v[1], r2 = y[2] - (float)h23.v[0], r3 = y[3] - (float)h23.v[1];
    hi = make_uint2(h01.u, h23.u);
    lo = make_uint2(cvt_pk_h2(r0, r1), cvt_pk_h2(r2, r3));
}

template <bool SS, bool EDGE>
__device__ __forceinline__ void commit_items_x3(const ChunkRegs& rg, unsigned char* d0, int plane, int nv, float slope) {
    if (EDGE)
        asm volatile("; commit x3: edge tile");
    else
        asm volatile("; commit x3: interior tile");
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        if (j < nv) {
            float y[4] = {__uint_as_float(rg.d[j].x), __uint_as_float(rg.d[j].y), __uint_as_float(rg.d[j].z), __uint_as_float(rg.d[j].w)};
            if (SS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float f = __builtin_fmaf(y[i], __uint_as_float(rg.ssw[2 * i]), __uint_as_float(rg.ssw[2 * i + 1]));
                    y[i] = f > 0.f ? f : f * slope;
                }
            }
            uint2 hi, lo;
            x3_split4(y, hi, lo);
            if (EDGE && !((rg.ok >> j) & 1u)) hi = lo = make_uint2(0, 0);
            *(uint2*)(d0 + j * (WS_PROD / 2 * 16)) = hi;
            *(uint2*)(d0 + plane + j * (WS_PROD / 2 * 16)) = lo;
        }
    }
}

__device__ __forceinline__ void prod_commit_x3(const ConvArgs& p, ChunkRegs& rg, unsigned char* dst_in, int q, int HV, int plane, int dbg) {
    const int nv = (dbg & 32) ? 0 : (HV + WS_PROD / 2 - 1) / (WS_PROD / 2);
    if (rg.skip_halo) return;
    unsigned char* d0 = dst_in + (q >> 1) * 16 + (q & 1) * 8;
    const bool edge = __builtin_amdgcn_ballot_w64(rg.ok != rg.live) != 0;
    if (rg.has_ss) {
        if (edge)
            commit_items_x3<true, true>(rg, d0, plane, nv, p.slope);
        else
            commit_items_x3<true, false>(rg, d0, plane, nv, p.slope);
    } else {
        if (edge)
            commit_items_x3<false, true>(rg, d0, plane, nv, p.slope);
        else
            commit_items_x3<false, false>(rg, d0, plane, nv, p.slope);
    }
}

