// k_conv_ws: wave-specialised, persistent implicit-GEMM conv for gfx950.
//
// One 512-thread workgroup per CU: waves 0-3 ("consumers") issue nothing but LDS fragment reads and
// v_mfma_f32_32x32x16_f16; waves 4-7 ("producers") gather the next 16-channel chunk of the input halo tile from
// HBM/L2, apply the producer layer's deferred InstanceNorm + LeakyReLU and write it (plus, when they do not fit
// resident, the chunk's weights) into the other LDS buffer.  MFMA and VALU/VMEM are separate pipes of a SIMD, so
// the two roles overlap; one __syncthreads() per chunk hands buffers over.  Workgroups are persistent and walk
// consecutive output tiles, so the staging of a tile's first chunk hides under the previous tile's last chunk and
// the chip-wide working set at any time is a contiguous run of tiles (halo reuse in L2).
//
// Consumer inner loop: the taps are compile-time (template K0,K1,K2), fully unrolled, with the A/B fragments of
// tap t+1 read into a second register set while the MFMAs of tap t issue (hipcc otherwise emits
// ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per fragment: ~200 cycles per MFMA instead of 32).
// Producer: per-tile voxel offsets live in registers; per chunk every global load is issued before the first
// is consumed.
//
// Same arithmetic as k_conv_mfma (conv.hip); statistics partials are written per consumer wave
// (partials[n][cout][2][tiles * 4]) so that no cross-wave reduction (and no extra barrier) is needed.
#include <stdlib.h>

#include "conv.h"

#define WS_THREADS 512

__device__ __forceinline__ int ws_plane_bytes(int HV) { return ((HV * 16 + 127) / 128) * 128 + 64; }

struct TileCoord {
    int n, cy, ox0, oy0, oz0;
    int sp;  // spatial tile index
};

__device__ __forceinline__ TileCoord decode_tile(const ConvArgs& p, int t) {
    // order: spatial tiles fastest (x fastest, then z, then y), then cout chunk, then n.  With the dominant halo
    // overlap along x (thin tiles in x), consecutive tiles share their x-halo planes.
    TileCoord c;
    const int nsp = p.t0 * p.t1 * p.t2;
    const int ncy = p.Cout / 32;
    c.sp = t % nsp;
    int r = t / nsp;
    c.cy = r % ncy;
    c.n = r / ncy;
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

// Tile sequence of a workgroup.  Workgroup b is observed to run on XCD b % 8 (each XCD has a private 4 MiB L2):
// the tile list is cut into 8 contiguous ranges, one per XCD, and the 32 workgroups of an XCD walk their range
// together (stride = workgroups per XCD), so the halos an XCD re-reads are the ones its own L2 just fetched.
// Placement only changes speed, never results.
struct TileWalk {
    int first, stride, count;
};

__device__ __forceinline__ TileWalk tile_walk(int total_tiles, bool getenv_free_walk_contiguous) {
    TileWalk w;
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    if (G % 8 != 0 || total_tiles < G) {
        w.first = b;
        w.stride = G;
        w.count = b < total_tiles ? (total_tiles - b + G - 1) / G : 0;
        return w;
    }
    const int xcd = b & 7, slot = b >> 3, per = G >> 3;  // per = workgroups per XCD
    const int q = total_tiles / 8, rem = total_tiles % 8;
    const int lo = xcd * q + min(xcd, rem);
    const int len = q + (xcd < rem ? 1 : 0);
    if (getenv_free_walk_contiguous) {
        // contiguous run per workgroup: consecutive tiles of a workgroup are neighbours along x (x-fastest tile order),
        // so the x-halo planes a tile re-reads were fetched by the same CU one tile earlier; neighbouring workgroups of
        // the XCD walk neighbouring lines
        const int cq = len / per, cr = len % per;
        w.first = lo + slot * cq + min(slot, cr);
        w.stride = 1;
        w.count = cq + (slot < cr ? 1 : 0);
        return w;
    }
    w.first = lo + slot;
    w.stride = per;
    w.count = slot < len ? (len - slot + per - 1) / per : 0;
    return w;
}

// ---- producer ----------------------------------------------------------------------------------------
// Producer thread q (0..255) owns halo voxels v = q + 256 * j and stages BOTH channel octets of the chunk for them,
// so the voxel bookkeeping is paid once per 32 bytes and the chunk's 16 (scale, shift) pairs are wave-uniform
// (kept in SGPRs).
#define WS_MAXV 6  // halo voxels per producer thread (HV <= 256 * WS_MAXV)

struct HaloStep {
    int hx0, hy0, hz0;  // halo coordinates of this thread's first voxel
    int dx, dy, dz;     // mixed-radix representation of the 256-voxel stride
};

__device__ __forceinline__ HaloStep halo_step(const ConvArgs& p, int q) {
    HaloStep h;
    h.hz0 = q % p.h2;
    const int t = q / p.h2;
    h.hy0 = t % p.h1;
    h.hx0 = t / p.h1;
    h.dz = 256 % p.h2;
    const int t2 = 256 / p.h2;
    h.dy = t2 % p.h1;
    h.dx = t2 / p.h1;
    return h;
}

struct ProdItems {
    int gi[WS_MAXV];  // input voxel index (0 when padding / beyond the halo)
    unsigned ok;      // bit j: voxel j is inside the input tensor
};

__device__ __forceinline__ void prod_setup(const ConvArgs& p, const TileCoord& tc, int q, int HV, const HaloStep& hs,
                                           ProdItems& it) {
    const int ix0 = tc.ox0 * p.s0 - p.p0, iy0 = tc.oy0 * p.s1 - p.p1, iz0 = tc.oz0 * p.s2 - p.p2;
    int v = q;
    int hz = hs.hz0, hy = hs.hy0, hx = hs.hx0;
    it.ok = 0;
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        const int ix = ix0 + hx, iy = iy0 + hy, iz = iz0 + hz;
        const bool ok = v < HV && ix >= 0 && ix < p.Di && iy >= 0 && iy < p.Hi && iz >= 0 && iz < p.Wi;
        it.gi[j] = ok ? (ix * p.Hi + iy) * p.Wi + iz : 0;
        it.ok |= ok ? (1u << j) : 0u;
        v += 256;
        hz += hs.dz;
        const int cz = hz >= p.h2 ? 1 : 0;
        hz -= cz * p.h2;
        hy += hs.dy + cz;
        const int cy_ = hy >= p.h1 ? 1 : 0;
        hy -= cy_ * p.h1;
        hx += hs.dx + cy_;
    }
}

#define OPAQUE4(a) "+v"((a).x), "+v"((a).y), "+v"((a).z), "+v"((a).w)

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.u = w[2 * i];
        t.u = w[2 * i + 1];
        const h2_t y = __builtin_elementwise_fma(x.v[i], s.v, t.v);
        x.v[i] = __builtin_elementwise_max(y, y * sl.v);
    }
    return x.u;
}

__device__ __forceinline__ void prod_stage(const ConvArgs& p, const TileCoord& tc, const ProdItems& it, int cc,
                                           unsigned char* dst_in, unsigned char* dst_w, int q, int HV, int plane,
                                           int taps, int dbg) {
    int cg = cc * 16;
    const size_t in_vox = (size_t)p.Di * p.Hi * p.Wi;
    const __half* base;
    const unsigned* ss;  // 16 words: 8 channel pairs x {packed scales, packed shifts}
    int C;
    if (cg < p.C0) {
        base = p.src0 + (size_t)tc.n * in_vox * p.C0 + cg;
        ss = p.ss16_0 ? p.ss16_0 + ((size_t)tc.n * p.C0 + cg) : nullptr;
        C = p.C0;
    } else {
        cg -= p.C0;
        base = p.src1 + (size_t)tc.n * in_vox * p.C1 + cg;
        ss = p.ss16_1 ? p.ss16_1 + ((size_t)tc.n * p.C1 + cg) : nullptr;
        C = p.C1;
    }
    // Every global load of the chunk (scale/shift, weights, halo) is issued before any result is consumed: the
    // producer is bound by memory round trips, not by instructions -- one round trip per chunk instead of five.
    const int nv = (dbg & 32) ? 0 : (HV + 255) >> 8;
    if (dbg & 64) ss = nullptr;
    uint4 lo[WS_MAXV], hi[WS_MAXV];
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        // unconditional: voxels beyond the halo have gi == 0 (a valid address whose data is discarded)
        const __half* src = base + (size_t)it.gi[j] * C;
        lo[j] = *(const uint4*)src;
        hi[j] = *(const uint4*)(src + 8);
    }
    constexpr int WB = 7;  // weight items per thread: taps * 64 / 256 <= 6.75 for 27 taps
    uint4 wv[WB];
    const int nw = taps * 64;
    if (dst_w) {
        const __half* wsrc = p.wpk + ((size_t)cc * taps * 2) * p.Cout * 8 + (size_t)tc.cy * 32 * 8;
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int i = min(q + b * 256, nw - 1);
            wv[b] = *(const uint4*)(wsrc + ((size_t)(i >> 5) * p.Cout + (i & 31)) * 8);
        }
    }
    unsigned ssw[16];
    if (ss) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ssw[j] = ss[j];
    }
    // ---- consume ----
    if (dst_w) {
        asm volatile("" : OPAQUE4(wv[0]), OPAQUE4(wv[1]), OPAQUE4(wv[2]), OPAQUE4(wv[3]), OPAQUE4(wv[4]), OPAQUE4(wv[5]),
                     OPAQUE4(wv[6]));
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int i = min(q + b * 256, nw - 1);  // clamped lanes rewrite item nw-1 with identical data
            *(uint4*)(dst_w + i * 16) = wv[b];
        }
    }
    // the chunk's 16 (scale, shift) values are the same for every thread: keep them scalar
    if (ss) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ssw[j] = __builtin_amdgcn_readfirstlane(ssw[j]);
    }
    union {
        unsigned u;
        h2_t v;
    } sl2;
    sl2.v = h2_t{(_Float16)p.slope, (_Float16)p.slope};
    unsigned char* d0 = dst_in + q * 16;
    unsigned char* d1 = d0 + plane;
#pragma unroll
    for (int j0 = 0; j0 < WS_MAXV; j0 += 3) {
        if (j0 < nv) {
            asm volatile("" : OPAQUE4(lo[j0]), OPAQUE4(hi[j0]), OPAQUE4(lo[j0 + 1]), OPAQUE4(hi[j0 + 1]), OPAQUE4(lo[j0 + 2]),
                         OPAQUE4(hi[j0 + 2]));
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int j = j0 + b;
                if (q + 256 * j < HV) {
                    uint4 o0 = lo[j], o1 = hi[j];
                    if (ss) {
                        o0 = norm_act8_pk(o0, ssw, sl2.u);
                        o1 = norm_act8_pk(o1, ssw + 8, sl2.u);
                    }
                    if (!((it.ok >> j) & 1u)) o0 = o1 = make_uint4(0, 0, 0, 0);
                    *(uint4*)(d0 + j * 4096) = o0;
                    *(uint4*)(d1 + j * 4096) = o1;
                }
            }
        }
    }
}

// ---- consumer ----------------------------------------------------------------------------------------
// One 16-channel chunk: acc[r] += W[tap] x X[tap][r] for all taps.  bp[r]: LDS address of this lane's voxel of
// M-tile r in this lane's k-half plane; ap: LDS address of this lane's row of the first A fragment.
template <int R, int K0, int K1, int K2>
__device__ __forceinline__ void consume_chunk(const unsigned char* const (&bp)[R], const unsigned char* ap, int h1, int h2,
                                              f32x16 (&acc)[R]) {
    // Prefetch distance is one tap.  (Distance 2 with the pipeline pinned by sched_barrier(0) was measured slower:
    // 1558 vs 1405 us on the 32->32 @128^3 layer at batch 8 -- 13 spilled VGPRs and a stiffer schedule.)
    constexpr int T = K0 * K1 * K2;
    f16x8 a[2];
    f16x8 b[2][R];
    const unsigned char* rb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rb[r] = bp[r];
    a[0] = *(const f16x8*)ap;
#pragma unroll
    for (int r = 0; r < R; ++r) b[0][r] = *(const f16x8*)rb[r];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int cb = t & 1, nb = cb ^ 1;
        if (t + 1 < T) {
            const int tn = t + 1;
            const int dzn = tn % K2, dyn = (tn / K2) % K1, dxn = tn / (K2 * K1);
            if (dzn == 0) {
                const int rowoff = (dxn * h1 + dyn) * h2 * 16;
#pragma unroll
                for (int r = 0; r < R; ++r) rb[r] = bp[r] + rowoff;
            }
            a[nb] = *(const f16x8*)(ap + tn * 1024);
#pragma unroll
            for (int r = 0; r < R; ++r) b[nb][r] = *(const f16x8*)(rb[r] + dzn * 16);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb], b[cb][r], acc[r], 0, 0, 0);
        // scheduling hint: the next tap's R+1 fragment reads go out before this tap's R MFMAs
        if (t + 1 < T) __builtin_amdgcn_sched_group_barrier(0x100, R + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, R, 0);
    }
}

template <int R, int K0, int K1, int K2>
__global__ __launch_bounds__(WS_THREADS) void k_conv_ws(ConvArgs p, int total_tiles, int resident_w, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    constexpr int taps = K0 * K1 * K2;
    const int HV = p.h0 * p.h1 * p.h2;
    const int plane = ws_plane_bytes(HV);
    const int ncc = (p.C0 + p.C1) / 16;
    const int nsp = p.t0 * p.t1 * p.t2;
    // LDS map.  resident: [all weights: ncc * taps KiB][halo buf 0][halo buf 1]
    //           streamed: [halo 0 | w 0][halo 1 | w 1]
    const int wres_bytes = resident_w ? ncc * taps * 1024 : 0;
    const int buf_bytes = resident_w ? 2 * plane : 2 * plane + taps * 1024;
    unsigned char* bufs = smem + wres_bytes;

    const TileWalk walk = tile_walk(total_tiles, !(dbg & 128));
    const int my_chunks = walk.count * ncc;

    if (resident_w) {
        // (only used when Cout == 32: every tile of the launch uses the same weights)
        const int nw = ncc * taps * 64;
        for (int i = tid; i < nw; i += WS_THREADS)
            *(uint4*)(smem + i * 16) = *(const uint4*)(p.wpk + ((size_t)(i >> 5) * p.Cout + (i & 31)) * 8);
        // visibility to the consumers is ordered by the first chunk barrier below
    }

    // consumer per-lane constants (tile independent)
    const int cw = wave & 3;
    const int lz = l31 & (p.w2 - 1);
    const int ly = (l31 >> p.lw2) & (p.w1 - 1);
    const int lx = l31 >> (p.lw2 + p.lw1);
    int hoff[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = cw * R + r;
        const int mz = m & (p.b2 - 1);
        const int my = (m >> p.lb2) & (p.b1 - 1);
        const int mx = m >> (p.lb2 + p.lb1);
        const int tx = mx * p.w0 + lx, ty = my * p.w1 + ly, tz = mz * p.w2 + lz;
        hoff[r] = (((tx * p.s0) * p.h1 + ty * p.s1) * p.h2 + tz * p.s2) * 16 + kh * plane;
    }
    f32x16 acc[R];
    const HaloStep hstep = halo_step(p, tid & 255);

    // producer state (valid across the chunks of one tile) and chunk counters of both roles
    TileCoord ptc;
    ptc.n = ptc.cy = ptc.ox0 = ptc.oy0 = ptc.oz0 = ptc.sp = 0;
    ProdItems items;
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) items.gi[j] = 0;
    items.ok = 0;
    int pk = 0, pcc = 0;  // producer: tile counter, chunk within tile (of chunk g + 1)
    int ck = 0, ccc = 0;  // consumer: the same for chunk g

    for (int g = -1; g < my_chunks; ++g) {
        if (producer) {
            if (g + 1 < my_chunks && !(dbg & 2)) {
                if (pcc == 0) {
                    ptc = decode_tile(p, walk.first + pk * walk.stride);
                    prod_setup(p, ptc, tid - 256, HV, hstep, items);
                }
                unsigned char* nxt = bufs + ((g + 1) & 1) * buf_bytes;
                prod_stage(p, ptc, items, pcc, nxt, (resident_w || (dbg & 16)) ? nullptr : nxt + 2 * plane, tid - 256, HV,
                           plane, taps, dbg);
                if (++pcc == ncc) {
                    pcc = 0;
                    ++pk;
                }
            }
        } else if (g >= 0) {
            const int k = ck, cc = ccc;
            if (++ccc == ncc) {
                ccc = 0;
                ++ck;
            }
            const unsigned char* cur = bufs + (g & 1) * buf_bytes;
            if (cc == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
            }
            const unsigned char* bp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // opaque per chunk: stops hipcc from hoisting the 9 x R row addresses out of the chunk loop (spills)
                int ho = hoff[r];
                asm volatile("" : "+v"(ho));
                bp[r] = cur + ho;
            }
            const unsigned char* ap = (resident_w ? smem + cc * taps * 1024 : cur + 2 * plane) + (kh * 32 + l31) * 16;
            if (!(dbg & 1)) consume_chunk<R, K0, K1, K2>(bp, ap, p.h1, p.h2, acc);
            if (cc == ncc - 1 && !(dbg & 8)) {
                // ---- epilogue: + bias, fp16 convert, InstanceNorm partials, LDS transpose, 16-byte coalesced stores
                // (D fragment = 4 couts per lane at a 64-byte voxel pitch: stored directly that is 16 strided 8-byte
                //  store instructions per wave and tile, which are store-issue bound -- ~600 cycles each; through a
                //  per-wave LDS slab each store instruction writes 1 KiB of whole 64-byte voxel records)
                const TileCoord tc = decode_tile(p, walk.first + k * walk.stride);
                const int cout0 = tc.cy * 32;
                const size_t out_vox = (size_t)p.Do * p.Ho * p.Wo;
                const int npart = nsp * 4;
                unsigned char* slab = bufs + 2 * buf_bytes + cw * (32 * 80);
                // after the transpose this lane owns couts [8 sq, 8 sq + 8) of voxels sv and sv + 16 of each M-tile
                const int sv = lane >> 2, sq = lane & 3;
                float s[8], q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
                // the tile's 16 bias values of this lane (4 independent loads, one wait)
                float4 bq[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) bq[gq] = *(const float4*)(p.bias + cout0 + 8 * gq + 4 * kh);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int m = cw * R + r;
                    const int mx0 = tc.ox0 + (m >> (p.lb2 + p.lb1)) * p.w0, my0 = tc.oy0 + ((m >> p.lb2) & (p.b1 - 1)) * p.w1,
                              mz0 = tc.oz0 + (m & (p.b2 - 1)) * p.w2;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        union {
                            uint2 u;
                            __half h[4];
                        } pk;
                        pk.h[0] = __float2half_rn(acc[r][gq * 4 + 0] + bq[gq].x);
                        pk.h[1] = __float2half_rn(acc[r][gq * 4 + 1] + bq[gq].y);
                        pk.h[2] = __float2half_rn(acc[r][gq * 4 + 2] + bq[gq].z);
                        pk.h[3] = __float2half_rn(acc[r][gq * 4 + 3] + bq[gq].w);
                        *(uint2*)(slab + l31 * 80 + (8 * gq + 4 * kh) * 2) = pk.u;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int v = sv + 16 * half;  // voxel of the M-tile
                        union {
                            uint4 u;
                            __half h[8];
                        } d;
                        d.u = *(const uint4*)(slab + v * 80 + sq * 16);
                        const int vz = v & (p.w2 - 1), vy = (v >> p.lw2) & (p.w1 - 1), vx = v >> (p.lw2 + p.lw1);
                        const int ox = mx0 + vx, oy = my0 + vy, oz = mz0 + vz;
                        if (ox < p.Do && oy < p.Ho && oz < p.Wo) {
                            if (!(dbg & 4)) *(uint4*)(p.out + ((size_t)tc.n * out_vox + ((size_t)ox * p.Ho + oy) * p.Wo + oz) * p.Cout +
                                      cout0 + sq * 8) = d.u;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float vr = __half2float(d.h[i]);
                                s[i] += vr;
                                q[i] = __builtin_fmaf(vr, vr, q[i]);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // reduce over the 16 lanes that share sq (lane bits 2..5)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int mm = 4; mm < 64; mm <<= 1) {
                        s[i] += __shfl_xor(s[i], mm);
                        q[i] += __shfl_xor(q[i], mm);
                    }
                }
                if (lane < 4) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = sq * 8 + i;
                        float* pp = p.partials + (((size_t)tc.n * p.Cout + cout0 + row) * 2) * npart + tc.sp * 4 + cw;
                        pp[0] = s[i];
                        pp[npart] = q[i];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- host side ---------------------------------------------------------------------------------------
static size_t ws_plane_host(int HV) { return ((size_t)(HV * 16 + 127) / 128) * 128 + 64; }

// resident weights need every tile of the launch to use the same weights: Cout == 32 (one cout chunk)
bool conv_ws_resident(int HV, int taps, int ncc, int Cout) {
    return Cout == 32 && (size_t)ncc * taps * 1024 + 4 * ws_plane_host(HV) + 4 * 32 * 80 <= 160 * 1024;
}

size_t conv_ws_lds_bytes(int HV, int taps, int ncc, int Cout) {
    const size_t slabs = 4 * 32 * 80;  // per-consumer-wave epilogue transpose slabs
    if (conv_ws_resident(HV, taps, ncc, Cout)) return (size_t)ncc * taps * 1024 + 4 * ws_plane_host(HV) + slabs;
    return 2 * (2 * ws_plane_host(HV) + (size_t)taps * 1024) + slabs;
}

bool conv_ws_supported(const int k[3], int HV) {
    const bool k333 = k[0] == 3 && k[1] == 3 && k[2] == 3;
    const bool k133 = k[0] == 1 && k[1] == 3 && k[2] == 3;
    return (k333 || k133) && HV <= 256 * WS_MAXV;
}

template <int R, int K0, int K1, int K2>
static void launch_ws_t(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, int total, int grid, int resident) {
    static bool once = (hipFuncSetAttribute((const void*)k_conv_ws<R, K0, K1, K2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL((k_conv_ws<R, K0, K1, K2>), dim3(grid), dim3(WS_THREADS), t.lds_bytes, ctx->stream, a, total, resident,
                       getenv("BOA_WS_DBG") ? atoi(getenv("BOA_WS_DBG")) : 0);
}

template <int R>
static int launch_ws_r(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, int total, int grid, int resident) {
    if (a.k0 == 3 && a.k1 == 3 && a.k2 == 3)
        launch_ws_t<R, 3, 3, 3>(ctx, a, t, total, grid, resident);
    else if (a.k0 == 1 && a.k1 == 3 && a.k2 == 3)
        launch_ws_t<R, 1, 3, 3>(ctx, a, t, total, grid, resident);
    else {
        boa_set_error("conv_ws: kernel %dx%dx%d not instantiated", a.k0, a.k1, a.k2);
        return BOA_EINVAL;
    }
    return BOA_OK;
}

int launch_conv_ws(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, double flops, double bytes) {
    const int total = t.tiles[0] * t.tiles[1] * t.tiles[2] * (a.Cout / 32) * a.N;
    const int grid = std::min(total, ctx->cu_count);
    const int taps = a.k0 * a.k1 * a.k2;
    const int HV = t.h[0] * t.h[1] * t.h[2];
    const int resident = conv_ws_resident(HV, taps, (a.C0 + a.C1) / 16, a.Cout) ? 1 : 0;
    KernelTimer tm(ctx, BOA_K_CONV_MFMA, flops, bytes);
    int rc;
    switch (t.R) {
        case 4: rc = launch_ws_r<4>(ctx, a, t, total, grid, resident); break;
        case 2: rc = launch_ws_r<2>(ctx, a, t, total, grid, resident); break;
        case 1: rc = launch_ws_r<1>(ctx, a, t, total, grid, resident); break;
        default:
            boa_set_error("conv_ws: unsupported R=%d", t.R);
            rc = BOA_EINVAL;
    }
    tm.stop();
    if (rc) return rc;
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
