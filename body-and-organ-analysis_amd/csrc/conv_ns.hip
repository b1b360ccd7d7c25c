// k_conv_ns: wave-specialised persistent implicit-GEMM 3x3x3 conv whose consumer waves split the COUT axis (gfx950).
//
// k_conv_ws (conv_ws.hip) gives each of its four consumer waves its own voxels and ONE 32-cout chunk per staged halo: a layer
// with Cout = 128 ... 320 stages every input halo Cout / 32 times and streams the chunk's weights through LDS for every tile.
// That is the right trade for the full-resolution 32-channel layers and the wrong one for the stride-2 and deep layers
// (profiles/r02_per_layer.txt: 150-400 TFLOP/s, producer / L2 bound).  Here
//   * the block's output tile is 4 x-planes of 4 x 8 voxels (M-tile = one plane = 32 voxels) and all four consumer waves work
//     on the SAME voxels with DIFFERENT cout chunks (cout group = 4 chunks = 128 couts per staged halo);
//   * a wave's weight (A) fragments are therefore private to it: they go straight from global / L2 into registers (scalar
//     base + lane offset, a ring of three (dy, dz) tap groups, prefetched two groups = 24 MFMAs ahead, across chunk and tile
//     boundaries) -- no weight staging, no LDS space, no LDS reads for weights;
//   * the input (B) fragments come from the LDS halo with reuse along x: per (dy, dz) group the wave reads the S * 3 + 3 input
//     planes once and feeds 12 MFMAs (stride 1: plane j serves the outputs r = j - dx; stride 2: the planes 2 r + dx), 0.5 /
//     0.75 KiB of LDS reads per MFMA;
//   * producers (waves 4-7), tile sequence / virtual workgroups / run tables, deferred InstanceNorm in the staging, bias as the
//     first MFMA's C operand, register-transpose epilogue and batch-invariant statistics are k_conv_ws's (conv_ws_dev.h).
// tools/consumer_ns.hip is the isolated consumer loop (1.4-1.6 PFLOP/s with random operands, weights up to 5 MB from L2).
// Instantiated for 3x3x3 kernels with stride 1 or 2 on all axes; everything else stays on k_conv_ws.
#include <stdlib.h>

#include "conv.h"

#include "conv_ws_dev.h"

// ---- LDS halo layout of the stride-2 instantiations ("SW") ------------------------------------------------
// A consumer lane's voxel of M-tile plane x is (y, z) = (2 ly + dy, 2 lz + dz): in the linear [x][y][z] halo the 32 lanes of a k-half
// sit 32 bytes apart along z and 2 * 17 * 16 = 544 bytes apart along y, i.e. on EVEN 16-byte slots only and on the same slots (mod 256
// bytes) in three of the four ly rows -- ds_read_b128 is serviced in four fixed 16-lane groups ({0-3, 12-15, 20-27}, ...: hardware
// guide, LDS table), and every group met 3-way bank conflicts: the fragment reads of the stride-2 layers cost three LDS cycles where
// one would do, on an LDS that the 0.75 - 0.83 KiB of reads per MFMA already fills (round-5 trace: 2 800 cycles per chunk of 54 MFMAs).
// The stride-2 halo is therefore stored DE-INTERLEAVED by the parity of y and z:
//   slot(x, y, z) = (x * 9 + yp(y)) * 24 + zp(z),   yp(y) = (y & 1) * 5 + (y >> 1),   zp(z) = (z & 1) * 10 + (z >> 1)
// (9 x 9 x 17 halo: 5 even + 4 odd rows, 9 even + 8 odd columns).  A tap's 32 voxels are then (ly, lz) -> const + ly * 24 + lz:
// 8 contiguous slots per row, rows 24 slots = 384 bytes apart (= 128 mod 256), so each of the four lane groups covers the 16 slots
// of a 256-byte bank row exactly once; the tap offsets stay compile-time immediates.  The producers' stores (8-lane groups = 4
// consecutive voxels x 2 k-half planes, 128-byte bank rows) stay conflict-free with the odd columns at slot 10 (= 2 mod 8) and the
// planes' 64-byte skew.  Lanes past the halo store into one dummy slot behind the plane.
#ifndef NS_SW
#define NS_SW 1   // 0: the linear layout for every stride (A/B builds)
#endif
#define NS_SWZ(S) (NS_SW && (S) == 2)
#define NS_SW_ROW 24
#define NS_SW_YODD 5
#define NS_SW_ZODD 10
__host__ __device__ constexpr int ns_sw_yp(int y) { return (y & 1) * NS_SW_YODD + (y >> 1); }
__host__ __device__ constexpr int ns_sw_zp(int z) { return (z & 1) * NS_SW_ZODD + (z >> 1); }
__host__ __device__ constexpr int ns_sw_slots() { return 9 * 9 * NS_SW_ROW; }            // slots of one k-half plane (+ 8 of padding: the dummy slot)
__host__ __device__ constexpr int ns_sw_plane_bytes() { return (ns_sw_slots() + 8) * 16 + 64; }   // = 64 mod 128: the k-half planes' stores interleave

#ifndef NS_PF2
#define NS_PF2 1   // two producer register sets: halo loads a whole chunk interval ahead (see the producer loop); 0 = never
#endif

// prod_commit / prod_commit_x3 (conv_ws_dev.h) with a per-item LDS offset instead of the linear voxel index
template <bool X3, bool SS, bool EDGE>
__device__ __forceinline__ void ns_commit_items_sw(const ConvArgs& p, const ChunkRegs& rg, unsigned char* d0, const int (&off)[WS_MAXV], int plane,
                                                   int nv, unsigned slope2) {
    if (EDGE)
        asm volatile("; commit sw: edge tile");
    else
        asm volatile("; commit sw: interior tile");
#pragma unroll
    for (int j = 0; j < WS_MAXV; ++j) {
        if (j < nv) {
            if constexpr (X3) {
                float y[4] = {__uint_as_float(rg.d[j].x), __uint_as_float(rg.d[j].y), __uint_as_float(rg.d[j].z), __uint_as_float(rg.d[j].w)};
                if (SS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float f = __builtin_fmaf(y[i], __uint_as_float(rg.ssw[2 * i]), __uint_as_float(rg.ssw[2 * i + 1]));
                        y[i] = f > 0.f ? f : f * p.slope;
                    }
                }
                uint2 hi, lo;
                x3_split4(y, hi, lo);
                if (EDGE && !((rg.ok >> j) & 1u)) hi = lo = make_uint2(0, 0);
                *(uint2*)(d0 + off[j]) = hi;
                *(uint2*)(d0 + plane + off[j]) = lo;
            } else {
                uint4 o = rg.d[j];
                if (SS) o = norm_act8_pk(o, rg.ssw, slope2);
                if (EDGE && !((rg.ok >> j) & 1u)) o = make_uint4(0, 0, 0, 0);
                *(uint4*)(d0 + off[j]) = o;
            }
        }
    }
}

template <bool X3>
__device__ __forceinline__ void ns_commit_sw(const ConvArgs& p, ChunkRegs& rg, unsigned char* dst_in, const int (&off)[WS_MAXV], int q, int HV,
                                             int plane, int dbg) {
    const int nv = (dbg & 32) ? 0 : (HV + WS_PROD / 2 - 1) / (WS_PROD / 2);
    if (rg.skip_halo) return;
    union {
        unsigned u;
        h2_t v;
    } sl2;
    sl2.v = h2_t{(_Float16)p.slope, (_Float16)p.slope};
    // fp16: lane q & 1 owns k-half plane q & 1 of its voxels; split precision: hi -> plane 0, lo -> plane 1, lane q & 1 the 8-byte half
    unsigned char* d0 = X3 ? dst_in + (q & 1) * 8 : dst_in + (q & 1) * plane;
    const bool edge = __builtin_amdgcn_ballot_w64(rg.ok != rg.live) != 0;
    if (rg.has_ss) {
        if (edge)
            ns_commit_items_sw<X3, true, true>(p, rg, d0, off, plane, nv, sl2.u);
        else
            ns_commit_items_sw<X3, true, false>(p, rg, d0, off, plane, nv, sl2.u);
    } else {
        if (edge)
            ns_commit_items_sw<X3, false, true>(p, rg, d0, off, plane, nv, sl2.u);
        else
            ns_commit_items_sw<X3, false, false>(p, rg, d0, off, plane, nv, sl2.u);
    }
}

// One 16-channel chunk: acc[r] += sum over the 27 taps.  b0p: LDS address of this lane's voxel in plane 0 of the wave's
// M-tiles, this lane's k-half plane (halo extents are compile-time: every fragment read has an immediate offset); wb: the
// packed weights (wave-uniform base); vcur / vnext: this lane's byte offset of (chunk, tap 0) of this and of the following
// chunk for the wave's cout chunk (always valid: the last chunk of the last tile prefetches a dummy); gs: bytes per tap.
// The accumulators enter a tile's first chunk holding the bias.
// X3 (split-precision mode, see k_conv_ws): the chunk is 8 fp32 channels as hi / lo fp16 planes, the weight fragments come in
// hi / lo pairs (lo = `lo_off` bytes behind hi in the packed array: the part stride Cout * 16) and every (plane, dx) pair is two MFMAs.
// NR: slots of the weight ring (3 or 9: a divisor of the 9 groups, so the slot of a group does not depend on the chunk); group g + NR - 1 is
// fetched while group g is consumed.  NR = 9 for the S = 2, RM = 2 instantiation: its groups are 6 MFMAs (192 cycles) long, two groups ahead
// was less than an L2 round trip (BOA_WS_TRACE: 3 900 / 2 850 cycles per chunk of 54 MFMAs).
template <int S, int RM, bool X3, int NR>
__device__ __forceinline__ void consume_chunk_x(const unsigned char* b0p, f32x16 (&acc)[RM], f16x8 (&a)[NR][3], f16x8 (&al)[X3 ? NR : 1][3],
                                                const WS_GLOBAL unsigned char* wb, unsigned vcur, unsigned vnext, unsigned gs, unsigned lo_off) {
    constexpr int NB = S * (RM - 1) + 3;
    constexpr int H1 = 3 * S + 3, H2 = 7 * S + 3;
    constexpr int MM = X3 ? 2 : 1;
    f16x8 b[NB];
    auto fetch_a = [&](unsigned vchunk, int g, int slot) {  // group g = dy * 3 + dz: taps g + 9 dx
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            unsigned vo = __umul24((unsigned)(g + 9 * dx), gs) + vchunk;
            asm volatile("" : "+v"(vo));  // keep the 32 -> 64 bit extension in this block: scalar-base load form
            a[slot][dx] = *(const WS_GLOBAL f16x8*)(wb + vo);
            if constexpr (X3) {
                unsigned vl = vo + lo_off;
                asm volatile("" : "+v"(vl));
                al[slot][dx] = *(const WS_GLOBAL f16x8*)(wb + vl);
            }
        }
    };
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) b[jj] = *(const f16x8*)(b0p + (NS_SWZ(S) ? jj * 9 * NS_SW_ROW : jj * H1 * H2) * 16);
    __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        const int slot = g % NR;
        if (g + NR - 1 < 9)
            fetch_a(vcur, g + NR - 1, (g + NR - 1) % NR);
        else
            fetch_a(vnext, g + NR - 1 - 9, (g + NR - 1) % NR);
        __builtin_amdgcn_sched_group_barrier(0x020, 3 * MM, 0);
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
            int cnt = 0;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int rr = jj - dx;
                if (rr < 0 || rr % S != 0 || rr / S >= RM) continue;
                const int r = rr / S;
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[slot][dx], b[jj], acc[r], 0, 0, 0);
                if constexpr (X3) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[slot][dx], b[jj], acc[r], 0, 0, 0);
                ++cnt;
            }
            if (cnt == 1)
                __builtin_amdgcn_sched_group_barrier(0x008, 1 * MM, 0);
            else if (cnt == 2)
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * MM, 0);
            else if (cnt == 3)
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * MM, 0);
            if (g + 1 < 9) {
                const int gn = g + 1, dy = gn / 3, dz = gn % 3;
                b[jj] = *(const f16x8*)(b0p + (NS_SWZ(S) ? (jj * 9 + ns_sw_yp(dy)) * NS_SW_ROW + ns_sw_zp(dz) : (jj * H1 + dy) * H2 + dz) * 16);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
}

// S: conv stride (all axes); WN: consumer waves along the cout axis (cout group = WN chunks), 4 / WN wave rows along x;
// RM: M-tiles (x-planes of 4 x 8 output voxels) per wave.  Block tile = (4 / WN) * RM planes.
template <int S, int WN, int RM, bool X3>
__global__ __launch_bounds__(WS_THREADS) void k_conv_ns(ConvArgs p, int dbg_arg, const int* __restrict__ desc, int desc_row) {
#ifdef WS_WITH_TRACE
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;   // (ablation switches: traced build only, see k_conv_ws)
    (void)dbg_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    const int HV = p.h0 * p.h1 * p.h2;
    const int plane = NS_SWZ(S) ? ns_sw_plane_bytes() : ws_plane_bytes(HV);
    const int ncc = (p.C0 + p.C1) / 16;
    const int buf_bytes = 2 * plane;  // LDS: [halo buf 0][halo buf 1], each two k-octet planes
    unsigned char* bufs = smem;

    // this workgroup's row of the launch's descriptor table (k_ws_build_desc, conv_ws.hip): entry 0 = its tile count, entry 1 + k =
    // its k-th tile; each role reads one 32-byte descriptor per tile with scalar loads instead of walking the run tables
    const int* __restrict__ drow = desc + (size_t)blockIdx.x * desc_row * 8;
    const int my_tiles = drow[0];
    const int my_chunks = my_tiles * ncc;
    for (int i = tid; i < p.Cout; i += WS_THREADS) ((float*)(smem + 2 * buf_bytes))[i] = p.bias[i];  // (ordered by the first barrier)
    const int tr_blk = WS_TRACING ? (int)p.trace[WS_TRACE_SLOTS - 5] : -1;   // (debug) the traced block: BOA_WS_TRACE=<block>

    if (producer) {
        // ---- producer waves: as in k_conv_ws without the weight staging (chunk g + 1 committed while the consumers work on
        // chunk g, its global loads issued one barrier interval earlier)
        const int q = tid - 256;
        const ProdConst pc = prod_const(p, q, HV);
        // SW layout: this lane's LDS byte offset of its halo voxels (kernel constants, like pc.rel)
        int ldso[WS_MAXV];
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) {
            const int v = (q >> 1) + (WS_PROD / 2) * j;
            const int hz = v % 17, t = v / 17;
            const int hy = t % 9, hx = t / 9;
            ldso[j] = (v < HV ? (hx * 9 + ns_sw_yp(hy)) * NS_SW_ROW + ns_sw_zp(hz) : ns_sw_slots()) * 16;
        }
        TileDesc pd;   // the tile whose chunks are being issued
        pd.flags = pd.vo = pd.ibase = 0;
        TileCoord& ptc = pd.tc;
        int pk = 0;
        ptc.n = ptc.cy = ptc.ox0 = ptc.oy0 = ptc.oz0 = ptc.sp = 0;
        ProdItems items;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) items.gi[j] = 0;
        items.ok = 0;
        // Two register sets (PF2): the loads of chunk g + 2 are issued at the HEAD of the interval, before chunk g + 1 is committed, so
        // they have a whole barrier interval to land.  The stride-2 layers stage 8 input voxels per output voxel and are bound by
        // the producers' HBM / L2 round trips (trace of an interior workgroup, 32 -> 64 @128^3: 330 + 1 220 cycles per tile waiting for
        // loads that were issued behind the previous commit); k_conv_ns has no weight DMA whose vmcnt(0) would also wait for
        // them (what made the same change a loss in k_conv_ws).  The interval loop is unrolled by two so that the sets alternate
        // without copies.  A/B on one box (tools/ab_layers.sh, batch 8): 64 -> 128 @64^3 -5 %, 128 -> 256 @32^3 -8 %, 256 -> 320 @16^3
        // -8 %; the 32 -> 64 layer at 128^3 (RM = 2: HBM-cold input, 3.6 TB/s of halo traffic) +16 % -- twice the loads in flight per
        // CU there only deepen the queue in front of the memory side -- so that instantiation keeps one set.
#ifdef NS_PF2_ALL
        constexpr bool PF2 = NS_PF2 != 0;   // (A/B builds)
#else
        constexpr bool PF2 = NS_PF2 && RM == 4;
#endif
        ChunkRegs rgA, rgB;
        auto clear_regs = [&](ChunkRegs& rg) {
#pragma unroll
            for (int j = 0; j < WS_MAXV; ++j) rg.d[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) rg.ssw[j] = 0;
            rg.ok = rg.live = 0;
            rg.has_ss = 0;
            rg.skip_halo = 0;
            rg.cc = rg.cy = 0;
        };
        clear_regs(rgA);
        clear_regs(rgB);
        int pcc = 0;
        int ptr_n = 0;
        // debug timeline of producer wave 4 of the traced block (second half of the trace buffer): 1 barrier passed, 8 next tile set up,
        // 7 loads landed, 2 chunk committed, 3 next loads issued
#define NS_PSTAMP(code)                                                                                                          \
    do {                                                                                                                         \
        if (WS_TRACING && (int)blockIdx.x == tr_blk && wave == 4 && lane == 0 && ptr_n < WS_TRACE_SLOTS / 2 - 8) {                                  \
            p.trace[WS_TRACE_SLOTS / 2 + ptr_n] = ((unsigned long long)(code) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); \
            ++ptr_n;                                                                                                             \
        }                                                                                                                        \
    } while (0)
        const bool live = !(dbg & 2);
        auto setup_next = [&]() {   // descriptor + halo addresses of the next tile (before its first chunk is issued)
            pd = load_desc(drow, ++pk);
            prod_setup_desc(p, pd, pc, items);
            NS_PSTAMP(8);
        };
        auto issue = [&](ChunkRegs& rg) {
            prod_issue(p, ptc, items, pc.in_halo, pcc, false, q, dbg, rg);
            if (++pcc == ncc) pcc = 0;
            NS_PSTAMP(3);
        };
        auto commit = [&](ChunkRegs& rg, unsigned char* dst) {
            if (WS_TRACING) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                NS_PSTAMP(7);
            }
            if constexpr (NS_SWZ(S))
                ns_commit_sw<X3>(p, rg, dst, ldso, q, HV, plane, dbg);
            else if constexpr (X3)
                prod_commit_x3(p, rg, dst, q, HV, plane, dbg);
            else
                prod_commit(p, rg, dst, q, HV, plane, dbg);
            NS_PSTAMP(2);
        };
        if (live && my_chunks > 0) {
            pd = load_desc(drow, 0);
            prod_setup_desc(p, pd, pc, items);
            prod_issue(p, ptc, items, pc.in_halo, 0, false, q, dbg, rgA);
            if (++pcc == ncc) pcc = 0;
        }
        if constexpr (PF2) {
            auto interval = [&](int g, ChunkRegs& cur, ChunkRegs& nxt) {
                if (live && g + 1 < my_chunks) {
                    NS_PSTAMP(1);
                    if (g + 2 < my_chunks) {
                        if (pcc == 0) setup_next();
                        issue(nxt);
                    }
                    commit(cur, bufs + ((g + 1) & 1) * buf_bytes);
                }
                __syncthreads();
            };
            for (int g = -1; g < my_chunks; g += 2) {
                interval(g, rgA, rgB);
                if (g + 1 < my_chunks) interval(g + 1, rgB, rgA);
            }
        } else {
            for (int g = -1; g < my_chunks; ++g) {
                if (live && g + 1 < my_chunks) {
                    NS_PSTAMP(1);
                    const bool do_issue = g + 2 < my_chunks;
                    if (do_issue && pcc == 0) setup_next();   // (before the commit: the loads go out right behind it, see k_conv_ws)
                    commit(rgA, bufs + ((g + 1) & 1) * buf_bytes);
                    if (do_issue) issue(rgA);
                }
                __syncthreads();
            }
        }
        return;
    }

    // ---- consumer waves --------------------------------------------------------------------------------
    if (!(dbg & 1024)) __builtin_amdgcn_s_setprio(3);
    int tr_n = 0;
    // debug timeline (BOA_WS_TRACE): consumer wave 0 of block 0; codes 1 tile start, 2 before the first chunk, 3 chunk done, 4 barrier passed
#define NS_STAMP(code)                                                                                                   \
    do {                                                                                                                 \
        if (WS_TRACING && (int)blockIdx.x == tr_blk && wave == 0 && lane == 0 && tr_n < WS_TRACE_SLOTS / 2) {                              \
            p.trace[tr_n] = ((unsigned long long)(code) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); \
            ++tr_n;                                                                                                      \
        }                                                                                                                \
    } while (0)
    const int cw = wave & 3;
    const int wn = cw % WN, wm = cw / WN;
    const int ly = l31 >> 3, lz = l31 & 7;
    const int nchunks_out = p.Cout / 32;
    // this lane's voxel in plane 0 of the wave's M-tiles, its k-half plane
    constexpr int H1 = 3 * S + 3, H2 = 7 * S + 3;  // halo extents along y, z (host: conv_ns_tile)
    const int hoff = (NS_SWZ(S) ? ((2 * (wm * RM)) * 9 + ly) * NS_SW_ROW + lz : ((S * (wm * RM)) * H1 + S * ly) * H2 + S * lz) * 16 + kh * plane;
    const int srel0 = ly * p.Wo + lz;
    const size_t out_vox = (size_t)p.Do * p.Ho * p.Wo;
    const int nslots = p.nslots;
    int slot = 0;
    // InstanceNorm partial sums in the D-fragment layout (see k_conv_ws)
    float st_s[16], st_q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st_s[i] = st_q[i] = 0.f;
    int st_n = -1, st_ch = 0;
    auto flush_stats = [&]() {
        if (st_n < 0) return;
#define NS_HALVE(M, HALF)                                                                                \
    {                                                                                                    \
        const bool up = (l31 & (M)) != 0;                                                                \
        _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                                             \
            const float ks = up ? st_s[i + (HALF)] : st_s[i], gs_ = up ? st_s[i] : st_s[i + (HALF)];     \
            const float kq = up ? st_q[i + (HALF)] : st_q[i], gq_ = up ? st_q[i] : st_q[i + (HALF)];     \
            st_s[i] = ks + __shfl_xor(gs_, (M));                                                         \
            st_q[i] = kq + __shfl_xor(gq_, (M));                                                         \
        }                                                                                                \
    }
        NS_HALVE(1, 8)
        NS_HALVE(2, 4)
        NS_HALVE(4, 2)
        NS_HALVE(8, 1)
#undef NS_HALVE
        st_s[0] += __shfl_xor(st_s[0], 16);
        st_q[0] += __shfl_xor(st_q[0], 16);
        if (l31 < 16) {
            const int i = ((l31 & 1) << 3) | ((l31 & 2) << 1) | ((l31 & 4) >> 1) | ((l31 & 8) >> 3);
            const int row = st_ch * 32 + 8 * (i >> 2) + 4 * kh + (i & 3);
            float* pp = p.partials + (((size_t)st_n * p.Cout + row) * 2) * nslots + slot;
            pp[0] = st_s[0];
            pp[nslots] = st_q[0];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) st_s[i] = st_q[i] = 0.f;
    };

    f32x16 acc[RM];
    auto epilogue = [&](const TileCoord& tc, int ch, int dfl, int dvo) {
        st_n = tc.n;  // (the caller flushed when the sample / cout chunk / run changed)
        st_ch = ch;
        const int cout0 = ch * 32;
        const bool full = (dfl & WS_DF_FULL) != 0;   // (block tile = 4 x 4 x 8 = b x w of conv_ns_tile: make_desc's test)
        const size_t ovox = (size_t)(unsigned)dvo;
        const size_t obase = ((size_t)tc.n * p.Cout + cout0) * out_vox + ovox * 16;
        const unsigned olane = ((unsigned)kh * (unsigned)out_vox + (unsigned)srel0) * 32u;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int mx = wm * RM + r;  // wave-uniform plane within the block tile
            const int mrel = mx * p.Ho * p.Wo;
            bool ok = true;
            if (!full) ok = tc.ox0 + mx < p.Do && tc.oy0 + ly < p.Ho && tc.oz0 + lz < p.Wo;
            const float dm = ok ? 1.f : 0.f;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = X3 ? acc[r][i] * p.winv : acc[r][i];
            // packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two entries per instruction; same operations and order per entry)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f2_t vm = f2_t{v[2 * i], v[2 * i + 1]};
                if (!full) vm = vm * f2_t{dm, dm};
                f2_t s2 = f2_t{st_s[2 * i], st_s[2 * i + 1]}, q2 = f2_t{st_q[2 * i], st_q[2 * i + 1]};
                s2 = s2 + vm;
                q2 = __builtin_elementwise_fma(vm, vm, q2);
                st_s[2 * i] = s2.x; st_s[2 * i + 1] = s2.y;
                st_q[2 * i] = q2.x; st_q[2 * i + 1] = q2.y;
            }
            if constexpr (X3) {   // fp32 octet planes (k_conv_ws<X3>'s store)
                if (ok && !(dbg & 4)) {
                    const size_t doff = ((size_t)tc.n * p.Cout + cout0) * out_vox * 4 + (ovox + (size_t)mrel) * 32;
                    const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)doff), dhi = __builtin_amdgcn_readfirstlane((unsigned)(doff >> 32));
                    WS_GLOBAL unsigned char* dst = sgpr_ptr((const unsigned char*)p.out + (((size_t)dhi << 32) | dlo));
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        unsigned ol = ((unsigned)srel0 * 32u + (unsigned)kh * 16u) + (unsigned)gq * ((unsigned)out_vox * 32u);
                        asm volatile("" : "+v"(ol));
#ifndef WS_TEMPORAL_STORES
                        __builtin_nontemporal_store(f32x4_t{v[gq * 4 + 0], v[gq * 4 + 1], v[gq * 4 + 2], v[gq * 4 + 3]}, (WS_GLOBAL f32x4_t*)(dst + ol));
#else
                        *(WS_GLOBAL f32x4_t*)(dst + ol) = f32x4_t{v[gq * 4 + 0], v[gq * 4 + 1], v[gq * 4 + 2], v[gq * 4 + 3]};
#endif
                    }
                }
                continue;
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned w[8];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float lo4[4], hi4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[pr * 4 + e]), __float_as_uint(v[(pr + 2) * 4 + e]), false, false);
                    lo4[e] = __uint_as_float(sw[0]);
                    hi4[e] = __uint_as_float(sw[1]);
                }
                w[pr * 4 + 0] = cvt_pk_h2(lo4[0], lo4[1]);
                w[pr * 4 + 1] = cvt_pk_h2(lo4[2], lo4[3]);
                w[pr * 4 + 2] = cvt_pk_h2(hi4[0], hi4[1]);
                w[pr * 4 + 3] = cvt_pk_h2(hi4[2], hi4[3]);
            }
            if (ok && !(dbg & 4)) {
                WS_GLOBAL unsigned char* dst = sgpr_ptr(p.out + (obase + (size_t)mrel * 16));
                unsigned ol = olane;
                asm volatile("" : "+v"(ol));
#ifndef WS_TEMPORAL_STORES
                // non-temporal stores: the layer's output is not read again before the launch ends, so it need not displace halo lines from
                // the XCD's L2 (A/B on one box with tools/build_alt.sh, layers repeated at the power cap: 1 780 -> 1 767, 957 -> 946,
                // 823 -> 817 us; bench step 1 820 -> 1 806 ms; the HBM-bound first conv measured 0 ... -10 % with them and keeps plain stores)
                __builtin_nontemporal_store(u32x4_t{w[0], w[1], w[2], w[3]}, (WS_GLOBAL u32x4_t*)(dst + ol));
                __builtin_nontemporal_store(u32x4_t{w[4], w[5], w[6], w[7]}, (WS_GLOBAL u32x4_t*)(dst + ol + 16));
#else
                *(WS_GLOBAL u32x4_t*)(dst + ol) = u32x4_t{w[0], w[1], w[2], w[3]};
                *(WS_GLOBAL u32x4_t*)(dst + ol + 16) = u32x4_t{w[4], w[5], w[6], w[7]};
#endif
            }
        }
    };

    // weights: wpk [(cc * 27 + tap) * 2 + kh][Cout][8 halves]; a wave's fragment of (cc, tap, cout chunk ch): lane (kh, l31)
    // reads 16 bytes at ((cc * 27 + tap) * 2 * Cout + kh * Cout + ch * 32 + l31) * 16
    const unsigned gs = 32u * (unsigned)p.Cout;  // bytes per tap
    const unsigned voff = X3 ? (unsigned)l31 * 16u : ((unsigned)kh * (unsigned)p.Cout + (unsigned)l31) * 16u;   // (X3: both k-halves read the same hi / lo fragments)
    const unsigned lo_off = (unsigned)p.Cout * 16u;
    auto woff = [&](int cc, int ch) -> unsigned { return voff + (unsigned)cc * 27u * gs + (unsigned)ch * 512u; };
    constexpr int NR = (S == 2 && RM == 2 && !X3) ? 9 : 3;
    f16x8 a[NR][3];
    f16x8 al[X3 ? NR : 1][3];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[i][j][e] = (_Float16)0.f;
                if (X3 || i == 0) al[X3 ? i : 0][j][e] = (_Float16)0.f;
            }
    auto prime = [&](unsigned vchunk) {  // groups 0 .. NR - 2 of a chunk into their ring slots
        const WS_GLOBAL unsigned char* wb = sgpr_ptr(p.wpk);
#pragma unroll
        for (int g = 0; g < NR - 1; ++g)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                unsigned vo = __umul24((unsigned)(g + 9 * dx), gs) + vchunk;
                asm volatile("" : "+v"(vo));
                a[g][dx] = *(const WS_GLOBAL f16x8*)(wb + vo);
                if constexpr (X3) {
                    unsigned vl = vo + lo_off;
                    asm volatile("" : "+v"(vl));
                    al[g][dx] = *(const WS_GLOBAL f16x8*)(wb + vl);
                }
            }
    };

    TileDesc cd;
    cd.flags = cd.vo = cd.ibase = 0;
    cd.tc.n = cd.tc.cy = cd.tc.ox0 = cd.tc.oy0 = cd.tc.oz0 = cd.tc.sp = 0;
    bool ring_valid = false;
    TileCoord done_tc = cd.tc;
    int done_ch = 0, done_fl = 0, done_vo = 0;
    bool done_active = false;
    __syncthreads();  // chunk 0 and the bias table staged
    for (int k = 0; k < my_tiles + 1; ++k) {
        const bool more = k < my_tiles;
        bool new_run = true;
        NS_STAMP(1);
        if (more) {
            cd = load_desc(drow, k);
            new_run = (cd.flags & WS_DF_NEWRUN) != 0;
        }
        const bool have_next = k + 1 < my_tiles;
        const TileCoord& tc = cd.tc;
        const int ch = tc.cy * WN + wn;
        if (k > 0 && done_active && !(dbg & 8)) epilogue(done_tc, done_ch, done_fl, done_vo);  // deferred: the previous tile's
        // ONE flush site: the accumulated partial sums go to their slot when the next tile belongs to another virtual
        // workgroup, sample or cout chunk, and at the end
        if (k > 0 && (!more || new_run || tc.n != st_n || ch != st_ch)) {
            flush_stats();
            st_n = -1;
        }
        if (!more) break;
        if (new_run) slot = (int)((unsigned)cd.flags >> 16) * 4 + cw;
        const bool active = ch < nchunks_out;
        // the chunk the ring is prefetched for after this tile: the next tile's (clamped to a valid chunk when this wave idles there)
        // (the cout group of the tile after this one: the weight ring is prefetched across tile boundaries)
        int ch_next = have_next ? drow[(size_t)(k + 2) * 8 + 1] * WN + wn : ch;
        const bool next_active = have_next && ch_next < nchunks_out;
        if (ch_next >= nchunks_out) ch_next = nchunks_out - 1;
        if (active) {
            // the accumulators start at the bias: this lane's 16 biases of the cout chunk in the D-fragment layout (entry 4 gq + e
            // <-> cout 8 gq + 4 kh + e) from the LDS copy (no bias add in the epilogue, no bias registers across the tile)
            const float* lb = (const float*)(smem + 2 * buf_bytes) + ch * 32 + 4 * kh;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4_t bv = *(const f32x4_t*)(lb + 8 * gq);
#pragma unroll
                for (int r = 0; r < RM; ++r) {
                    acc[r][gq * 4 + 0] = X3 ? bv[0] * p.wscale : bv[0]; acc[r][gq * 4 + 1] = X3 ? bv[1] * p.wscale : bv[1];
                    acc[r][gq * 4 + 2] = X3 ? bv[2] * p.wscale : bv[2]; acc[r][gq * 4 + 3] = X3 ? bv[3] * p.wscale : bv[3];
                }
            }
            if (!ring_valid) prime(woff(0, ch));
        }
        NS_STAMP(2);
        for (int cc = 0; cc < ncc; ++cc) {
            const int g = k * ncc + cc;
            if (active) {
                const unsigned char* cur = bufs + (g & 1) * buf_bytes;
                int ho = hoff;
                asm volatile("" : "+v"(ho));
                const WS_GLOBAL unsigned char* wb = sgpr_ptr(p.wpk);
                const unsigned vcur = woff(cc, ch);
                const unsigned vnext = cc + 1 < ncc ? woff(cc + 1, ch) : woff(0, ch_next);
                consume_chunk_x<S, RM, X3, NR>(cur + ho, acc, a, al, wb, vcur, vnext, gs, lo_off);
            }
            NS_STAMP(3);
            __syncthreads();
            NS_STAMP(4);
        }
        ring_valid = active && next_active;
        done_tc.n = __builtin_amdgcn_readfirstlane(tc.n);
        done_tc.cy = __builtin_amdgcn_readfirstlane(tc.cy);
        done_tc.sp = __builtin_amdgcn_readfirstlane(tc.sp);
        done_tc.ox0 = __builtin_amdgcn_readfirstlane(tc.ox0);
        done_tc.oy0 = __builtin_amdgcn_readfirstlane(tc.oy0);
        done_tc.oz0 = __builtin_amdgcn_readfirstlane(tc.oz0);
        done_ch = __builtin_amdgcn_readfirstlane(ch);
        done_fl = __builtin_amdgcn_readfirstlane(cd.flags);
        done_vo = __builtin_amdgcn_readfirstlane(cd.vo);
        done_active = active;
    }
}

// ---- host side ---------------------------------------------------------------------------------------
const int* ws_run_table(boa_ctx* ctx, const ConvArgs& a, int tiles_per_sample, int vw);
const int* ws_desc_table(boa_ctx* ctx, const ConvArgs& a, int tiles_per_sample, int grid, int* row_out);
int conv_ws_vw(int tiles_per_sample, int cu_count);

static size_t ns_plane_host(int HV) { return ((size_t)(HV + WS_PROD / 2 - 1) / (WS_PROD / 2)) * (WS_PROD / 2) * 16 + 64; }

// The layers k_conv_ns takes (a function of the layer geometry only): 3x3x3 kernels with stride 2 on all axes, at least two cout
// chunks and an output of at least 8^3 voxels -- measured per layer against k_conv_ws on the `total` geometry (8 tiles): 32 -> 64
// @128^3, 64 -> 128 @64^3 453 -> 194 us, 128 -> 256 @32^3 224 -> 96 us, 256 -> 320 @16^3 114 -> 81 us.  The stride-1 layers with
// Cout >= 128 run at par with k_conv_ws (the epilogue of four M-tiles per 128-voxel tile costs what the fourfold halo reuse
// gains) and the 8^3 / 4^3 layers are latency-bound either way; BOA_NS_ALL=1 sends them here as well (experiments).
bool conv_ns_applicable(const ConvGeom& g) {
    static const bool off = getenv("BOA_NO_NS") != nullptr;
    static const bool all = getenv("BOA_NS_ALL") != nullptr;
    if (off) return false;
    const bool k333 = g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3;
    const bool s1 = g.s[0] == 1 && g.s[1] == 1 && g.s[2] == 1, s2 = g.s[0] == 2 && g.s[1] == 2 && g.s[2] == 2;
    if (!k333 || g.Cout % 32 != 0) return false;
    if (all) return (s1 || s2) && g.Cout >= 128 && g.Do >= 2 && g.Ho >= 2 && g.Wo >= 4;
    return s2 && g.Cout >= 64 && g.Do >= 8 && g.Ho >= 8 && g.Wo >= 8;
}

// consumer waves along the cout axis: four when the layer has four or more cout chunks, else two (Cout = 64 / 96)
static int ns_wn(int Cout) { return Cout >= 128 ? 4 : 2; }

void conv_ns_tile(const ConvGeom& g, ConvTile* t) {
    t->variant = 2;
    t->xs = 0;   // (its own LDS layouts; ConvArgs::xs = h1 * h2 for the shared producer constants)
    t->R = 4;
    t->w[0] = 1; t->w[1] = 4; t->w[2] = 8;
    t->b[0] = 4; t->b[1] = 1; t->b[2] = 1;
    const int ext[3] = {4, 4, 8};
    const int dims[3] = {g.Do, g.Ho, g.Wo};
    size_t HV = 1;
    for (int d = 0; d < 3; ++d) {
        t->h[d] = (ext[d] - 1) * g.s[d] + 3;
        t->tiles[d] = (dims[d] + ext[d] - 1) / ext[d];
        HV *= (size_t)t->h[d];
    }
    // two halo buffers (stride 2: the de-interleaved layout, see NS_SW_ROW) + the bias table
    t->lds_bytes = 4 * (NS_SWZ(g.s[0]) ? (size_t)ns_sw_plane_bytes() : ns_plane_host((int)HV)) + (size_t)g.Cout * sizeof(float);
}

int conv_ns_ncy(int Cout) { return (Cout / 32 + ns_wn(Cout) - 1) / ns_wn(Cout); }

template <int S, int WN, int RM, bool X3 = false>
static void launch_ns(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, int grid, const int* desc, int desc_row) {
    static bool once = (hipFuncSetAttribute((const void*)k_conv_ns<S, WN, RM, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL((k_conv_ns<S, WN, RM, X3>), dim3(grid), dim3(WS_THREADS), t.lds_bytes, ctx->stream, a,
                       getenv("BOA_WS_DBG") ? atoi(getenv("BOA_WS_DBG")) : 0, desc, desc_row);
}

int launch_conv_ns(boa_ctx* ctx, const ConvArgs& a_in, const ConvTile& t, double flops, double bytes, bool x3) {
    ConvArgs a = a_in;
    a.ncy = conv_ns_ncy(a.Cout);
    const int total = t.tiles[0] * t.tiles[1] * t.tiles[2] * a.ncy;  // tiles of one sample
    const int vw = conv_ws_vw(total, ctx->cu_count);
    const int grid = (int)std::min<long long>((long long)vw * a.N, ctx->cu_count);
    BOA_REQUIRE((double)a.Di * a.Hi * a.Wi <= 16777216.0, "conv_ns: more than 2^24 input voxels per sample (24-bit offset multiply)");
    BOA_REQUIRE((double)a.Di * a.Hi * a.Wi * std::max(a.C0, a.C1) * 2.0 < 4294967296.0,
                "conv_ns: one sample of the input exceeds 4 GiB (32-bit voxel offsets)");
    BOA_REQUIRE(2 * t.h[0] * t.h[1] * t.h[2] <= WS_PROD * WS_MAXV, "conv_ns: halo too large");
#ifdef WS_WITH_TRACE
    static const bool want_trace = getenv("BOA_WS_TRACE") != nullptr;
#else
    static const bool want_trace = false;   // (the stamps are compiled out of the production build: conv_ws_dev.h)
#endif
    a.trace = nullptr;
    if (want_trace) {
        hipMalloc(&a.trace, WS_TRACE_SLOTS * 8);
        hipMemsetAsync(a.trace, 0, WS_TRACE_SLOTS * 8, ctx->stream);
        const unsigned long long blk = (unsigned long long)std::min(std::max(atoi(getenv("BOA_WS_TRACE")), 0), grid - 1);
        hipMemcpyAsync(a.trace + WS_TRACE_SLOTS - 5, &blk, 8, hipMemcpyHostToDevice, ctx->stream);
    }
    a.nslots = 4 * vw;
    a.vw = vw;
    a.vstep_n = grid / vw;
    a.vstep_j = grid % vw;
    a.cy_fast = 0;
    a.runs = ws_run_table(ctx, a, total, vw);
    BOA_REQUIRE(a.runs != nullptr, "conv_ns: could not allocate the run table");
    int desc_row = 0;
    const int* desc = ws_desc_table(ctx, a, total, grid, &desc_row);
    BOA_REQUIRE(desc != nullptr, "conv_ns: could not allocate the tile descriptor table");
    KernelTimer tm(ctx, BOA_K_CONV_MFMA, flops, bytes);
    ctx->counters[x3 ? BOA_CNT_CONV_X3 : BOA_CNT_CONV_WS]++;
    if (x3) {
        BOA_REQUIRE(a.s0 == 2, "conv_ns: the split-precision instantiations are stride 2 only");
        if (ns_wn(a.Cout) == 2)
            launch_ns<2, 2, 2, true>(ctx, a, t, grid, desc, desc_row);
        else
            launch_ns<2, 4, 4, true>(ctx, a, t, grid, desc, desc_row);
    } else if (ns_wn(a.Cout) == 2) {
        BOA_REQUIRE(a.s0 == 2, "conv_ns: the two-chunk cout group is instantiated for stride 2 only");
        launch_ns<2, 2, 2>(ctx, a, t, grid, desc, desc_row);
    } else if (a.s0 == 1)
        launch_ns<1, 4, 4>(ctx, a, t, grid, desc, desc_row);
    else
        launch_ns<2, 4, 4>(ctx, a, t, grid, desc, desc_row);
    tm.stop();
    if (want_trace && a.trace) {
        static unsigned long long host[WS_TRACE_SLOTS];
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(host, a.trace, sizeof(host), hipMemcpyDeviceToHost);
        hipFree(a.trace);
        fprintf(stderr, "[ns-trace] Cin=%d Cout=%d in=%d s=%d:", a.C0 + a.C1, a.Cout, a.Di, a.s0);
        for (int i = 1; i < 80 && host[i]; ++i)
            fprintf(stderr, " %d:%llu", (int)(host[i] >> 56), (host[i] & 0x00ffffffffffffffull) - (host[i - 1] & 0x00ffffffffffffffull));
        fprintf(stderr, "\n[ns-trace] producer:");
        const unsigned long long* hp = host + WS_TRACE_SLOTS / 2;
        for (int i = 41; i < 110 && hp[i]; ++i)
            fprintf(stderr, " %d:%llu", (int)(hp[i] >> 56), (hp[i] & 0x00ffffffffffffffull) - (hp[i - 1] & 0x00ffffffffffffffull));
        fprintf(stderr, "\n");
    }
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
