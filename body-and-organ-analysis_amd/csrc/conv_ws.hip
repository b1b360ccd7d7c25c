// k_conv_ws: wave-specialised, persistent implicit-GEMM conv for gfx950.
//
// One 512-thread workgroup per CU with two roles that run SEPARATE loops and meet at one __syncthreads() per
// 16-channel chunk:
//   waves 0-3 ("consumers") read A (weights) / B (input voxels) fragments from LDS and issue v_mfma_f32_32x32x16_f16
//     (compile-time taps, fully unrolled, fragments of tap t+1 in flight while tap t's MFMAs issue); after a tile's
//     last chunk they add the bias, accumulate the InstanceNorm partial sums in registers, transpose the D fragment
//     into voxel records with v_permlane32_swap and store 2 x 16 bytes per lane;
//   waves 4-7 ("producers") gather the next chunk of the input halo tile from HBM/L2 (loads issued one barrier interval
//     ahead: prod_issue / prod_commit), apply the producer layer's deferred InstanceNorm + LeakyReLU in packed fp16 and
//     write it (plus, when they do not fit resident, the chunk's weights) into the other LDS buffer.
// Workgroups are persistent and walk a contiguous run of output tiles (x fastest; cout chunk fastest for the
// 2-chunk / 2-cout-chunk case, where the staged halo is reused), so the staging of a tile's first chunk hides under
// the previous tile's last chunk and the chip-wide working set is a contiguous run of tiles (halo reuse in L2).
// Per-tile bookkeeping is ONE scalar load per role: the tile sequence of every workgroup (conv_ws_dev.h: virtual workgroups, run
// tables) is walked once per (layer geometry, batch, grid) by k_ws_build_desc below into a table of 32-byte descriptors (tile
// coordinates, statistics slot, halo class and faces, output offset).  A role is a single wave per SIMD: it issues an instruction
// every ~4.7 cycles and pays ~25 cycles per SGPR-spill reload and ~22 per taken branch (tools/issue_rate.hip), so what the role
// loops carry per tile -- not the MFMA pipe -- sets the tile time (profiles/r05_conv_ws_experiments.txt).
//
// Same arithmetic as k_conv_mfma (conv.hip).  Statistics: partials[n][cout][2][nslots], one slot per (workgroup,
// consumer wave), written once per (sample, cout chunk) a wave works on; the table is zero elsewhere (zeroed at
// allocation, re-zeroed by k_norm_finalize).
// Debug (TRACED build only, -DWS_WITH_TRACE: tools/build_alt.sh trace -DWS_WITH_TRACE; the production build compiles both out of the
// role loops): `dbg` bits (BOA_WS_DBG) skip stages for ablation -- 2 producers, 4 output stores, 8 epilogue, 16 weight staging,
// 32 halo commit, 64 transform, 128 halo loads, 256 / 1024 default wave priorities off, 2048 producers first, 4096 consumers without
// fragment reads and MFMAs; results are then wrong by design (except the priority bits).
// BOA_WS_TRACE=<block> records s_memtime stamps of that workgroup (consumer wave 0: 4 chunk start, 5 MFMA loop done, 6 before the
// barrier, with -DWS_TRACE_EPILOGUE 13 / 14 loop top, 9 - 12 epilogue; producer wave 4: 1 barrier passed, 7 loads landed, 2 committed,
// 8 tile set up, 3 next loads issued).
#include <stdlib.h>

#include "conv.h"


#ifndef WS_PF2
#define WS_PF2 0   // (experiment, measured slower) two producer register sets: halo loads issued a whole chunk interval ahead
#endif

#include "conv_ws_dev.h"

// Builds the descriptor rows of a launch: thread b walks physical workgroup b's tile sequence with the kernels' own code.
__global__ __launch_bounds__(64) void k_ws_build_desc(ConvArgs p, int grid, int row, int* __restrict__ out) {
    const int b = (int)blockIdx.x * 64 + (int)threadIdx.x;
    if (b >= grid) return;
    int* r = out + (size_t)b * row * 8;
    int my = 0;
    for (int v = b; v < p.N * p.vw; v += grid) my += p.runs[(v % p.vw) * 8];
    if (my > row - 1) my = row - 1;   // (unreachable: ws_desc_table checks every workgroup's tile count against the row length on the host before the launch)
    r[0] = my;
    for (int i = 1; i < 8; ++i) r[i] = 0;
    TileSeq s;
    s.n = s.left = s.j = 0;
    for (int k = 0; k < my; ++k) {
        bool new_run = true;
        if (k == 0)
            seq_first_of(p, s, b);
        else
            new_run = seq_next(p, s);
        const TileDesc d = make_desc(p, s, new_run);
        int* e = r + (size_t)(k + 1) * 8;
        e[0] = d.tc.n; e[1] = d.tc.cy; e[2] = d.tc.ox0; e[3] = d.tc.oy0; e[4] = d.tc.oz0; e[5] = d.flags; e[6] = d.vo; e[7] = d.ibase;
    }
}

// ---- consumer ----------------------------------------------------------------------------------------
// One 16-channel chunk: acc[r] += W[tap] x X[tap][r] for all taps.  bp[r]: LDS address of this lane's voxel of
// M-tile r in this lane's k-half plane; ap: LDS address of this lane's row of the first A fragment.
#ifndef WS_PF
#define WS_PF 1  // fragment prefetch distance in taps (2 measured slower: 610 vs 664 TFLOP/s on 32->32 @128^3, batch 8)
#endif

// X3 (split-precision mode, conv_x3 notes below): the chunk holds 8 fp32 channels as [hi plane | lo plane] fp16, the B fragment of a
// lane is the hi (k-half 0) or lo (k-half 1) part of its voxel's 8 channels, and a tap is TWO MFMAs: [Wh | Wh] x [Xh ; Xl] and
// [Wl | Wl] x [Xh ; Xl] -- all four cross terms of (Wh + Wl)(Xh + Xl) in fp32 accumulators.  Both k-halves of a lane pair read the
// same weight fragment (hi at ap + tap KiB, lo 512 B behind it).
template <int R, int K0, int K1, int K2, bool FIRST, bool X3>
__device__ __forceinline__ void consume_chunk(const unsigned char* const (&bp)[R], const unsigned char* ap, int xs, int h2,
                                              f32x16 (&acc)[R], const f32x16& c0) {
    // Software pipeline over the (compile-time) taps: the A/B fragments of tap t + WS_PF are read while the MFMAs of
    // tap t issue.  The sched_group_barrier sequence pins that order: [PF x (R+1) reads] then per tap
    // [(R+1) reads][R MFMAs].  The prologue groups matter: without them the per-tap groups are filled one tap late
    // and every tap waits for reads it has just issued.
    constexpr int T = K0 * K1 * K2;
    constexpr int NS = WS_PF + 1;
    f16x8 a[NS];
    f16x8 al[X3 ? NS : 1];
    f16x8 b[NS][R];
    constexpr int NRD = R + (X3 ? 2 : 1);   // LDS reads per tap
    auto fetch = [&](int tn, int slot) {
        const int dzn = tn % K2, dyn = (tn / K2) % K1, dxn = tn / (K2 * K1);
        const int off = (dxn * xs + dyn * h2 + dzn) * 16;   // (xs: voxels between x-planes, ConvTile::xs)
        a[slot] = *(const f16x8*)(ap + tn * 1024);
        if constexpr (X3) al[slot] = *(const f16x8*)(ap + tn * 1024 + 512);
#pragma unroll
        for (int r = 0; r < R; ++r) b[slot][r] = *(const f16x8*)(bp[r] + off);
    };
#pragma unroll
    for (int t = 0; t < WS_PF && t < T; ++t) {
        fetch(t, t % NS);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int cb = t % NS;
        if (t + WS_PF < T) fetch(t + WS_PF, (t + WS_PF) % NS);
        if (FIRST && t == 0) {
            // first tap of a tile: C = the conv bias in the D-fragment layout (the accumulators start at bias: no zeroing,
            // no bias add in the epilogue)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb], b[cb][r], c0, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb], b[cb][r], acc[r], 0, 0, 0);
        }
        if constexpr (X3) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], b[cb][r], acc[r], 0, 0, 0);
        }
        if (t + WS_PF < T) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, X3 ? 2 * R : R, 0);
    }
}

// Row-reuse form of the chunk loop (YR kernels: 3 taps along y, stride 1, the wave's R M-tiles are consecutive rows):
// the B fragment of tap dy at output row r is the input row r + dy, so one (dx, dz) group reads R + 2 input rows and the
// 3 weight fragments once and feeds 3 R MFMAs -- (R + 5) KiB of LDS reads per 3 R MFMAs instead of 3 (R + 1) KiB.
// The per-tap form asks the LDS for 160 B/clk per CU at the full MFMA rate with R = 4 (4 waves x 5 KiB per 4 MFMAs of
// 32 clk), more than the 128 B/clk it has; this form needs 96 B/clk.
template <int R, int K0, int K2, bool FIRST, bool X3>
__device__ __forceinline__ void consume_chunk_y(const unsigned char* b0p, const unsigned char* ap, int xs, int h2,
                                                f32x16 (&acc)[R], const f32x16& c0) {
    // MFMAs run in input-row order (row j feeds the pairs r + dy = j), so row j's registers are dead after its last
    // MFMA and take the same row of the next (dx, dz) group straight away: one set of R + 2 row fragments streams
    // through the groups, only the 3 weight fragments are double-buffered.
    constexpr int G = K0 * K2;
    constexpr int NB = R + 2;
    f16x8 a[2][3];
    f16x8 al[X3 ? 2 : 1][3];
    f16x8 b[NB];
    constexpr int NA = X3 ? 6 : 3;   // weight fragment reads per (dx, dz) group
    constexpr int MM = X3 ? 2 : 1;   // MFMAs per (row, dy) pair
    auto fetch_a = [&](int g, int slot) {
        const int dz = g % K2, dx = g / K2;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            a[slot][dy] = *(const f16x8*)(ap + ((dx * 3 + dy) * K2 + dz) * 1024);
            if constexpr (X3) al[slot][dy] = *(const f16x8*)(ap + ((dx * 3 + dy) * K2 + dz) * 1024 + 512);
        }
    };
    auto fetch_b = [&](int g, int jj) {
        const int dz = g % K2, dx = g / K2;
        b[jj] = *(const f16x8*)(b0p + (dx * xs + jj * h2 + dz) * 16);
    };
    fetch_a(0, 0);
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) fetch_b(0, jj);
    __builtin_amdgcn_sched_group_barrier(0x100, NB + NA, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int cb = g & 1;
        if (g + 1 < G) {
            fetch_a(g + 1, cb ^ 1);
            __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
        }
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
            int cnt = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = jj - dy;
                if (r < 0 || r >= R) continue;
                ++cnt;
                if (FIRST && g == 0 && dy == 0) {
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb][0], b[jj], c0, 0, 0, 0);  // C = bias (see consume_chunk)
                } else {
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb][dy], b[jj], acc[r], 0, 0, 0);
                }
            }
            if constexpr (X3) {   // the lo weight parts of the same (row, dy) pairs
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = jj - dy;
                    if (r < 0 || r >= R) continue;
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb][dy], b[jj], acc[r], 0, 0, 0);
                }
            }
            // row jj is used by min(jj, R - 1) - max(jj - 2, 0) + 1 MFMAs (compile-time per unrolled iteration)
            if (jj == 0 || jj == NB - 1)
                __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
            else if (jj == 1 || jj == NB - 2)
                __builtin_amdgcn_sched_group_barrier(0x008, (R >= 2 ? 2 : 1) * MM, 0);
            else
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * MM, 0);
            (void)cnt;
            if (g + 1 < G) {
                fetch_b(g + 1, jj);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
}

// Split-precision form of the row-reuse loop with TAP PAIRS (round 6): 3 MFMAs per two taps instead of 4.
// A product (Wh + Wl)(Xh + Xl) needs Wh Xh + Wh Xl + Wl Xh (Wl Xl is 2^-22 of the product: below the fp32 rounding of the sum) = three
// K-halves of 8 channels per tap.  The two (dx, dz) groups g0, g1 of a pair share their three MFMAs per (row, dy):
//   [Wh(g0) | Wh(g1)] x [Xh(g0) ; Xh(g1)]  +  [Wh(g0) | Wh(g1)] x [Xl(g0) ; Xl(g1)]  +  [Wl(g0) | Wl(g1)] x [Xh(g0) ; Xh(g1)]
// Every operand is ONE ds_read_b128 whose address depends on the lane's k-half: the kh = 1 lanes read the same plane at group g1's tap
// offset (B: the (dx, dz) shift of the halo voxel; A: the tap's KiB of the weight buffer) -- no register shuffles, no second resident
// chunk.  Per two groups: 12 row fragments + 6 weight fragments (18 KiB of LDS reads, 24 before) feed 9 R MFMAs (12 R before).
// The odd last group pairs its dy = 0 / 1 taps the same way (the kh = 1 lanes read the NEXT input row: R fragment pairs
// [Xh_r ; Xh_r+1], [Xl_r ; Xl_r+1]) and runs dy = 2 in the old form ([Wh | Wh], [Wl | Wl] x [Xh ; Xl], which also carries its Wl Xl):
// 5 R MFMAs instead of 6 R.  Per chunk of 27 taps: 41 R MFMAs (54 R in the round-4 form; 40.5 R is the floor of a 3-term product).
template <int R, int K0, int K2>
__device__ __forceinline__ void consume_chunk_y_x3(const unsigned char* b0p, const unsigned char* ap, int xs, int h2, int plane, int kh,
                                                   f32x16 (&acc)[R]) {
    constexpr int G = K0 * K2;
    constexpr int NP = G / 2;          // pairs of groups
    constexpr int NB = R + 2;
    static_assert((G & 1) == 1 && NP >= 1 && R >= 2 && R <= 4, "consume_chunk_y_x3: 3 or 9 groups, 2 .. 4 rows per wave");
    f16x8 ah[2][3], al[2][3];
    f16x8 fh[NB], fl[NB];
    const unsigned char* bhi = b0p - kh * plane;   // the hi plane for both k-halves (b0p points at this lane's own plane)
    auto gtap = [](int g, int dy) { return (((g / K2) * 3 + dy) * K2 + (g % K2)) * 1024; };
    constexpr int GL = G - 1;                      // the single last group
    auto fetch_a = [&](int u, int slot) {
        if (u < NP) {
            const int g0 = 2 * u, g1 = g0 + 1;
            const unsigned char* a0 = ap + kh * (gtap(g1, 0) - gtap(g0, 0));   // (the difference does not depend on dy)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                ah[slot][dy] = *(const f16x8*)(a0 + gtap(g0, dy));
                al[slot][dy] = *(const f16x8*)(a0 + gtap(g0, dy) + 512);
            }
        } else {   // slot entries 0: the (dy 0 | dy 1) pair, 2: dy 2 for both k-halves
            const unsigned char* a0 = ap + kh * (gtap(GL, 1) - gtap(GL, 0));
            ah[slot][0] = *(const f16x8*)(a0 + gtap(GL, 0));
            al[slot][0] = *(const f16x8*)(a0 + gtap(GL, 0) + 512);
            ah[slot][2] = *(const f16x8*)(ap + gtap(GL, 2));
            al[slot][2] = *(const f16x8*)(ap + gtap(GL, 2) + 512);
        }
    };
    // row fragments of pair u: fh[jj] / fl[jj] = hi / lo plane of input row jj at (g0 | g1)
    auto fetch_b = [&](int u, int jj) {
        const int g0 = 2 * u, g1 = g0 + 1;
        const int o0 = ((g0 / K2) * xs + (g0 % K2)) * 16, o1 = ((g1 / K2) * xs + (g1 % K2)) * 16;
        const unsigned char* q = bhi + (kh ? o1 : o0) + jj * h2 * 16;
        fh[jj] = *(const f16x8*)q;
        fl[jj] = *(const f16x8*)(q + plane);
    };
    // row fragments of the last group, two per call (jj = the pair row whose registers have just been freed):
    //   jj < R: fh[jj] / fl[jj] = [X_jj ; X_jj+1] of the hi / lo plane (the dy 0 | dy 1 pair of output row jj)
    //   jj >= R: input rows 2 + 2 (jj - R), + 1 in the [Xh ; Xl] form -> fh[jj], fl[jj] (dy 2 of output rows 2 (jj - R), + 1)
    auto single_rows = [](int jj) { return jj < R ? 2 : (2 * (jj - R) + 1 < R ? 2 : (2 * (jj - R) < R ? 1 : 0)); };
    auto fetch_s = [&](int jj) {
        const int o = ((GL / K2) * xs + (GL % K2)) * 16;
        if (jj < R) {
            const unsigned char* q = bhi + o + (jj + kh) * h2 * 16;
            fh[jj] = *(const f16x8*)q;
            fl[jj] = *(const f16x8*)(q + plane);
        } else {
            const int i = 2 * (jj - R);
            if (i < R) fh[jj] = *(const f16x8*)(b0p + o + (i + 2) * h2 * 16);
            if (i + 1 < R) fl[jj] = *(const f16x8*)(b0p + o + (i + 3) * h2 * 16);
        }
    };
    fetch_a(0, 0);
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) fetch_b(0, jj);
    __builtin_amdgcn_sched_group_barrier(0x100, 6 + 2 * NB, 0);
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int cb = u & 1;
        fetch_a(u + 1, cb ^ 1);
        if (u + 1 < NP)
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        else
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = jj - dy;
                if (r < 0 || r >= R) continue;
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][dy], fh[jj], acc[r], 0, 0, 0);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = jj - dy;
                if (r < 0 || r >= R) continue;
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][dy], fl[jj], acc[r], 0, 0, 0);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = jj - dy;
                if (r < 0 || r >= R) continue;
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb][dy], fh[jj], acc[r], 0, 0, 0);
            }
            if (jj == 0 || jj == NB - 1)
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            else if (jj == 1 || jj == NB - 2)
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            else
                __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
            if (u + 1 < NP) {
                fetch_b(u + 1, jj);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            } else {
                fetch_s(jj);
                if (single_rows(jj) == 2)
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (single_rows(jj) == 1)
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
    // the last group
    {
        constexpr int cb = NP & 1;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][0], fh[r], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][0], fl[r], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb][0], fh[r], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f16x8& s2 = (r & 1) ? fl[R + r / 2] : fh[R + r / 2];
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][2], s2, acc[r], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f16x8& s2 = (r & 1) ? fl[R + r / 2] : fh[R + r / 2];
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb][2], s2, acc[r], 0, 0, 0);
        }
    }
}

// X3 = split-precision mode (boa_net precision 2, the label-contract mode): activations are fp32 in OCTET planes
// [N][C/8][voxel][8 floats] -- byte for byte the geometry of the fp16 chunk planes, so p.C0 / p.C1 count 2-byte units (2 x the real
// channels), p.src* / p.ss16_* point at fp32 data and everything that only moves bytes (tile walk, halo addressing, weight DMA) is
// shared; the producers normalise in fp32 and split into hi / lo fp16 LDS planes, the consumers issue two MFMAs per tap
// (consume_chunk), the epilogue un-scales the accumulators and stores fp32.  Measured against an fp32 FMA chain the product of
// split operands is the more accurate of the two (tools/x3_probe.hip: rms error 3.2e-7 vs 5.3e-7 of the output rms at K = 864).
// FX (experiment, -DWS_FIXED_SHAPE): the halo strides of the dominant tile shape (w = 1 x 1 x 32, b = 2 x 8 x 1: xs = 340, h2 = 34) as
// compile-time constants in the consumers' fragment addresses (immediate offsets instead of 18 v_add + ~20 SGPRs per chunk)
#define WS_FX_XS 340
#define WS_FX_H2 34
template <int R, int K0, int K1, int K2, bool YR, bool X3, bool FX = false>
__global__ __launch_bounds__(WS_THREADS) void k_conv_ws(ConvArgs p, int total_tiles, int resident_w, int dbg_arg, const int* __restrict__ desc,
                                                        int desc_row) {
    // (round 5, measured and not kept: hipcc re-loads kernel arguments from the kernarg segment in the per-tile paths instead of keeping
    //  them -- chains of s_load_dwordx8/x16 + s_waitcnt in the tile walk and the epilogue addressing; passing every argument through an
    //  empty asm removed all of those loads from the loops (16 instead of 38 s_load, 480 instead of 314 v_readlane) and made every
    //  layer 2 ... 16 % SLOWER, 7 % over a forward: profiles/r05_conv_ws_experiments.txt)
    // (the BOA_WS_DBG ablation switches exist in the traced build only: in the production build they are compile-time zero, which
    //  removes their tests and their wave-uniform masks from the role loops)
#ifdef WS_WITH_TRACE
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    constexpr int taps = K0 * K1 * K2;
    const int HV = p.h0 * p.xs;   // LDS slots of one k-half plane (x-planes xs voxels apart, xs >= h1 * h2)
    const int plane = ws_plane_bytes(HV);
    const int ncc = (p.C0 + p.C1) / 16;
    // LDS map.  resident: [all weights: ncc * taps KiB][halo buf 0][halo buf 1]
    //           streamed: [halo 0 | w 0][halo 1 | w 1]
    const int wres_bytes = resident_w ? ncc * taps * 1024 : 0;
    const int buf_bytes = resident_w ? 2 * plane : 2 * plane + taps * 1024;
    unsigned char* bufs = smem + wres_bytes;

    // this workgroup's tiles: the runs of the virtual workgroups b, b + G, ...
    const int* __restrict__ drow = desc + (size_t)blockIdx.x * desc_row * 8;
    const int my_tiles = drow[0];
    const int my_chunks = my_tiles * ncc;

    // X3: the (scaled) bias in the D-fragment order lives in LDS behind the buffers and is read straight INTO the accumulators at
    // a tile's first chunk -- a persistent 16-register bias fragment made the 256-register kernel spill (11 / 5 VGPRs, round 4)
    float* const s_bias = (float*)(bufs + 2 * buf_bytes);
    if constexpr (X3) {
        for (int i = tid; i < p.Cout; i += WS_THREADS) s_bias[i] = p.bias[i] * p.wscale;   // (ordered by the first chunk barrier)
    }
    if (resident_w) {
        // (only used when Cout == 32: every tile of the launch uses the same weights)
        const int nw = ncc * taps * 64;
        for (int i = tid; i < nw; i += WS_THREADS)
            *(uint4*)(smem + i * 16) = *(const uint4*)(p.wpk + ((size_t)(i >> 5) * p.Cout + (i & 31)) * 8);
        // visibility to the consumers is ordered by the first chunk barrier below
    }
    int tr_n = 0;
    const int tr_blk = WS_TRACING ? (int)p.trace[WS_TRACE_SLOTS - 5] : -1;   // (debug) the traced block

    if (producer) {
        // ---- producer waves: chunk g + 1 is committed to LDS while the consumers work on chunk g; its global loads
        // were issued one barrier earlier (prod_issue), those of chunk g + 2 are issued right after the commit.
        const int q = tid - 256;
        // R = 1 kernels (stride-2 and thin deep layers) are producer-bound: the producers win issue arbitration there
        // (measured +8 % on the 32 -> 64 stride-2 layer); the stride-1 kernels give the consumers the higher priority
        if (R == 1 && !(dbg & 256)) __builtin_amdgcn_s_setprio(3);
        if (R > 1 && (dbg & 2048)) __builtin_amdgcn_s_setprio(3);   // (experiment: producers first on the stride-1 kernels too)
        const ProdConst pc = prod_const(p, q, HV);
        TileDesc pd;   // the tile whose chunks are being issued
        pd.flags = pd.vo = pd.ibase = 0;
        TileCoord& ptc = pd.tc;
        int pk = 0;
        ptc.n = ptc.cy = ptc.ox0 = ptc.oy0 = ptc.oz0 = ptc.sp = 0;
        ProdItems items;
#pragma unroll
        for (int j = 0; j < WS_MAXV; ++j) items.gi[j] = 0;
        items.ok = 0;
        // One register set: the loads of chunk g + 2 are issued behind the commit of chunk g + 1 and have the rest of the interval
        // and the barrier to land.  WS_PF2 (round 5, measured and not adopted) issues them at the HEAD of the interval into a second
        // set, a whole interval ahead (interval loop unrolled by two so that the sets alternate without copies): the 128^3 layers,
        // whose halos are HBM-cold, did not move (+0.2 / -1.5 %), the streamed-weight layers lost 4 .. 9 % (their end-of-commit
        // vmcnt(0) for the weight DMA then also waits for the twelve fresh halo loads): +3.1 % per forward.
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
        // halo reuse across the cout chunks of one spatial tile (cy-fast order, 2 chunks = both stay resident)
        bool reuse = false;
        const bool want_w = !(resident_w || (dbg & 16));
        int pcc = 0;  // chunk within the tile of the next chunk to issue
        const bool live = !(dbg & 2);
        // issue the next chunk into `rg`; behind a tile's LAST chunk the next tile is set up straight away (descriptor load + halo
        // addresses: the interval of a tile's last chunk is the producers' short one -- setting up at the head of the long one, where
        // the consumers' epilogue and the commit share the VALU, measured 2.3 % slower per forward)
        auto issue_next = [&](ChunkRegs& rg, bool tile_after) {
            prod_issue(p, ptc, items, pc.in_halo, pcc, reuse, q, dbg, rg);
            rg.cc = pcc;   // (the weights of the chunk go by DMA when it is committed)
            rg.cy = ptc.cy;
            if (++pcc == ncc) {
                pcc = 0;
                if (tile_after) {
                    pd = load_desc(drow, ++pk);
                    reuse = (pd.flags & WS_DF_REUSE) != 0;  // same spatial tile as the previous tile of this run
                    if (!reuse) prod_setup_desc(p, pd, pc, items);
                    WS_STAMP(8);
                }
            }
        };
        auto commit = [&](ChunkRegs& rg, unsigned char* dst) {
            if (WS_TRACING) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WS_STAMP(7);
            }
            // The halo registers of the chunk to commit are made "used" HERE: hipcc does not see the LDS-DMA loads of the
            // inline asm below, so the s_waitcnt it places in front of the first use of rg.d is vmcnt(0) -- placed after the
            // DMA issue it also waited for the seven DMA round trips (L2 latency on the producers' critical path in every
            // chunk of every streamed-weight layer; found in the ISA in round 4).
#pragma unroll
            for (int j = 0; j < WS_MAXV; ++j) touch128(rg.d[j]);   // as ONE 128-bit tuple (per component hipcc splits the load destinations)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(rg.ssw[j]));
            // (unconditionally: at the join behind a conditional use hipcc would wait again)
            if (want_w) dma_weights(p, __builtin_amdgcn_readfirstlane(rg.cc), __builtin_amdgcn_readfirstlane(rg.cy), dst + 2 * plane, q, taps);
            if constexpr (X3)
                prod_commit_x3(p, rg, dst, q, HV, plane, dbg);
            else
                prod_commit(p, rg, dst, q, HV, plane, dbg);
            // the DMA was issued before the commit's ~2 000 cycles of work.  (Waiting at the end of the interval instead, with
            // vmcnt(number of halo loads issued since), measured 6 % slower.)
            if (want_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WS_STAMP(2);
        };
        if (live && my_chunks > 0) {
            pd = load_desc(drow, 0);
            prod_setup_desc(p, pd, pc, items);
            issue_next(rgA, 1 < my_chunks);
        }
#if WS_PF2
        auto interval = [&](int g, ChunkRegs& cur, ChunkRegs& nxt) {
            if (live && g + 1 < my_chunks) {
                WS_STAMP(1);
                if (g + 2 < my_chunks) issue_next(nxt, g + 3 < my_chunks);
                WS_STAMP(3);
                commit(cur, bufs + ((g + 1) & 1) * buf_bytes);
            }
            __syncthreads();
        };
        for (int g = -1; g < my_chunks; g += 2) {
            interval(g, rgA, rgB);
            if (g + 1 < my_chunks) interval(g + 1, rgB, rgA);
        }
#else
        for (int g = -1; g < my_chunks; ++g) {
            if (live && g + 1 < my_chunks) {
                WS_STAMP(1);
                commit(rgA, bufs + ((g + 1) & 1) * buf_bytes);
                if (g + 2 < my_chunks) issue_next(rgA, g + 3 < my_chunks);
                WS_STAMP(3);
            }
            __syncthreads();
        }
#endif
        return;
    }

    // ---- consumer waves --------------------------------------------------------------------------------
    if (R > 1 && !(dbg & (1024 | 2048))) __builtin_amdgcn_s_setprio(3);  // MFMA / fragment-read issue before the SIMD's producer wave (+1-6 %)
    // per-lane constants (tile independent)
    const int cw = wave & 3;
    int hoff[R];
    {
        const int lz = l31 & (p.w2 - 1);
        const int ly = (l31 >> p.lw2) & (p.w1 - 1);
        const int lx = l31 >> (p.lw2 + p.lw1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = cw * R + r;
            const int mz = m & (p.b2 - 1);
            const int my = (m >> p.lb2) & (p.b1 - 1);
            const int mx = m >> (p.lb2 + p.lb1);
            const int tx = mx * p.w0 + lx, ty = my * p.w1 + ly, tz = mz * p.w2 + lz;
            hoff[r] = ((tx * p.s0) * p.xs + (ty * p.s1) * p.h2 + tz * p.s2) * 16 + kh * plane;
        }
    }
    // epilogue constants: output voxel index relative to the tile origin = srel0 (this lane's voxel l31 within an
    // M-tile) + the M-tile's own offset (wave-uniform, recomputed from cw, r).
    auto vox_rel = [&](int v, int& x, int& y, int& z) {
        x = v >> (p.lw2 + p.lw1);
        y = (v >> p.lw2) & (p.w1 - 1);
        z = v & (p.w2 - 1);
    };
    int srel0;
    {
        int x, y, z;
        vox_rel(l31, x, y, z);
        srel0 = (x * p.Ho + y) * p.Wo + z;
    }
    const size_t out_vox = (size_t)p.Do * p.Ho * p.Wo;
    const int nslots = p.nslots;
    int slot = 0;  // 4 j + wave of the virtual workgroup whose run is being accumulated
    // InstanceNorm partial sums of this wave in the D-fragment layout: entry gq * 4 + e <-> cout 8 gq + 4 kh + e, summed
    // over the voxels (lane l31 of every M-tile) this lane produced since the last flush, from the fp32 accumulators;
    // one flush per (n, cout chunk) the wave works on.
    // (The statistics are those of acc + bias, the value that is stored: the bias add is kept although the InstanceNorm that
    // follows every conv of the stack cancels it -- dropping it was measured slower)
    // Two sets: with the cout-chunk-fastest tile order (p.cy_fast, 2 cout chunks, R == 1 kernels only) consecutive tiles
    // alternate between cout chunk 0 (set A) and 1 (set B); otherwise only set A is used.
    constexpr bool TWO_SETS = (R == 1);
    struct StatSet {
        float s[16], q[16];
    };
    StatSet stA, stB;
#pragma unroll
    for (int i = 0; i < 16; ++i) stA.s[i] = stA.q[i] = stB.s[i] = stB.q[i] = 0.f;
    int st_n = -1, st_cy = 0;
    auto flush_set = [&](StatSet& st, int cy) {
        // recursive halving over the 32 lanes that share kh: after step m a lane keeps half of its entries, summed with
        // its partner's copy (8 + 4 + 2 + 1 shuffles per quantity instead of 5 x 16); entry index = bits 0..3 of the lane
        // in reversed significance, the last step folds lanes 16-31 onto 0-15.  The order of the additions is fixed.
#define WS_HALVE(M, HALF)                                                                                \
    {                                                                                                    \
        const bool up = (l31 & (M)) != 0;                                                                \
        _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                                             \
            const float ks = up ? st.s[i + (HALF)] : st.s[i], gs = up ? st.s[i] : st.s[i + (HALF)];      \
            const float kq = up ? st.q[i + (HALF)] : st.q[i], gq = up ? st.q[i] : st.q[i + (HALF)];      \
            st.s[i] = ks + __shfl_xor(gs, (M));                                                          \
            st.q[i] = kq + __shfl_xor(gq, (M));                                                          \
        }                                                                                                \
    }
        WS_HALVE(1, 8)
        WS_HALVE(2, 4)
        WS_HALVE(4, 2)
        WS_HALVE(8, 1)
#undef WS_HALVE
        st.s[0] += __shfl_xor(st.s[0], 16);
        st.q[0] += __shfl_xor(st.q[0], 16);
        if (l31 < 16) {
            const int i = ((l31 & 1) << 3) | ((l31 & 2) << 1) | ((l31 & 4) >> 1) | ((l31 & 8) >> 3);  // entry this lane ended up with
            const int row = cy * 32 + 8 * (i >> 2) + 4 * kh + (i & 3);
            float* pp = p.partials + (((size_t)st_n * p.Cout + row) * 2) * nslots + slot;
            pp[0] = st.s[0];
            pp[nslots] = st.q[0];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) st.s[i] = st.q[i] = 0.f;
    };
    auto flush_stats = [&]() {
        if (st_n < 0) return;
        if (TWO_SETS && p.cy_fast) {
            flush_set(stA, 0);
            flush_set(stB, 1);
        } else {
            flush_set(stA, st_cy);
        }
    };

    TileDesc cd;
    cd.flags = cd.vo = cd.ibase = 0;
    TileCoord& tc = cd.tc;
    int done_fl = 0, done_vo = 0;
    tc.n = tc.cy = tc.ox0 = tc.oy0 = tc.oz0 = tc.sp = 0;
    if (WS_TRACING && blockIdx.x == 0 && tid == 0) {
        p.trace[WS_TRACE_SLOTS - 4] = __builtin_readcyclecounter();
        p.trace[WS_TRACE_SLOTS - 3] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();  // chunk 0 staged (pairs with the producers' g = -1 barrier)
    // The epilogue of a tile (bias, statistics, transpose, stores: registers and global memory only, no LDS) is DEFERRED past
    // the chunk barrier into the next tile's first interval (WS_DEFER_EPILOGUE): that interval is the producers' long one (they
    // issue the HBM-cold loads of the tile after next), the interval of a tile's last chunk their short one, so the roles'
    // long and short intervals now coincide instead of alternating in opposite phase.
    f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;   // (the first iteration's no-op epilogue reads them)
    // `valid` = there is a previous tile (false in the workgroup's first iteration only): the epilogue runs UNCONDITIONALLY -- as a
    // conditional block its results (partial sums in fresh registers) met the untouched ones at a join and hipcc copied 32 .. 48 register
    // pairs per tile -- and is a numerical no-op then: the accumulators start at zero, the tile counts as sticking out with every voxel
    // masked (no stores, + 0 to the partial sums), the flush logic is skipped.
    auto epilogue = [&](const TileCoord& tc, int dfl, int dvo, bool valid) {
        // ---- epilogue: + bias, InstanceNorm partial sums (fp32), register transpose (v_permlane32_swap), fp16
        // convert, two 16-byte stores per lane (32 contiguous bytes of the voxel's record)
        const bool two = TWO_SETS && p.cy_fast;
        if (valid && (tc.n != st_n || (!two && tc.cy != st_cy))) {
            flush_stats();
            st_n = tc.n;
            st_cy = tc.cy;
        }
        const int cout0 = tc.cy * 32;
        const bool full = valid && (dfl & WS_DF_FULL) != 0;
        const size_t ovox = (size_t)(unsigned)dvo;
        // wave-uniform base + 32-bit lane offset (scalar-base global_store / global_load forms, see prod_issue)
        // chunk-planar output [N][Cout/16][voxel][16]: this lane writes the 16 couts of plane (cout0 / 16 + kh) of its voxel
        const size_t obase = ((size_t)tc.n * p.Cout + cout0) * out_vox + ovox * 16;
        const unsigned olane = ((unsigned)kh * (unsigned)out_vox + (unsigned)srel0) * 32u;
        const unsigned char* const tile_dst = (const unsigned char*)(p.out + obase);   // (the 64-bit part once per tile; an M-tile adds 32 bits)
#ifdef WS_TRACE_EPILOGUE
        WS_STAMP(9);
#endif
#pragma unroll
        for (int r = 0; r < R; ++r) {
#ifdef WS_TRACE_EPILOGUE
            if (r) WS_STAMP(10);
#endif
            const int m = cw * R + r;  // wave-uniform M-tile origin within the block tile
            const int mx = (m >> (p.lb2 + p.lb1)) * p.w0, my = ((m >> p.lb2) & (p.b1 - 1)) * p.w1, mz = (m & (p.b2 - 1)) * p.w2;
            const int mrel = (mx * p.Ho + my) * p.Wo + mz;
            bool ok = true;
            if (!full) {  // wave-uniform: only tiles that stick out of the tensor mask statistics and stores
                int x, y, z;
                vox_rel(l31, x, y, z);
                ok = valid && tc.ox0 + mx + x < p.Do && tc.oy0 + my + y < p.Ho && tc.oz0 + mz + z < p.Wo;
            }
            const float dm = ok ? 1.f : 0.f;
            float v[16];  // conv + bias: the accumulators were started at the bias
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = X3 ? acc[r][i] * p.winv : acc[r][i];
            // Statistics of this M-tile, BEFORE its registers are transposed in place.  Two things hipcc did to the plain loop
            // (round-5 ISA): it turned the wave-uniform `full ? v : v * dm` into a multiply + two selects per pair (5 VALU per pair
            // instead of 2), and it sank all four M-tiles' statistics behind the last store -- which kept the accumulators alive
            // across the destructive v_permlane32_swap and cost 16 register copies per M-tile.  The two arms are real branches
            // (markers), and the empty asm at the end pins the partial sums here.
            // (pairs: the partial sums live in 64-bit register pairs for v_pk_add_f32 / v_pk_fma_f32; pinning the 32 floats one by one
            //  made hipcc drop the packed forms)
            auto stats = [&](StatSet& st) {
                f2_t ps[8], pq[8], pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ps[j] = f2_t{st.s[2 * j], st.s[2 * j + 1]};
                    pq[j] = f2_t{st.q[2 * j], st.q[2 * j + 1]};
                    pv[j] = f2_t{v[2 * j], v[2 * j + 1]};
                }
                if (full) {
                    asm volatile("; stats: tile inside the tensor");
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ps[j] += pv[j];
                        pq[j] = __builtin_elementwise_fma(pv[j], pv[j], pq[j]);
                    }
                    // (pinned inside the arm: behind the join hipcc shares the additions of both arms, unpacked)
                    asm volatile("" : "+v"(ps[0]), "+v"(ps[1]), "+v"(ps[2]), "+v"(ps[3]), "+v"(ps[4]), "+v"(ps[5]), "+v"(ps[6]), "+v"(ps[7]));
                    asm volatile("" : "+v"(pq[0]), "+v"(pq[1]), "+v"(pq[2]), "+v"(pq[3]), "+v"(pq[4]), "+v"(pq[5]), "+v"(pq[6]), "+v"(pq[7]));
                } else {
                    asm volatile("; stats: tile sticks out");
                    const f2_t dm2 = f2_t{dm, dm};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f2_t vm = pv[j] * dm2;
                        ps[j] += vm;
                        pq[j] = __builtin_elementwise_fma(vm, vm, pq[j]);
                    }
                    asm volatile("" : "+v"(ps[0]), "+v"(ps[1]), "+v"(ps[2]), "+v"(ps[3]), "+v"(ps[4]), "+v"(ps[5]), "+v"(ps[6]), "+v"(ps[7]));
                    asm volatile("" : "+v"(pq[0]), "+v"(pq[1]), "+v"(pq[2]), "+v"(pq[3]), "+v"(pq[4]), "+v"(pq[5]), "+v"(pq[6]), "+v"(pq[7]));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    st.s[2 * j] = ps[j][0]; st.s[2 * j + 1] = ps[j][1];
                    st.q[2 * j] = pq[j][0]; st.q[2 * j + 1] = pq[j][1];
                }
            };
            if (TWO_SETS && two && tc.cy != 0)
                stats(stB);
            else
                stats(stA);
            if constexpr (X3) {
                // fp32 octet planes [N][Cout/8][voxel][8]: this lane's entries 4 gq .. 4 gq + 3 are couts 8 gq + 4 kh .. + 3 of its
                // voxel = 16 contiguous bytes of plane cout0 / 8 + gq; the two k-halves of a voxel fill its 32-byte record
                if (ok && !(dbg & 4)) {
                    // (wave-uniform by construction; the explicit readfirstlane keeps hipcc's divergence analysis from rejecting the
                    //  SGPR pin inside this branch: "illegal VGPR to SGPR copy")
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
            // (the statistics read v before the swaps below destroy it: without this fence hipcc hoists the swaps and pays 16
            //  register copies per M-tile to keep v alive)
            __builtin_amdgcn_sched_barrier(0);
            // D fragment -> voxel records without LDS: v_permlane32_swap exchanges the upper half of one register
            // with the lower half of another.  Lane (kh, voxel) holds couts 8 gq + 4 kh + e; swapping group gq with
            // group gq + 2 leaves the kh = 0 lane with couts [0, 16) and the kh = 1 lane with couts [16, 32) of its
            // voxel: 32 contiguous bytes each.
            unsigned w[8];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {  // pr 0: groups (0, 2) -> couts 0-7 | 16-23; pr 1: groups (1, 3) -> 8-15 | 24-31
                float lo4[4], hi4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[pr * 4 + e]), __float_as_uint(v[(pr + 2) * 4 + e]), false, false);
                    lo4[e] = __uint_as_float(sw[0]);  // kh 0: cout 8 pr + e      | kh 1: cout 16 + 8 pr + e
                    hi4[e] = __uint_as_float(sw[1]);  // kh 0: cout 8 pr + 4 + e  | kh 1: cout 16 + 8 pr + 4 + e
                }
                // one v_cvt_pk_f16_f32 (RTNE) per output word
                w[pr * 4 + 0] = cvt_pk_h2(lo4[0], lo4[1]);
                w[pr * 4 + 1] = cvt_pk_h2(lo4[2], lo4[3]);
                w[pr * 4 + 2] = cvt_pk_h2(hi4[0], hi4[1]);
                w[pr * 4 + 3] = cvt_pk_h2(hi4[2], hi4[3]);
            }
            if (ok && !(dbg & 4)) {
                WS_GLOBAL unsigned char* dst = sgpr_ptr(tile_dst + (unsigned)mrel * 32u);
                unsigned ol = olane;
                asm volatile("" : "+v"(ol));  // keep the 32 -> 64 bit extension in this block (instruction selection is per block)
#ifndef WS_TEMPORAL_STORES
                // non-temporal stores: the layer's output is not read again before the launch ends, so it need not displace halo lines from
                // the XCD's L2 (A/B on one box with tools/build_alt.sh, layers repeated at the power cap: 1 780 -> 1 767, 957 -> 946,
                // 823 -> 817 us; bench step 1 820 -> 1 806 ms; the HBM-bound first conv measured 0 ... -10 % with them and keeps plain stores)
                __builtin_nontemporal_store(u32x4_t{w[0], w[1], w[2], w[3]}, (WS_GLOBAL u32x4_t*)(dst + ol));
#if defined(WS_ABL_STORE_HALF)        // ablation: half the bytes, half the store instructions
#elif defined(WS_ABL_STORE_SAME)      // ablation: both instructions, the same bytes (half the HBM bytes, all the requests)
                __builtin_nontemporal_store(u32x4_t{w[4], w[5], w[6], w[7]}, (WS_GLOBAL u32x4_t*)(dst + ol));
#else
                __builtin_nontemporal_store(u32x4_t{w[4], w[5], w[6], w[7]}, (WS_GLOBAL u32x4_t*)(dst + ol + 16));
#endif
#else
                *(WS_GLOBAL u32x4_t*)(dst + ol) = u32x4_t{w[0], w[1], w[2], w[3]};
                *(WS_GLOBAL u32x4_t*)(dst + ol + 16) = u32x4_t{w[4], w[5], w[6], w[7]};
#endif
            }
        }
#ifdef WS_TRACE_EPILOGUE
        WS_STAMP(11);
#endif
    };
    f32x16 biasv;
#pragma unroll
    for (int i = 0; i < 16; ++i) biasv[i] = 0.f;
    int bias_cy = -1;
    TileCoord done_tc;
    done_tc.n = done_tc.cy = done_tc.ox0 = done_tc.oy0 = done_tc.oz0 = done_tc.sp = 0;
    // (one extra iteration for the last tile's deferred epilogue: a single call site -- two inlined copies of the epilogue
    //  made hipcc's backend fail with "illegal VGPR to SGPR copy")
    for (int k = 0; k < my_tiles + WS_DEFER_EPILOGUE; ++k) {
        bool new_run = true;
        const bool more = k < my_tiles;
#ifdef WS_TRACE_EPILOGUE
        WS_STAMP(13);
#endif
        if (more) {
            cd = load_desc(drow, k);
            new_run = (cd.flags & WS_DF_NEWRUN) != 0;
        }
#ifdef WS_TRACE_EPILOGUE
        WS_STAMP(14);
#endif
#if WS_DEFER_EPILOGUE
        if (!(dbg & 8)) epilogue(done_tc, done_fl, done_vo, k > 0);  // the previous tile's (its statistics belong to the previous run: before the flush)
        // The accumulators are dead from here (the tile's first MFMA / bias load overwrites them), which hipcc cannot see through the
        // chunk loop's `cc == 0` test: an empty asm "defines" them -- on EVERY path, so that no path has to carry the old values
        // (inside the conditional epilogue it made hipcc copy all 64 accumulator registers to a second bank and back once per tile)
        // and the epilogue's v_permlane32_swap transposes them in place (16 v_mov per M-tile otherwise).
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "=v"(acc[r]));
        if (!more) break;
#endif
        if (new_run) {  // the partial sums of a virtual workgroup go to its own slot
            flush_stats();
            st_n = -1;
            slot = (int)((unsigned)cd.flags >> 16) * 4 + cw;
        }
        if (!X3 && tc.cy != bias_cy) {  // this lane's 16 biases of the cout chunk, D-fragment layout (entry 4 gq + e <-> cout 8 gq + 4 kh + e)
            bias_cy = tc.cy;
            const WS_GLOBAL unsigned char* bbase = sgpr_ptr(p.bias + __builtin_amdgcn_readfirstlane(tc.cy) * 32);
            unsigned bl = (unsigned)kh * 16u;
            asm volatile("" : "+v"(bl));
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4_t bv = *(const WS_GLOBAL f32x4_t*)(bbase + bl + 32 * gq);
                biasv[gq * 4 + 0] = bv[0]; biasv[gq * 4 + 1] = bv[1]; biasv[gq * 4 + 2] = bv[2]; biasv[gq * 4 + 3] = bv[3];
            }
            if constexpr (X3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) biasv[i] *= p.wscale;
            }
        }
#ifdef WS_TRACE_EPILOGUE
        WS_STAMP(12);
#endif
        for (int cc = 0; cc < ncc; ++cc) {
            const int g = k * ncc + cc;
            const unsigned char* cur = bufs + (g & 1) * buf_bytes;
            const unsigned char* bp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // opaque per chunk: stops hipcc from hoisting the 9 x R row addresses out of the chunk loop (spills)
                int ho = hoff[r];
                asm volatile("" : "+v"(ho));
                bp[r] = cur + ho;
            }
            const unsigned char* ap = (resident_w ? smem + cc * taps * 1024 : cur + 2 * plane) + (X3 ? l31 : kh * 32 + l31) * 16;
            WS_STAMP(4);
            if (dbg & 4096) {   // (ablation: consumers without fragment reads and MFMAs -- the producers' stand-alone pace)
            } else if constexpr (X3) {
                if (cc == 0) {   // accumulators start at bias * wscale (LDS table, D-fragment order: entry 4 gq + e <-> cout 8 gq + 4 kh + e)
                    const float* sb = s_bias + tc.cy * 32 + 4 * kh;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4_t bv = *(const f32x4_t*)(sb + 8 * gq);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            acc[r][gq * 4 + 0] = bv[0]; acc[r][gq * 4 + 1] = bv[1]; acc[r][gq * 4 + 2] = bv[2]; acc[r][gq * 4 + 3] = bv[3];
                        }
                    }
                }
                if constexpr (YR) {
#ifdef WS_X3_UNPAIRED   // (A/B: the round-4 form, 4 MFMAs per two taps)
                    consume_chunk_y<R, K0, K2, false, X3>(bp[0], ap, p.xs, p.h2, acc, biasv);
#else
                    consume_chunk_y_x3<R, K0, K2>(bp[0], ap, p.xs, p.h2, plane, kh, acc);
#endif
                } else
                    consume_chunk<R, K0, K1, K2, false, X3>(bp, ap, p.xs, p.h2, acc, biasv);
            } else if constexpr (YR && FX) {
                if (cc == 0)
                    consume_chunk_y<R, K0, K2, true, X3>(bp[0], ap, WS_FX_XS, WS_FX_H2, acc, biasv);
                else
                    consume_chunk_y<R, K0, K2, false, X3>(bp[0], ap, WS_FX_XS, WS_FX_H2, acc, biasv);
            } else if constexpr (YR) {
                if (cc == 0)
                    consume_chunk_y<R, K0, K2, true, X3>(bp[0], ap, p.xs, p.h2, acc, biasv);
                else
                    consume_chunk_y<R, K0, K2, false, X3>(bp[0], ap, p.xs, p.h2, acc, biasv);
            } else if (cc == 0)
                consume_chunk<R, K0, K1, K2, true, X3>(bp, ap, p.xs, p.h2, acc, biasv);
            else
                consume_chunk<R, K0, K1, K2, false, X3>(bp, ap, p.xs, p.h2, acc, biasv);
            WS_STAMP(5);
#if !WS_DEFER_EPILOGUE
            if (cc == ncc - 1 && !(dbg & 8)) epilogue(tc, cd.flags, cd.vo, true);
#endif
            WS_STAMP(6);
            __syncthreads();
        }
        // (wave-uniform values: pinned to SGPRs, the epilogue's scalar-base stores need them there)
        done_tc.n = __builtin_amdgcn_readfirstlane(tc.n);
        done_tc.cy = __builtin_amdgcn_readfirstlane(tc.cy);
        done_tc.sp = __builtin_amdgcn_readfirstlane(tc.sp);
        done_tc.ox0 = __builtin_amdgcn_readfirstlane(tc.ox0);
        done_tc.oy0 = __builtin_amdgcn_readfirstlane(tc.oy0);
        done_tc.oz0 = __builtin_amdgcn_readfirstlane(tc.oz0);
        done_fl = __builtin_amdgcn_readfirstlane(cd.flags);
        done_vo = __builtin_amdgcn_readfirstlane(cd.vo);
    }
    if (WS_TRACING && blockIdx.x == 0 && tid == 0) {
        p.trace[WS_TRACE_SLOTS - 2] = __builtin_readcyclecounter();
        p.trace[WS_TRACE_SLOTS - 1] = __builtin_amdgcn_s_memrealtime();
    }
    flush_stats();
}

// ---- host side ---------------------------------------------------------------------------------------
static size_t ws_plane_host(int HV) { return ((size_t)(HV + WS_PROD / 2 - 1) / (WS_PROD / 2)) * (WS_PROD / 2) * 16 + 64; }

// resident weights need every tile of the launch to use the same weights: Cout == 32 (one cout chunk)
bool conv_ws_resident(int HV, int taps, int ncc, int Cout) {
    return Cout == 32 && (size_t)ncc * taps * 1024 + 4 * ws_plane_host(HV) <= 160 * 1024 - 2048;
}

size_t conv_ws_lds_bytes(int HV, int taps, int ncc, int Cout) {
    if (conv_ws_resident(HV, taps, ncc, Cout)) return (size_t)ncc * taps * 1024 + 4 * ws_plane_host(HV);
    return 2 * (2 * ws_plane_host(HV) + (size_t)taps * 1024);
}

// virtual workgroups per sample (a function of the layer geometry only; BOA_WS_VW: experiment hook, changes the statistics
// grouping) and the statistics slots of a layer: one per (virtual workgroup, consumer wave) -- sized per layer, so that
// k_norm_finalize reads and re-zeroes 4 vw entries per (n, cout), not 4 x CUs (the deep layers have 10-64 tiles per sample)
int conv_ws_vw(int tiles_per_sample, int cu_count) {
    static const int vw_cap = getenv("BOA_WS_VW") ? atoi(getenv("BOA_WS_VW")) : 0;
    return std::min(tiles_per_sample, vw_cap > 0 ? std::min(vw_cap, cu_count) : cu_count);
}
int conv_ws_nslots(int tiles_per_sample, int cu_count) { return conv_ws_vw(tiles_per_sample, cu_count) * 4; }

bool conv_ws_supported(const int k[3], int HV) {
    const bool k333 = k[0] == 3 && k[1] == 3 && k[2] == 3;
    const bool k133 = k[0] == 1 && k[1] == 3 && k[2] == 3;
    return (k333 || k133) && 2 * HV <= WS_PROD * WS_MAXV;
}

template <int R, int K0, int K1, int K2, bool YR, bool X3>
static void launch_ws_y(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, int total, int grid, int resident, const int* desc, int desc_row) {
#ifdef WS_FIXED_SHAPE
    if constexpr (YR && R == 4 && !X3) {
        if (a.xs == WS_FX_XS && a.h2 == WS_FX_H2) {
            static bool once_fx = (hipFuncSetAttribute((const void*)k_conv_ws<R, K0, K1, K2, YR, X3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
            (void)once_fx;
            hipLaunchKernelGGL((k_conv_ws<R, K0, K1, K2, YR, X3, true>), dim3(grid), dim3(WS_THREADS), t.lds_bytes + 2048, ctx->stream, a, total, resident,
                               getenv("BOA_WS_DBG") ? atoi(getenv("BOA_WS_DBG")) : 0, desc, desc_row);
            return;
        }
    }
#endif
    static bool once = (hipFuncSetAttribute((const void*)k_conv_ws<R, K0, K1, K2, YR, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL((k_conv_ws<R, K0, K1, K2, YR, X3>), dim3(grid), dim3(WS_THREADS), t.lds_bytes + 2048, ctx->stream, a, total, resident,
                       getenv("BOA_WS_DBG") ? atoi(getenv("BOA_WS_DBG")) : 0, desc, desc_row);
}

template <int R, int K0, int K1, int K2, bool X3>
static void launch_ws_t(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, int total, int grid, int resident, const int* desc, int desc_row) {
    // row reuse: the R M-tiles of a wave are consecutive output rows (m = cw * R + r, my = m & (b1 - 1) when b2 == 1),
    // one voxel high, stride 1 along y
    static const bool off = getenv("BOA_WS_NO_YREUSE") != nullptr;
    const bool yr = R > 1 && K1 == 3 && a.s1 == 1 && a.w1 == 1 && a.b2 == 1 && a.b1 % R == 0 && !off;
    if (R > 1 && yr)
        launch_ws_y<R, K0, K1, K2, (R > 1), X3>(ctx, a, t, total, grid, resident, desc, desc_row);
    else
        launch_ws_y<R, K0, K1, K2, false, X3>(ctx, a, t, total, grid, resident, desc, desc_row);
}

template <int R, bool X3>
static int launch_ws_r(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, int total, int grid, int resident, const int* desc, int desc_row) {
    if (a.k0 == 3 && a.k1 == 3 && a.k2 == 3)
        launch_ws_t<R, 3, 3, 3, X3>(ctx, a, t, total, grid, resident, desc, desc_row);
    else if (a.k0 == 1 && a.k1 == 3 && a.k2 == 3)
        launch_ws_t<R, 1, 3, 3, X3>(ctx, a, t, total, grid, resident, desc, desc_row);
    else {
        boa_set_error("conv_ws: kernel %dx%dx%d not instantiated", a.k0, a.k1, a.k2);
        return BOA_EINVAL;
    }
    return BOA_OK;
}

// Run table of a layer: the tiles of ONE sample (decode_tile's order) cut into vw contiguous runs, 8 XCD ranges first
// (virtual workgroup j <-> XCD j % 8) and then the virtual workgroups of an XCD; entry j = {count, cy, sp, ox0, oy0, oz0}
// of the run's first tile.  A function of the layer geometry only (never of the batch size).
const int* ws_run_table(boa_ctx* ctx, const ConvArgs& a, int tiles_per_sample, int vw) {
    std::vector<int> key = {a.t0, a.t1, a.t2, a.b0, a.b1, a.b2, a.w0, a.w1, a.w2, a.Cout, a.cy_fast, vw, a.ncy};
    for (auto& e : ctx->ws_runs)
        if (e.first == key) return (const int*)e.second;
    std::vector<int> tab((size_t)vw * 8, 0);
    const int nsp = a.t0 * a.t1 * a.t2, ncy = a.ncy;
    for (int j = 0; j < vw; ++j) {
        int first, count;
        if (vw % 8 != 0) {
            const int q = tiles_per_sample / vw, r = tiles_per_sample % vw;
            first = j * q + std::min(j, r);
            count = q + (j < r ? 1 : 0);
        } else {
            const int xcd = j & 7, slot = j >> 3, per = vw >> 3;
            const int q = tiles_per_sample / 8, rem = tiles_per_sample % 8;
            const int lo = xcd * q + std::min(xcd, rem), len = q + (xcd < rem ? 1 : 0);
            const int cq = len / per, cr = len % per;
            first = lo + slot * cq + std::min(slot, cr);
            count = cq + (slot < cr ? 1 : 0);
        }
        int cy, sp;
        if (a.cy_fast) {
            cy = first % ncy;
            sp = first / ncy;
        } else {
            sp = first % nsp;
            cy = first / nsp;
        }
        int bt = sp;
        const int tx = bt % a.t0;
        bt /= a.t0;
        const int tz = bt % a.t2, ty = bt / a.t2;
        int* e = &tab[(size_t)j * 8];
        e[0] = count; e[1] = cy; e[2] = sp; e[3] = tx * a.b0 * a.w0; e[4] = ty * a.b1 * a.w1; e[5] = tz * a.b2 * a.w2;
    }
    void* dev = nullptr;
    if (hipMalloc(&dev, tab.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(dev);
        return nullptr;
    }
    ctx->ws_runs.emplace_back(std::move(key), dev);
    ctx->ws_runs_host.emplace_back(std::move(tab));
    return (const int*)dev;
}

// Descriptor table of a launch (WS_DESC): a function of the layer geometry, the batch and the grid; built once per key on the
// device by k_ws_build_desc and kept for the context's lifetime (8 samples of the 128^3 layers: 32 768 tiles = 1 MiB).
const int* ws_desc_table(boa_ctx* ctx, const ConvArgs& a, int tiles_per_sample, int grid, int* row_out) {
    std::vector<int> key = {-1,   a.t0, a.t1, a.t2, a.b0, a.b1, a.b2, a.w0, a.w1, a.w2, a.Cout, a.cy_fast, a.vw, a.ncy, a.N,  grid,
                            a.Do, a.Ho, a.Wo, a.Di, a.Hi, a.Wi, a.s0, a.s1, a.s2, a.p0, a.p1,   a.p2,      a.h0, a.h1,  a.h2, a.xs};
    // rows: an upper bound of a workgroup's tiles.  A run has at most tiles_per_sample / vw + 2 tiles (ws_run_table: the 8 XCD ranges
    // differ by one tile, the runs inside a range by one more) and a workgroup executes ceil(N vw / grid) runs.  The tables live as long
    // as the context (one per layer geometry, batch and grid: the tail batches of a volume add a few; <= ~150 MB for every batch size
    // 1 .. 25 of one net geometry).
    const int vper = (a.N * a.vw + grid - 1) / grid;
    const int row = vper * (tiles_per_sample / a.vw + 2) + 1;
    *row_out = row;
    for (auto& e : ctx->ws_runs)
        if (e.first == key) return (const int*)e.second;
    // the row length is an upper bound of every workgroup's tile count: checked here against the host copy of the run table, so the
    // builder kernel can neither truncate a row nor write past it (an error return instead of a device trap)
    {
        const std::vector<int>* runs_host = nullptr;
        for (size_t i = 0; i < ctx->ws_runs.size(); ++i)
            if (ctx->ws_runs[i].second == (const void*)a.runs) runs_host = &ctx->ws_runs_host[i];
        if (!runs_host || runs_host->size() < (size_t)a.vw * 8) return nullptr;
        for (int b = 0; b < grid; ++b) {
            long long my = 0;
            for (long long v = b; v < (long long)a.N * a.vw; v += grid) my += (*runs_host)[(size_t)(v % a.vw) * 8];
            if (my > row - 1) {
                boa_set_error("conv_ws: workgroup %d walks %lld tiles, descriptor rows hold %d", b, my, row - 1);
                return nullptr;
            }
        }
    }
    void* dev = nullptr;
    const size_t bytes = (size_t)grid * row * 8 * sizeof(int);
    if (hipMalloc(&dev, bytes) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(k_ws_build_desc, dim3((grid + 63) / 64), dim3(64), 0, ctx->stream, a, grid, row, (int*)dev);
    ctx->ws_runs.emplace_back(std::move(key), dev);
    ctx->ws_runs_host.emplace_back();
    ctx->ws_desc_bytes += bytes;
    return (const int*)dev;
}

int launch_conv_ws(boa_ctx* ctx, const ConvArgs& a_in, const ConvTile& t, double flops, double bytes, bool x3) {
    const ConvArgs& a0 = a_in;
    // tiles of one sample; its virtual workgroups (batch-invariant statistics, see tile_walk); physical grid
    const int total = t.tiles[0] * t.tiles[1] * t.tiles[2] * (a0.Cout / 32);
    const int vw = conv_ws_vw(total, ctx->cu_count);
    const int grid = (int)std::min<long long>((long long)vw * a0.N, ctx->cu_count);
    const int taps = a0.k0 * a0.k1 * a0.k2;
    const int HV = t.h[0] * (t.xs > 0 ? t.xs : t.h[1] * t.h[2]);
    const int resident = conv_ws_resident(HV, taps, (a0.C0 + a0.C1) / 16, a0.Cout) ? 1 : 0;
    BOA_REQUIRE(!x3 || a0.Cout * 4 <= 2048, "conv_ws (split precision): Cout %d exceeds the 512-entry bias table kept in LDS behind the halo buffers", a0.Cout);
    BOA_REQUIRE((double)a0.Di * a0.Hi * a0.Wi <= 16777216.0 && std::max(a0.C0, a0.C1) * 2 < 16777216,
                "conv_ws: more than 2^24 input voxels per sample (24-bit offset multiply)");
    BOA_REQUIRE((double)a0.Di * a0.Hi * a0.Wi * std::max(a0.C0, a0.C1) * 2.0 < 4294967296.0,
                "conv_ws: one sample of the input exceeds 4 GiB (32-bit voxel offsets)");
#ifdef WS_WITH_TRACE
    static const bool want_trace = getenv("BOA_WS_TRACE") != nullptr;
#else
    static const bool want_trace = false;
    static const bool trace_note = getenv("BOA_WS_TRACE") != nullptr && (fprintf(stderr, "[boa_hip] BOA_WS_TRACE needs a library built with -DWS_WITH_TRACE (tools/build_alt.sh trace -DWS_WITH_TRACE)\n"), true);
    (void)trace_note;
#endif
    ConvArgs a = a_in;
    a.trace = nullptr;
    // statistics slots: one per (workgroup, consumer wave); waves that never touch an (n, cout chunk) leave zeros
    a.nslots = 4 * vw;  // = conv_nblk(t, cu_count, Cout): the stride the caller allocated and k_norm_finalize reads
    a.vw = vw;
    a.vstep_n = grid / vw;
    a.vstep_j = grid % vw;
    a.ncy = a.Cout / 32;
    a.cy_fast = ((a.C0 + a.C1) == 32 && a.Cout == 64 && t.R == 1 && !getenv("BOA_WS_NO_CYFAST")) ? 1 : 0;
    a.runs = ws_run_table(ctx, a, total, vw);
    BOA_REQUIRE(a.runs != nullptr, "conv_ws: could not allocate the run table");
    const int* desc = nullptr;
    int desc_row = 0;
    desc = ws_desc_table(ctx, a, total, grid, &desc_row);
    BOA_REQUIRE(desc != nullptr, "conv_ws: could not allocate the tile descriptor table");
    // a.partials must be all zero on entry: the callers zero it once (allocation / test seam) and k_norm_finalize
    // clears what it has read, so no per-launch memset is needed
    if (want_trace) {
        hipMalloc(&a.trace, WS_TRACE_SLOTS * 8);
        hipMemsetAsync(a.trace, 0, WS_TRACE_SLOTS * 8, ctx->stream);
        const unsigned long long blk = (unsigned long long)std::min(std::max(atoi(getenv("BOA_WS_TRACE")), 0), grid - 1);
        hipMemcpyAsync(a.trace + WS_TRACE_SLOTS - 5, &blk, 8, hipMemcpyHostToDevice, ctx->stream);
        hipStreamSynchronize(ctx->stream);
    }
    KernelTimer tm(ctx, BOA_K_CONV_MFMA, flops, bytes);
    ctx->counters[x3 ? BOA_CNT_CONV_X3 : BOA_CNT_CONV_WS]++;
    int rc;
    switch (t.R + (x3 ? 8 : 0)) {
        case 4: rc = launch_ws_r<4, false>(ctx, a, t, total, grid, resident, desc, desc_row); break;
#ifndef WS_BISECT
        case 2: rc = launch_ws_r<2, false>(ctx, a, t, total, grid, resident, desc, desc_row); break;
        case 1: rc = launch_ws_r<1, false>(ctx, a, t, total, grid, resident, desc, desc_row); break;
        case 12: rc = launch_ws_r<4, true>(ctx, a, t, total, grid, resident, desc, desc_row); break;
        case 10: rc = launch_ws_r<2, true>(ctx, a, t, total, grid, resident, desc, desc_row); break;
        case 9: rc = launch_ws_r<1, true>(ctx, a, t, total, grid, resident, desc, desc_row); break;
#endif
        default:
            boa_set_error("conv_ws: unsupported R=%d", t.R);
            rc = BOA_EINVAL;
    }
    tm.stop();
    if (want_trace && a.trace) {
        // debug: print the per-event deltas (cycles) of block 0: consumer wave 0 then producer wave 4
        static unsigned long long host[WS_TRACE_SLOTS];
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(host, a.trace, sizeof(host), hipMemcpyDeviceToHost);
        hipFree(a.trace);
        fprintf(stderr, "[ws-clock] Cin=%d Cout=%d in=%d: %llu cycles in %llu x10ns -> %.3f GHz, %d tiles per block\n", a0.C0 + a0.C1, a0.Cout, a0.Di,
                host[WS_TRACE_SLOTS - 2] - host[WS_TRACE_SLOTS - 4], host[WS_TRACE_SLOTS - 1] - host[WS_TRACE_SLOTS - 3],
                (double)(host[WS_TRACE_SLOTS - 2] - host[WS_TRACE_SLOTS - 4]) / (10.0 * (double)(host[WS_TRACE_SLOTS - 1] - host[WS_TRACE_SLOTS - 3])),
                (total * a0.N + grid - 1) / grid);
        for (int role = 0; role < 2; ++role) {
            const unsigned long long* h = host + role * (WS_TRACE_SLOTS / 2);
            fprintf(stderr, "[ws-trace] %s Cin=%d Cout=%d in=%d R=%d:", role ? "producer" : "consumer", a0.C0 + a0.C1, a0.Cout, a0.Di, t.R);
            const int lo = 40, hi = 40 + (role ? 50 : 80);
            for (int i = lo; i < hi && h[i]; ++i)
                fprintf(stderr, " %d:%llu", (int)(h[i] >> 56), (h[i] & 0x00ffffffffffffffull) - (h[i - 1] & 0x00ffffffffffffffull));
            fprintf(stderr, "\n");
        }
    }
    if (rc) return rc;
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
