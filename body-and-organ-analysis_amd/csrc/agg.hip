// HBM-bound voxel aggregations of the BOA body-composition / measurement path (integer arithmetic only):
// tissue subclassification + slice-wise counts and HU sums, per-label HU histograms, label/HU masks,
// separable binary erosion, 26-connected component labelling by atomic union-find.
#include <string.h>

#include <algorithm>

#include "common.h"

// ------------------------------------------------------------------------------------------------------
// BCA/tissue/subclassification.py:38-53 with the rules of BCA/tissue/definition.py:22-30 applied in enum order
// (later rules overwrite): MUSCLE(-29..150 in MUSCLE=2), BONE(-1000..3000 in BONE=5), SAT/VAT/IMAT/PAT/EAT
// (-190..-30 in SUBCUTANEOUS=1 / ABDOMINAL=3 / MUSCLE=2 / MEDIASTINUM=9 / PERICARDIUM=7).
__device__ __forceinline__ int tissue_of(int hu, int region) {
    const bool adip = hu >= -190 && hu <= -30;
    int t = 0;
    if (region == 2 && hu >= -29 && hu <= 150) t = 1;
    if (region == 5 && hu >= -1000 && hu <= 3000) t = 2;
    if (adip) {
        if (region == 1) t = 3;
        if (region == 3) t = 4;
        if (region == 2) t = 5;
        if (region == 9) t = 6;
        if (region == 7) t = 7;
    }
    return t;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_tissue_aggregate(const short* __restrict__ ct,
                                                          const short* __restrict__ ct_rules,
                                                          const unsigned char* __restrict__ regions,
                                                          const unsigned char* __restrict__ parts,
                                                          unsigned char* __restrict__ tissues, int slice_vox,
                                                          unsigned int* __restrict__ counts,
                                                          long long* __restrict__ sums) {
    __shared__ unsigned int s_cnt[16];
    __shared__ long long s_sum[16];
    const int z = blockIdx.y;
    if (threadIdx.x < 16) {
        s_cnt[threadIdx.x] = 0;
        s_sum[threadIdx.x] = 0;
    }
    __syncthreads();
    int cnt[2][8];
    int sum[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 8; ++t) cnt[a][t] = sum[a][t] = 0;
    const size_t base = (size_t)z * slice_vox;
    const int nvec = slice_vox / VEC;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        short hu[VEC] __attribute__((aligned(16)));
        short hr[VEC] __attribute__((aligned(16)));
        unsigned char rg[VEC] __attribute__((aligned(8)));
        unsigned char pt[VEC] __attribute__((aligned(8)));
        unsigned char ts[VEC] __attribute__((aligned(8)));
        const size_t o = base + (size_t)i * VEC;
        if (VEC == 8) {
            *(uint4*)hu = *(const uint4*)(ct + o);
            *(uint4*)hr = ct_rules ? *(const uint4*)(ct_rules + o) : *(const uint4*)hu;
            *(uint2*)rg = *(const uint2*)(regions + o);
            if (parts) *(uint2*)pt = *(const uint2*)(parts + o);
        } else {
            hu[0] = ct[o];
            hr[0] = ct_rules ? ct_rules[o] : hu[0];
            rg[0] = regions[o];
            if (parts) pt[0] = parts[o];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int t = tissue_of(hr[j], rg[j]);
            ts[j] = (unsigned char)t;
            const bool torso = parts && pt[j] == 1;
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const bool m = (t == k);
                cnt[0][k] += m ? 1 : 0;
                sum[0][k] += m ? hu[j] : 0;
                cnt[1][k] += (m && torso) ? 1 : 0;
                sum[1][k] += (m && torso) ? hu[j] : 0;
            }
        }
        if (tissues) {
            if (VEC == 8)
                *(uint2*)(tissues + o) = *(const uint2*)ts;
            else
                tissues[o] = ts[0];
        }
    }
    // wave reduce, then one LDS atomic per wave and counter, then one global atomic per block and counter
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            int c = cnt[a][k];
            long long s = sum[a][k];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                c += __shfl_xor(c, m);
                s += __shfl_xor(s, m);
            }
            if ((threadIdx.x & 63) == 0 && c) {
                atomicAdd(&s_cnt[a * 8 + k], (unsigned int)c);
                atomicAdd((unsigned long long*)&s_sum[a * 8 + k], (unsigned long long)s);
            }
        }
    __syncthreads();
    if (threadIdx.x < 16 && s_cnt[threadIdx.x]) {
        atomicAdd(&counts[(size_t)z * 16 + threadIdx.x], s_cnt[threadIdx.x]);
        atomicAdd((unsigned long long*)&sums[(size_t)z * 16 + threadIdx.x], (unsigned long long)s_sum[threadIdx.x]);
    }
}

extern "C" int boa_tissue_aggregate(boa_ctx* c, const int16_t* dev_ct, const int16_t* dev_ct_rules,
                                    const uint8_t* dev_regions, const uint8_t* dev_parts, uint8_t* dev_tissues_out, int Z, int Y, int X,
                                    uint32_t* dev_counts, int64_t* dev_hu_sums) {
    BOA_REQUIRE(c && dev_ct && dev_regions && dev_counts && dev_hu_sums, "boa_tissue_aggregate: NULL argument");
    BOA_REQUIRE(Z > 0 && Y > 0 && X > 0 && (long long)Y * X < (1ll << 30), "boa_tissue_aggregate: bad dims");
    BOA_HIP_TRY(hipMemsetAsync(dev_counts, 0, (size_t)Z * 16 * sizeof(uint32_t), c->stream));
    BOA_HIP_TRY(hipMemsetAsync(dev_hu_sums, 0, (size_t)Z * 16 * sizeof(int64_t), c->stream));
    const int sv = Y * X;
    const bool vec8 = (sv % 8 == 0) && (((uintptr_t)dev_ct) % 16 == 0) && (((uintptr_t)dev_regions) % 8 == 0) &&
                      (((uintptr_t)dev_parts) % 8 == 0) && (((uintptr_t)dev_tissues_out) % 8 == 0);
    const int nvec = vec8 ? sv / 8 : sv;
    // workgroups per slice: 16 (each thread reduces >= 64 voxels before the wave / block reduction and its 14 global atomics; 64
    // workgroups per slice measured 0.43 ms per 512^3 volume, 8 ... 32: 0.35 ms); $BOA_TISSUE_GX: experiment hook
    static const int gx_cap = getenv("BOA_TISSUE_GX") ? atoi(getenv("BOA_TISSUE_GX")) : 16;
    int gx = std::min(ceil_div(nvec, 256), gx_cap);
    const double vox = (double)Z * sv;
    KernelTimer t(c, BOA_K_AGG, 0, vox * (3.0 + (dev_ct_rules ? 2 : 0) + (dev_parts ? 1 : 0) + (dev_tissues_out ? 1 : 0)));
    if (vec8)
        hipLaunchKernelGGL(k_tissue_aggregate<8>, dim3(gx, Z), dim3(256), 0, c->stream, dev_ct, dev_ct_rules, dev_regions,
                           dev_parts, dev_tissues_out, sv, dev_counts, (long long*)dev_hu_sums);
    else
        hipLaunchKernelGGL(k_tissue_aggregate<1>, dim3(gx, Z), dim3(256), 0, c->stream, dev_ct, dev_ct_rules, dev_regions,
                           dev_parts, dev_tissues_out, sv, dev_counts, (long long*)dev_hu_sums);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slice_presence(const unsigned char* __restrict__ labels, int slice_vox,
                                                        unsigned char* __restrict__ present) {
    __shared__ unsigned int flags[256];
    const int z = blockIdx.y;
    flags[threadIdx.x] = 0;
    __syncthreads();
    const unsigned char* p = labels + (size_t)z * slice_vox;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < slice_vox; i += gridDim.x * 256) flags[p[i]] = 1;
    __syncthreads();
    if (flags[threadIdx.x]) present[(size_t)z * 256 + threadIdx.x] = 1;
}

extern "C" int boa_slice_label_presence(boa_ctx* c, const uint8_t* dev_labels, int Z, int Y, int X,
                                        uint8_t* dev_present) {
    BOA_REQUIRE(c && dev_labels && dev_present && Z > 0 && Y > 0 && X > 0, "boa_slice_label_presence: bad argument");
    BOA_HIP_TRY(hipMemsetAsync(dev_present, 0, (size_t)Z * 256, c->stream));
    const int sv = Y * X;
    int gx = std::min(ceil_div(sv, 256 * 8), 32);
    KernelTimer t(c, BOA_K_AGG, 0, (double)Z * sv);
    hipLaunchKernelGGL(k_slice_presence, dim3(gx, Z), dim3(256), 0, c->stream, dev_labels, sv, dev_present);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// tissue projections: the reductions behind the report's coronal / sagittal tissue heat maps
// (BCA/report/plots/heatmaps.py:29-101): per selected tissue `tissue_mask.sum(axis=1)` and `.sum(axis=2)` of the (z,y,x)
// tissue volume, and `((regions > 0) & (regions < 255)).any(axis)` for the body silhouette -- 2 B per voxel in one pass
// (7 tissues x 2 axes = 14 full-volume numpy passes in the reference).  One block per z slice; a thread owns its x
// columns (coronal counts need no atomics), the sagittal row counts are wave ballots + one LDS add per wave and tissue.
struct ProjArgs {
    const unsigned char* tissues;
    const unsigned char* regions;
    int Y, X, T;
    unsigned int* cor;       // [T][Z][X]
    unsigned int* sag;       // [T][Z][Y]
    unsigned char* mcor;     // [Z][X]
    unsigned char* msag;     // [Z][Y]
    int Z;
    unsigned char lut[256];  // tissue value -> index in [0, T) or 255
};

__global__ __launch_bounds__(256) void k_tissue_projections(ProjArgs a) {
    extern __shared__ unsigned int sm[];
    unsigned int* s_cor = sm;                               // [T][X]
    unsigned int* s_sag = sm + (size_t)a.T * a.X;           // [T][Y]
    unsigned int* s_mc = s_sag + (size_t)a.T * a.Y;         // [X]
    unsigned int* s_ms = s_mc + a.X;                        // [Y]
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int tot = a.T * (a.X + a.Y) + a.X + a.Y;
    for (int i = tid; i < tot; i += 256) sm[i] = 0;
    __syncthreads();
    const size_t base = (size_t)z * a.Y * a.X;
    const int xr = (a.X + 255) / 256 * 256;  // whole waves so that the ballots see every lane
    for (int y = 0; y < a.Y; ++y) {
        const unsigned char* trow = a.tissues + base + (size_t)y * a.X;
        const unsigned char* rrow = a.regions + base + (size_t)y * a.X;
        for (int x = tid; x < xr; x += 256) {
            int idx = 255;
            bool body = false;
            if (x < a.X) {
                idx = a.lut[trow[x]];
                const unsigned char r = rrow[x];
                body = r > 0 && r < 255;
                if (idx < a.T) s_cor[idx * a.X + x] += 1;   // this thread is the only writer of column x
                if (body) s_mc[x] = 1;
            }
            for (int t = 0; t < a.T; ++t) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(idx == t);
                if (m && lane == 0) atomicAdd(&s_sag[t * a.Y + y], (unsigned)__builtin_popcountll(m));
            }
            if (__builtin_amdgcn_ballot_w64(body) && lane == 0) s_ms[y] = 1;
        }
    }
    __syncthreads();
    for (int i = tid; i < a.T * a.X; i += 256) a.cor[((size_t)(i / a.X) * a.Z + z) * a.X + i % a.X] = s_cor[i];
    for (int i = tid; i < a.T * a.Y; i += 256) a.sag[((size_t)(i / a.Y) * a.Z + z) * a.Y + i % a.Y] = s_sag[i];
    for (int i = tid; i < a.X; i += 256) a.mcor[(size_t)z * a.X + i] = (unsigned char)s_mc[i];
    for (int i = tid; i < a.Y; i += 256) a.msag[(size_t)z * a.Y + i] = (unsigned char)s_ms[i];
}

extern "C" int boa_tissue_projections(boa_ctx* c, const uint8_t* dev_tissues, const uint8_t* dev_regions, int Z, int Y, int X,
                                      const uint8_t* host_values, int n_values, uint32_t* dev_coronal, uint32_t* dev_sagittal,
                                      uint8_t* dev_mask_coronal, uint8_t* dev_mask_sagittal) {
    BOA_REQUIRE(c && dev_tissues && dev_regions && host_values && dev_coronal && dev_sagittal && dev_mask_coronal &&
                    dev_mask_sagittal && Z > 0 && Y > 0 && X > 0,
                "boa_tissue_projections: bad argument");
    BOA_REQUIRE(n_values >= 1 && n_values <= 16, "boa_tissue_projections: %d tissue values (1..16)", n_values);
    const size_t lds = ((size_t)n_values * (X + Y) + X + Y) * 4;
    BOA_REQUIRE(lds <= 160 * 1024, "boa_tissue_projections: slice %dx%d with %d tissues needs %zu bytes of LDS", Y, X, n_values, lds);
    ProjArgs a;
    a.tissues = dev_tissues; a.regions = dev_regions; a.Y = Y; a.X = X; a.T = n_values; a.Z = Z;
    a.cor = dev_coronal; a.sag = dev_sagittal; a.mcor = dev_mask_coronal; a.msag = dev_mask_sagittal;
    for (int i = 0; i < 256; ++i) a.lut[i] = 255;
    for (int t = 0; t < n_values; ++t) a.lut[host_values[t]] = (unsigned char)t;
    static bool once = (hipFuncSetAttribute((const void*)k_tissue_projections, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    KernelTimer t(c, BOA_K_AGG, 0, 2.0 * Z * Y * X);
    hipLaunchKernelGGL(k_tissue_projections, dim3(Z), dim3(256), lds, c->stream, a);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// per-label HU histogram (label 0 = background is never measured by the reference and is skipped)
// Per-label HU histogram with a workgroup-private HASH TABLE in LDS.  Per-voxel global atomics are hopeless here: a CT's
// histogram is concentrated (a label's voxels fall into ~100 neighbouring HU bins, a handful of cache lines), and
// device-scope atomics from the 8 XCDs on the same lines are serialised on the memory side (57 ms per pass at 512^3,
// however the voxels are batched).  Each workgroup counts its voxels into an open-addressing table keyed by
// (label, bin) -- HIST_TAB entries of {key, count} in LDS, linear probing, claimed with an LDS compare-and-swap -- whatever
// the mix of labels (compact organs or the salt-and-pepper output of a random-weight net), and adds the table to the
// global one when it fills up and at the end: one global atomic per DISTINCT key of ~10^5 voxels instead of one per
// voxel.  A wave whose 1 024 voxels all carry the same key (air around the patient) issues one LDS atomic, or none when
// the key is "not measured".
// Table size: 2^LOG2 entries of {key, count}.  LOG2 = 14 (128 KiB) leaves ONE workgroup (4 waves) per CU -- the voxel loads of an
// iteration are then a chain of exposed HBM round trips; LOG2 = 12 (32 KiB) lets four workgroups share a CU and still holds the
// distinct (label, HU) keys of a contiguous voxel range of compact organs between flushes; LOG2 = 13 (two workgroups) is the default.
#define HIST_EMPTY 0xFFFFFFFFu

// HT: threads per workgroup.  What bounds the pass on compact organs is the FLUSH: every workgroup sends each distinct (label, HU) key of
// its voxel range to the global table with one device-scope atomic, and those serialise in the fabric (the reason for the LDS tables in the
// first place) -- 2 048 workgroups of 256 threads re-send the same ~1 000 keys 2 048 times.  1 024-thread workgroups cover 4x the voxels
// per table at the same waves per CU: a quarter of the workgroups, a little more than a quarter of the atomics.
template <int LOG2, int HT>
__global__ __launch_bounds__(HT) void k_label_hist(const short* __restrict__ ct, const unsigned char* __restrict__ labels,
                                                    const unsigned char* __restrict__ mask, size_t n_all, size_t head, int hu_min,
                                                    int nbins, unsigned int* __restrict__ hist, size_t groups_per_block) {
    constexpr int HIST_TAB = 1 << LOG2;
    constexpr int HIST_FLUSH = HIST_TAB * 2 / 3;     // distinct keys in the table that trigger a flush (load factor 2/3)
    extern __shared__ __attribute__((aligned(16))) unsigned char hist_smem[];
    unsigned int* keys = (unsigned int*)hist_smem;   // [HIST_TAB]
    unsigned int* cnts = keys + HIST_TAB;            // [HIST_TAB]
    __shared__ int nkeys;
    const int tid = threadIdx.x;
    for (int i = tid; i < HIST_TAB; i += HT) {
        keys[i] = HIST_EMPTY;
        cnts[i] = 0u;
    }
    if (tid == 0) nkeys = 0;
    __syncthreads();
    // voxels [0, head) and the last (n - head) % 16 are handled one by one (unaligned views: z-slabs of a volume)
    const unsigned char* labels0 = labels;
    const short* ct0 = ct;
    const unsigned char* mask0 = mask;
    labels += head;
    ct += head;
    if (mask) mask += head;
    const size_t n = n_all - head;
    const size_t n16 = n / 16;
    auto bin_of = [&](int hu) {
        int b = hu - hu_min;
        return b < 0 ? 0 : (b >= nbins ? nbins - 1 : b);
    };
    auto count = [&](unsigned key, unsigned c) {  // key = label << 16 | bin  (nbins <= 65536)
        unsigned h = (key * 2654435761u) >> (32 - LOG2);
#pragma unroll 1
        for (int probe = 0; probe < 16; ++probe, h = (h + 1) & (HIST_TAB - 1)) {
            unsigned k = keys[h];
            if (k == HIST_EMPTY) {
                k = atomicCAS(&keys[h], HIST_EMPTY, key);
                if (k == HIST_EMPTY) {
                    atomicAdd(&nkeys, 1);
                    k = key;
                }
            }
            if (k == key) {
                atomicAdd(&cnts[h], c);
                return;
            }
        }
        atomicAdd(&hist[(size_t)(key >> 16) * nbins + (key & 0xFFFFu)], c);   // a long probe chain: straight to the global table
    };
    auto flush = [&]() {  // whole workgroup
        for (int i = tid; i < HIST_TAB; i += HT) {
            const unsigned k = keys[i];
            if (k != HIST_EMPTY) {
                atomicAdd(&hist[(size_t)(k >> 16) * nbins + (k & 0xFFFFu)], cnts[i]);
                keys[i] = HIST_EMPTY;
                cnts[i] = 0u;
            }
        }
        __syncthreads();
        if (tid == 0) nkeys = 0;
        __syncthreads();
    };
    const size_t g_begin = (size_t)blockIdx.x * groups_per_block * HT;   // groups of 16 voxels, HT per iteration
    // the next iteration's 64 bytes per lane are fetched BEFORE this iteration's table work (round 6): with 32 - 128 KiB of LDS per
    // workgroup only 1 - 4 workgroups fit a CU, and a loop of load -> wait -> LDS work -> barrier left HBM idle most of the time
    union LabV {
        uint4 u;
        unsigned char b[16];
    };
    union HuV {
        uint4 u[2];
        short h[16];
    };
    LabV nlb, nmb;
    HuV nhb;
    nlb.u = nmb.u = nhb.u[0] = nhb.u[1] = make_uint4(0, 0, 0, 0);
    auto fetch = [&](size_t it) {
        const size_t g = g_begin + it * HT + tid;
        if (it < groups_per_block && g < n16) {
            nlb.u = *(const uint4*)(labels + g * 16);
            nhb.u[0] = *(const uint4*)(ct + g * 16);
            nhb.u[1] = *(const uint4*)(ct + g * 16 + 8);
            if (mask) nmb.u = *(const uint4*)(mask + g * 16);
        }
    };
    fetch(0);
    for (size_t it = 0; it < groups_per_block; ++it) {
        const size_t g = g_begin + it * HT + tid;
        const bool live = g < n16;
        unsigned key[16];   // 0 = not measured (label 0 / masked out)
        const LabV lb = nlb, mb = nmb;
        const HuV hb = nhb;
        fetch(it + 1);
        if (live) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned l = (mask && !mb.b[i]) ? 0u : lb.b[i];
                key[i] = l ? ((l << 16) | (unsigned)bin_of(hb.h[i])) : 0u;   // label >= 1: never 0
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) key[i] = 0u;
        }
        // whole wave on one key: one atomic (or none)
        bool uni = true;
#pragma unroll
        for (int i = 1; i < 16; ++i) uni = uni && key[i] == key[0];
        const unsigned k0 = __builtin_amdgcn_readfirstlane(key[0]);
        const bool same = live ? (uni && key[0] == k0) : true;
        if (__builtin_amdgcn_ballot_w64(!same) == 0) {
            const unsigned c = 16u * (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live));
            if (k0 != 0u && (tid & 63) == 0 && c) count(k0, c);
        } else if (live) {
#ifdef BOA_HIST_SERIAL   // (A/B: the round-5 form, one dependent probe chain per run of equal keys)
            unsigned rk = key[0], run = 1;
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                if (key[i] == rk) {
                    ++run;
                } else {
                    if (rk) count(rk, run);
                    rk = key[i];
                    run = 1;
                }
            }
            if (rk) count(rk, run);
#else
            // Round 6: the 16 voxels of a lane are looked up TOGETHER.  In the steady state a key already sits in its home slot (a slot
            // keeps its key until the next flush, and flushes are behind the workgroup barrier): 16 independent ds_read_b32 of the home
            // slots, then one return-less ds_add_u32 per hit -- throughput instead of 16 dependent LDS round trips with a probe loop
            // each.  Misses (first occurrence of a key, displaced keys) take the probing insert.  Runs of equal keys are merged first
            // (CT noise makes them rare inside an organ, but label 0 / masked voxels form long runs of key 0, which cost nothing).
            unsigned hh[16], kk[16], cc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                hh[i] = (key[i] * 2654435761u) >> (32 - LOG2);
                cc[i] = 1u;
            }
#pragma unroll
            for (int i = 15; i > 0; --i) {   // fold a voxel into its left neighbour when the keys agree (counts flow to the run's first voxel)
                const bool eq = key[i] == key[i - 1];
                cc[i - 1] += eq ? cc[i] : 0u;
                cc[i] = eq ? 0u : cc[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) kk[i] = keys[hh[i]];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (key[i] != 0u && cc[i] != 0u) {
                    if (kk[i] == key[i])
                        atomicAdd(&cnts[hh[i]], cc[i]);
                    else
                        count(key[i], cc[i]);
                }
            }
#endif
        }
        __syncthreads();
        if (nkeys > HIST_FLUSH) flush();   // (uniform: nkeys is read after the barrier by everyone)
    }
    // tail (n % 16 voxels) and head, one by one straight into the global table
    if (blockIdx.x == 0) {
        if (tid < (int)(n - n16 * 16)) {
            const size_t i = n16 * 16 + tid;
            const int l = (mask && !mask[i]) ? 0 : labels[i];
            if (l) atomicAdd(&hist[(size_t)l * nbins + bin_of(ct[i])], 1u);
        }
        if (tid < (int)head) {
            const int l = (mask0 && !mask0[tid]) ? 0 : labels0[tid];
            if (l) atomicAdd(&hist[(size_t)l * nbins + bin_of(ct0[tid])], 1u);
        }
    }
    __syncthreads();
    flush();
}

__global__ __launch_bounds__(256) void k_label_hist_scalar(const short* __restrict__ ct, const unsigned char* __restrict__ labels,
                                                           const unsigned char* __restrict__ mask, size_t n, int hu_min, int nbins,
                                                           unsigned int* __restrict__ hist) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const int l = labels[i];
        if (l == 0 || (mask && !mask[i])) continue;
        int b = (int)ct[i] - hu_min;
        b = b < 0 ? 0 : (b >= nbins ? nbins - 1 : b);
        atomicAdd(&hist[(size_t)l * nbins + b], 1u);
    }
}

extern "C" int boa_label_hu_histogram(boa_ctx* c, const int16_t* dev_ct, const uint8_t* dev_labels,
                                      const uint8_t* dev_mask, size_t n, int hu_min, int nbins, uint32_t* dev_hist) {
    BOA_REQUIRE(c && dev_ct && dev_labels && dev_hist && nbins > 0 && nbins <= 65536, "boa_label_hu_histogram: bad argument");
    BOA_HIP_TRY(hipMemsetAsync(dev_hist, 0, (size_t)256 * nbins * sizeof(uint32_t), c->stream));
    if (n == 0) return BOA_OK;
    // 16-byte vector loads need the three arrays aligned at the same voxel: skip `head` voxels (views into a volume start
    // anywhere); arrays that cannot be aligned together are processed one voxel at a time
    size_t head = (size_t)((16 - ((uintptr_t)dev_labels & 15)) & 15);
    if (head > n) head = n;
    const bool together = (((uintptr_t)dev_ct + 2 * head) & 15) == 0 && (!dev_mask || (((uintptr_t)dev_mask + head) & 15) == 0);
    // contiguous voxel ranges per workgroup (few labels each): ~4 workgroups per CU, at least one 4 096-voxel iteration
    // table size / workgroups per CU, kernel time in us (tools/hist_sweep.sh, tools/hist_bench.sh; 512^3):   structured phantom | bench labels (noise-like)
    //   2^14, 4:  845 | 947      2^13, 8:  679 | 1 124      2^12, 8:  571 | 1 905      2^12, 16:  650 | 2 056
    // compact organs want occupancy, salt-and-pepper labels a table that merges more duplicates before it spills: 2^13 is the default
    // round 6 (tools/r6_hist_sweep.sh, profiles/r06_hist_sweep.txt; kernel us on the structured phantom | on salt-and-pepper labels, 512^3):
    //   2^13 x 256 threads x 8 per CU (round 5)  638 | 2 937      2^13 x 1 024 x 2   486 | 2 600      2^12 x 1 024 x 2   547 | 2 650
    //   2^14 x 1 024 x 1  420 | 2 106  <- default: ONE 1 024-thread workgroup per CU with the largest table = the fewest flush atomics
    // ($BOA_HIST_THREADS=256 restores the small workgroups)
    static const int hist_log2 = getenv("BOA_HIST_LOG2") ? atoi(getenv("BOA_HIST_LOG2")) : 14;
    static const int hist_threads = getenv("BOA_HIST_THREADS") ? atoi(getenv("BOA_HIST_THREADS")) : 1024;
    const int HT = hist_threads == 256 ? 256 : (hist_threads == 512 ? 512 : 1024);
    static const int hist_wg = getenv("BOA_HIST_WG") ? atoi(getenv("BOA_HIST_WG")) : 0;   // workgroups per CU (0: what fits next to each other)
    const int wg = hist_wg > 0 ? hist_wg : (HT == 256 ? (hist_log2 == 14 ? 4 : 8) : std::max(1, std::min((160 * 1024) / (8 << hist_log2), 2048 / HT)));
    const size_t iters = ((n - head) / 16 + HT - 1) / HT;
    const size_t gpb = std::max<size_t>(1, (iters + (size_t)c->cu_count * wg - 1) / ((size_t)c->cu_count * wg));
    const int grid = (int)std::max<size_t>(1, (iters + gpb - 1) / gpb);
    KernelTimer t(c, BOA_K_AGG, 0, (double)n * (3.0 + (dev_mask ? 1 : 0)));
    if (together) {
#define BOA_HIST_LAUNCH(L, T)                                                                                                         \
    do {                                                                                                                              \
        static bool once = (hipFuncSetAttribute((const void*)k_label_hist<L, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024), true); \
        (void)once;                                                                                                                   \
        hipLaunchKernelGGL((k_label_hist<L, T>), dim3(grid), dim3(T), (size_t)8 << L, c->stream, dev_ct, dev_labels, dev_mask, n, head, hu_min, \
                           nbins, dev_hist, gpb);                                                                                     \
    } while (0)
#define BOA_HIST_BY_T(L)                                  \
    do {                                                  \
        if (HT == 256) BOA_HIST_LAUNCH(L, 256);           \
        else if (HT == 512) BOA_HIST_LAUNCH(L, 512);      \
        else BOA_HIST_LAUNCH(L, 1024);                    \
    } while (0)
        if (hist_log2 == 14)
            BOA_HIST_BY_T(14);
        else if (hist_log2 == 13)
            BOA_HIST_BY_T(13);
        else
            BOA_HIST_BY_T(12);
#undef BOA_HIST_BY_T
#undef BOA_HIST_LAUNCH
    } else {
        hipLaunchKernelGGL(k_label_hist_scalar, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)c->cu_count * 32)), dim3(256), 0,
                           c->stream, dev_ct, dev_labels, dev_mask, n, hu_min, nbins, dev_hist);
    }
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
struct Lut256 {
    unsigned char v[256];
};

__global__ __launch_bounds__(256) void k_label_hu_mask(const short* __restrict__ ct, const unsigned char* __restrict__ labels,
                                                       Lut256 lut, int mode, int lo, int hi, size_t n,
                                                       unsigned char* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        bool m = lut.v[labels[i]] != 0;
        if (mode != 0) {
            const int hu = ct[i];
            const bool inside = hu >= lo && hu <= hi;
            m = m && (mode == 1 ? inside : !inside);
        }
        out[i] = m ? 1 : 0;
    }
}

extern "C" int boa_label_hu_mask(boa_ctx* c, const int16_t* dev_ct, const uint8_t* dev_labels, const uint8_t* host_lut,
                                 int mode, int hu_lo, int hu_hi, size_t n, uint8_t* dev_mask_out) {
    BOA_REQUIRE(c && dev_labels && host_lut && dev_mask_out && (mode == 0 || dev_ct), "boa_label_hu_mask: bad argument");
    BOA_REQUIRE(mode >= 0 && mode <= 2, "boa_label_hu_mask: mode %d", mode);
    if (n == 0) return BOA_OK;
    Lut256 lut;
    memcpy(lut.v, host_lut, 256);
    int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)c->cu_count * 32);
    KernelTimer t(c, BOA_K_AGG, 0, (double)n * 4.0);
    hipLaunchKernelGGL(k_label_hu_mask, dim3(grid), dim3(256), 0, c->stream, dev_ct, dev_labels, lut, mode, hu_lo, hu_hi,
                       n, dev_mask_out);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

__global__ __launch_bounds__(256) void k_label_select(const unsigned char* __restrict__ labels, size_t n, int mode,
                                                      int a, int b, int cc, unsigned char* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const int l = labels[i];
        bool m;
        if (mode == 0)
            m = l == a;
        else if (mode == 1)
            m = l > 0;
        else
            m = (l == a) || (l == b) || (l == cc);
        out[i] = m ? 1 : 0;
    }
}

extern "C" int boa_label_select(boa_ctx* c, const uint8_t* dev_labels, size_t n, int mode, const int vals[3],
                                uint8_t* dev_mask_out) {
    BOA_REQUIRE(c && dev_labels && dev_mask_out && mode >= 0 && mode <= 2, "boa_label_select: bad argument");
    BOA_REQUIRE(mode == 1 || vals, "boa_label_select: vals is NULL");
    if (n == 0) return BOA_OK;
    int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)c->cu_count * 32);
    hipLaunchKernelGGL(k_label_select, dim3(grid), dim3(256), 0, c->stream, dev_labels, n, mode, vals ? vals[0] : 0,
                       vals ? vals[1] : 0, vals ? vals[2] : 0, dev_mask_out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// erode_region: AND over offsets [-(k/2) .. k - 1 - k/2] for even k (footprint padded at the end), symmetric for
// odd k; outside the volume counts as set.  One pass per axis.
__global__ __launch_bounds__(256) void k_erode_axis(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                    int Z, int Y, int X, int axis, int lo, int hi) {
    const size_t n = (size_t)Z * Y * X;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    idx3(i, Y, X, z, y, x);
    const int pos = axis == 0 ? z : (axis == 1 ? y : x);
    const int len = axis == 0 ? Z : (axis == 1 ? Y : X);
    const size_t st = axis == 0 ? (size_t)Y * X : (axis == 1 ? (size_t)X : 1);
    unsigned char r = 1;
    for (int d = lo; d <= hi; ++d) {
        const int q = pos + d;
        if (q < 0 || q >= len) continue;
        r &= (in[i + (long long)d * (long long)st] != 0) ? 1 : 0;
    }
    out[i] = r;
}

extern "C" int boa_bits_erode_u8(boa_ctx* c, const uint8_t* dev_mask, uint8_t* dev_out, int Z, int Y, int X, int lo, int hi);

extern "C" int boa_binary_erode(boa_ctx* c, const uint8_t* dev_mask, uint8_t* dev_out, uint8_t* dev_tmp, int Z, int Y,
                                int X, int kernel_value) {
    BOA_REQUIRE(c && dev_mask && dev_out && dev_tmp && kernel_value >= 1, "boa_binary_erode: bad argument");
    BOA_REQUIRE(dev_out != dev_mask && dev_tmp != dev_mask && dev_tmp != dev_out, "boa_binary_erode: buffers must differ");
    const int k = kernel_value;
    const int center = (k % 2 == 0) ? (k + 1) / 2 : k / 2;  // centre of the (padded) footprint
    const int lo = -center, hi = k - 1 - center;
    const size_t n = (size_t)Z * Y * X;
    // bit-mask form (csrc/ccl_bits.hip): 1 byte read + 1 byte written per voxel and three passes over 1 / 8 byte per voxel, instead of
    // three byte passes of one thread per voxel (1.9 ms -> 0.2 ms per 512^3 mask); $BOA_ERODE_BYTES=1 keeps the byte passes
    static const bool bytes_only = getenv("BOA_ERODE_BYTES") != nullptr;
    if (!bytes_only && lo > -32 && hi < 32) return boa_bits_erode_u8(c, dev_mask, dev_out, Z, Y, X, lo, hi);
    unsigned grid = (unsigned)((n + 255) / 256);
    KernelTimer t(c, BOA_K_AGG, 0, (double)n * 6.0);
    hipLaunchKernelGGL(k_erode_axis, dim3(grid), dim3(256), 0, c->stream, dev_mask, dev_out, Z, Y, X, 2, lo, hi);
    hipLaunchKernelGGL(k_erode_axis, dim3(grid), dim3(256), 0, c->stream, dev_out, dev_tmp, Z, Y, X, 1, lo, hi);
    hipLaunchKernelGGL(k_erode_axis, dim3(grid), dim3(256), 0, c->stream, dev_tmp, dev_out, Z, Y, X, 0, lo, hi);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// 26-connected components by atomic union-find (roots = smallest linear index of each component)
#define AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__device__ __forceinline__ int uf_find(int* L, int i) {
    // path halving: every node on the way is re-pointed to its grandparent (parents only ever move towards the root, so a racing
    // walker at worst takes the longer way).  The store is an agent-scope atomic store like every other access to L in the
    // union kernels: a plain store stays dirty in the L2 of the XCD that issued it, and when that line is written back it can
    // take stale copies of NEIGHBOURING words with it -- words that another XCD's atomicMin has meanwhile changed at the
    // memory side.  Measured: with plain stores 1 labelling in ~200 lost one union (a voxel keeps a root that was merged away)
    // whenever a second stream kept the GPU busy (tools/ccl_stress.py, tests/test_gpu_lanes.py); with atomic stores 0 in 2 400.
    int p = AGENT_LOAD(&L[i]);
    while (p != i) {
        const int gp = AGENT_LOAD(&L[p]);
        if (gp != p) __hip_atomic_store(&L[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i = p;
        p = gp;
    }
    return i;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    while (true) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(&L[a], b);  // link the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_ccl_init(const unsigned char* __restrict__ mask, size_t n, int* __restrict__ L) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) L[i] = mask[i] ? (int)i : -1;
}

__global__ __launch_bounds__(256) void k_ccl_merge(const unsigned char* __restrict__ mask, int Z, int Y, int X, int* L) {
    const size_t n = (size_t)Z * Y * X;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !mask[i]) return;
    const int x = (int)(i % X);
    const int y = (int)((i / X) % Y);
    const int z = (int)(i / ((size_t)X * Y));
    // The 13 neighbours with larger linear index: (x+1) in this row and the x-1, x, x+1 triples of the four "later" rows
    // (dz,dy) = (0,1), (1,-1), (1,0), (1,1).  When the left neighbour (x-1) of this row is foreground it is in our component
    // and has already linked itself to the x-2, x-1, x voxels of those rows, so only their x+1 voxel is new information;
    // likewise within a triple one link is enough when consecutive voxels of the later row are foreground (they are linked to
    // each other by that row's own x+1 link).  This cuts the unions from up to 13 to ~5 per voxel without changing the
    // components.
    const bool left = x > 0 && mask[i - 1];
    if (x + 1 < X && mask[i + 1]) uf_union(L, (int)i, (int)(i + 1));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int dz = r == 0 ? 0 : 1, dy = r == 0 ? 1 : r - 2;
        const int zz = z + dz, yy = y + dy;
        if (zz >= Z || yy < 0 || yy >= Y) continue;
        const size_t row = ((size_t)zz * Y + yy) * X;
        const bool m0 = x > 0 && mask[row + x - 1], m1 = mask[row + x] != 0, m2 = x + 1 < X && mask[row + x + 1];
        if (!left) {
            if (m1) {
                uf_union(L, (int)i, (int)(row + x));            // x-1 and x+1 of that row hang on its x voxel
            } else {
                if (m0) uf_union(L, (int)i, (int)(row + x - 1));
                if (m2) uf_union(L, (int)i, (int)(row + x + 1));
            }
        } else if (m2 && !m1) {
            uf_union(L, (int)i, (int)(row + x + 1));            // (with m1 set, x+1 is linked to x, which the left voxel linked)
        }
    }
}

// ---- two-level labelling: tiles of CCL_TX x CCL_TY x CCL_TZ voxels are labelled in LDS first, then only the unions that
// cross a tile face go through global memory.  The per-voxel global version above spends its time in device-scope atomics and
// pointer chasing through HBM (5.7 ms per 512^3 mask); inside a tile the same union-find runs on LDS words.  The result is the
// same forest invariant (parent index <= own index, root = smallest linear index of the component), so k_ccl_compress and
// everything downstream see identical roots and sizes.
#define CCL_TX 32
#define CCL_TY 16
#define CCL_TZ 16
#define CCL_TILE (CCL_TX * CCL_TY * CCL_TZ)

__device__ __forceinline__ int lds_find(volatile int* L, int i) {
    int p = L[i];
    while (p != i) {
        const int gp = L[p];
        if (gp != p) L[i] = gp;  // path halving (a racing walker at worst takes the longer way)
        i = p;
        p = gp;
    }
    return i;
}

__device__ __forceinline__ void lds_union(int* L, int a, int b) {
    while (true) {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_ccl_local(const unsigned char* __restrict__ mask, int Z, int Y, int X, int tiles_x, int tiles_y,
                                                   int* __restrict__ L, unsigned int* __restrict__ sizes) {
    __shared__ int lab[CCL_TILE];  // union-find parents; reused for the component sizes once every voxel knows its root
    __shared__ unsigned int rowbits[CCL_TY * CCL_TZ];  // bit lx of word (lz, ly): voxel is foreground
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, tz = t / tiles_y;
    const int x0 = tx * CCL_TX, y0 = ty * CCL_TY, z0 = tz * CCL_TZ;
    // one wave-wide ballot per row pair: lane (row r of the pair, lx)
    for (int r2 = tid >> 5; r2 < CCL_TY * CCL_TZ; r2 += 8) {
        const int lx = tid & 31, ly = r2 % CCL_TY, lz = r2 / CCL_TY;
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        const bool fg = x < X && y < Y && z < Z && mask[((size_t)z * Y + y) * X + x] != 0;
        const unsigned long long b = __ballot(fg);
        if (lx == 0) rowbits[r2] = (unsigned int)(b >> (32 * ((tid >> 5) & 1)));
    }
    __syncthreads();
    // uniform tiles (all background, or a full tile of foreground: one component rooted at its first voxel) skip the union-find:
    // body-sized masks and their inverses are mostly such tiles.  Same forest as the general path (root = smallest index).
    {
        unsigned int w_and = 0xffffffffu, w_or = 0u;
        for (int r2 = tid; r2 < CCL_TY * CCL_TZ; r2 += 256) {
            w_and &= rowbits[r2];
            w_or |= rowbits[r2];
        }
        const int all0 = __syncthreads_and(w_or == 0u);
        const int all1 = __syncthreads_and(w_and == 0xffffffffu);
        if (all0 || all1) {
            const int root = (int)(((size_t)z0 * Y + y0) * X + x0);
#pragma unroll
            for (int k = 0; k < CCL_TILE / 256; ++k) {
                const int r2 = (tid >> 5) + 8 * k, lx = tid & 31, ly = r2 % CCL_TY, lz = r2 / CCL_TY;
                const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
                if (x >= X || y >= Y || z >= Z) continue;   // (all1 implies the tile lies inside the volume)
                const size_t gi = ((size_t)z * Y + y) * X + x;
                L[gi] = all1 ? root : -1;
                sizes[gi] = (all1 && r2 == 0 && lx == 0) ? (unsigned int)CCL_TILE : 0u;
            }
            return;
        }
    }
    // parents start at the first voxel of the voxel's x-run (the runs of a row are its components: no unions along x at all)
    for (int r2 = tid >> 5; r2 < CCL_TY * CCL_TZ; r2 += 8) {
        const int lx = tid & 31;
        const unsigned int me = rowbits[r2];
        const unsigned int starts = me & ~(me << 1);                       // first voxel of every run
        const unsigned int upto = starts & (0xffffffffu >> (31 - lx));     // run starts at or left of lx
        lab[r2 * CCL_TX + lx] = ((me >> lx) & 1u) ? r2 * CCL_TX + (31 - __clz((int)upto)) : -1;
    }
    __syncthreads();
    // unions between the runs of neighbouring rows (the four forward rows (dz, dy) = (0, 1), (1, -1), (1, 0), (1, 1)): ONE union per
    // pair of touching runs -- at the first voxel where both rows are set, or, for runs that only touch diagonally, at the run end
    // facing the other run.  (The first version linked every voxel to the voxel below it: ~5 LDS union-finds per voxel.)
    for (int r2 = tid >> 5; r2 < CCL_TY * CCL_TZ; r2 += 8) {
        const int lx = tid & 31, ly = r2 % CCL_TY, lz = r2 / CCL_TY;
        const unsigned int me = rowbits[r2];
        if (!((me >> lx) & 1u)) continue;
        const int i = r2 * CCL_TX + lx;
        const bool a_l = lx > 0 && ((me >> (lx - 1)) & 1u), a_r = lx + 1 < CCL_TX && ((me >> (lx + 1)) & 1u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dz = r == 0 ? 0 : 1, dy = r == 0 ? 1 : r - 2;
            const int zz = lz + dz, yy = ly + dy;
            if (zz >= CCL_TZ || yy < 0 || yy >= CCL_TY) continue;
            const int rr = zz * CCL_TY + yy;
            const unsigned int w = rowbits[rr];
            const bool m0 = lx > 0 && ((w >> (lx - 1)) & 1u), m1 = (w >> lx) & 1u, m2 = lx + 1 < CCL_TX && ((w >> (lx + 1)) & 1u);
            const int row = rr * CCL_TX;
            if (m1) {
                if (!(a_l && m0)) lds_union(lab, i, row + lx);          // first voxel of the overlap of the two runs
            } else {
                if (m2 && !a_r) lds_union(lab, i, row + lx + 1);        // my run ends here, the other starts diagonally
                if (m0 && !a_l) lds_union(lab, i, row + lx - 1);        // my run starts here, the other ends diagonally
            }
        }
    }
    __syncthreads();
    // global labels: the tile-local root's linear index in the volume; voxel counts of the local components (LDS atomics:
    // the per-voxel global atomics of the one-level version were its second most expensive part)
    int myroot[CCL_TILE / 256];
#pragma unroll
    for (int k = 0; k < CCL_TILE / 256; ++k) {
        const int r2 = (tid >> 5) + 8 * k, lx = tid & 31;
        myroot[k] = -1;
        if ((rowbits[r2] >> lx) & 1u) myroot[k] = lds_find(lab, r2 * CCL_TX + lx);
    }
    __syncthreads();
    unsigned int* cnt = (unsigned int*)lab;  // (a second 32 KiB array halved the occupancy: 2.1 -> 4.0 ms per 512^3 mask)
#pragma unroll
    for (int k = 0; k < CCL_TILE / 256; ++k) cnt[tid + 256 * k] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CCL_TILE / 256; ++k) {
        const int lx = tid & 31;
        // one LDS atomic per run of equal roots in the row (a solid tile would otherwise put 8 192 atomics on one word)
        const int prev = __shfl_up(myroot[k], 1);
        const bool lead = lx == 0 || prev != myroot[k];
        const unsigned int leads = (unsigned int)(__ballot(lead) >> (32 * ((tid >> 5) & 1)));  // this row's half of the wave
        if (lead && myroot[k] >= 0) {
            const unsigned int after = lx == 31 ? 0u : (leads >> (lx + 1));
            const int len = after ? __ffs((int)after) : 32 - lx;
            atomicAdd(&cnt[myroot[k]], (unsigned int)len);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CCL_TILE / 256; ++k) {
        const int r2 = (tid >> 5) + 8 * k, lx = tid & 31, ly = r2 % CCL_TY, lz = r2 / CCL_TY;
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        if (x >= X || y >= Y || z >= Z) continue;
        int out = -1;
        if (myroot[k] >= 0) {
            const int rt = myroot[k];
            const int rx = rt % CCL_TX, rr = rt / CCL_TX;
            out = (int)(((size_t)(z0 + rr / CCL_TY) * Y + (y0 + rr % CCL_TY)) * X + (x0 + rx));
        }
        const size_t gi = ((size_t)z * Y + y) * X + x;
        L[gi] = out;
        sizes[gi] = cnt[r2 * CCL_TX + lx];  // > 0 only at tile-local roots (every voxel is written: no memset of `sizes`)
    }
}

// after the border unions: every voxel points at its global root; a tile-local root that is not the global root hands its count
// over (one global atomic per tile-local component instead of one per voxel)
__global__ __launch_bounds__(256) void k_ccl_resolve(size_t n, int* L, unsigned int* sizes, int* n_comp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int is_root = 0;
    unsigned int pend_c = 0u;
    int pend_root = 0;
    if (i < n) {
        const int p0 = L[i];
        if (p0 >= 0) {
            int root = p0, p = L[root];
            while (p != root) {
                root = p;
                p = L[root];
            }
            if (root != p0) L[i] = root;
            if (root == (int)i) {
                is_root = 1;
            } else {
                const unsigned int c = sizes[i];
                if (c) {
                    pend_c = c;
                    pend_root = root;
                    // (agent-scope store, not a plain one: the same line may hold a root's count that other XCDs are adding to)
                    __hip_atomic_store(&sizes[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    // hand the counts over: body-sized masks are ONE giant component, so nearly every tile-local root of the volume adds to the same
    // word -- a device-scope atomic per tile-local component serialises at that address (resolve took 0.2 ms on a mask of scattered
    // specks and 2 ms on a solid one).  Two rounds of wave-level aggregation on the most common root of the wave, then the rest one
    // by one.
    {
        const int lane = threadIdx.x & 63;
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            const unsigned long long act = __ballot(pend_c != 0u);
            if (!act) break;
            const int leader = __ffsll((long long)act) - 1;
            const int r0 = __shfl(pend_root, leader);
            const bool mine = pend_c != 0u && pend_root == r0;
            unsigned int sum = mine ? pend_c : 0u;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
            if (lane == leader) atomicAdd(&sizes[r0], sum);
            if (mine) pend_c = 0u;
        }
        if (pend_c) atomicAdd(&sizes[pend_root], pend_c);
    }
    const unsigned long long b = __ballot(is_root);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_comp, __popcll(b));
}

// unions across tile faces only: voxels on the faces x = 0 (its x - 1 neighbours in the later rows), x = TX-1, y = 0 (the
// (dz, dy) = (1, -1) row), y = TY-1 and z = TZ-1 of their tile have forward neighbours in another tile
__device__ __forceinline__ void ccl_border_voxel(const unsigned char* __restrict__ mask, int Z, int Y, int X, int* L, int x, int y, int z) {
    const size_t i = ((size_t)z * Y + y) * X + x;
    const int lx = x % CCL_TX, ly = y % CCL_TY, lz = z % CCL_TZ;
    if (!mask[i]) return;
    if (lx == CCL_TX - 1 && x + 1 < X && mask[i + 1]) uf_union(L, (int)i, (int)(i + 1));
    const bool left = x > 0 && mask[i - 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int dz = r == 0 ? 0 : 1, dy = r == 0 ? 1 : r - 2;
        const int zz = z + dz, yy = y + dy;
        if (zz >= Z || yy < 0 || yy >= Y) continue;
        const bool row_other = (dz && lz == CCL_TZ - 1) || (dy == 1 && ly == CCL_TY - 1) || (dy == -1 && ly == 0);
        const size_t row = ((size_t)zz * Y + yy) * X;
        const bool m0 = x > 0 && mask[row + x - 1], m1 = mask[row + x] != 0, m2 = x + 1 < X && mask[row + x + 1];
        // (a link inside the tile was made in LDS; x - 1 / x + 1 of the other row hang on its x voxel through that row's own links)
        if (row_other) {
            // the whole row lies in another tile.  As in k_ccl_merge: a foreground left neighbour (same face, so it runs this
            // code too -- in this tile or the one to the left) has linked itself to x - 2, x - 1, x of that row already
            if (!left) {
                if (m1) {
                    uf_union(L, (int)i, (int)(row + x));
                } else {
                    if (m0) uf_union(L, (int)i, (int)(row + x - 1));
                    if (m2) uf_union(L, (int)i, (int)(row + x + 1));
                }
            } else if (m2 && !m1) {
                uf_union(L, (int)i, (int)(row + x + 1));
            }
        } else if (!m1) {
            if (m0 && lx == 0) uf_union(L, (int)i, (int)(row + x - 1));
            if (m2 && lx == CCL_TX - 1) uf_union(L, (int)i, (int)(row + x + 1));
        }
    }
}

// The face voxels are enumerated directly (23 % of the volume; the first version launched over every voxel and returned for the
// rest: 1.9 of the 5.5 ms of a 512^3 mask).  mode 0: whole rows of the planes lz = TZ-1 (grid: x blocks, Y, planes);
// mode 1: the rows ly = 0 and ly = TY-1 of the other planes (grid: x blocks, 2 rows per y tile, Z);
// mode 2: the x-face voxels lx = 0 / TX-1 of the remaining rows (thread <-> (x tile, face, y), grid: blocks, 1, Z).
__global__ __launch_bounds__(256) void k_ccl_border(const unsigned char* __restrict__ mask, int Z, int Y, int X, int* L, int mode) {
    if (mode == 0) {
        const int x = (int)blockIdx.x * 256 + (int)threadIdx.x, y = (int)blockIdx.y, z = (int)blockIdx.z * CCL_TZ + CCL_TZ - 1;
        if (x >= X || z >= Z) return;
        ccl_border_voxel(mask, Z, Y, X, L, x, y, z);
    } else if (mode == 1) {
        const int x = (int)blockIdx.x * 256 + (int)threadIdx.x, z = (int)blockIdx.z;
        const int y = ((int)blockIdx.y >> 1) * CCL_TY + (((int)blockIdx.y & 1) ? CCL_TY - 1 : 0);
        if (x >= X || y >= Y || (z % CCL_TZ) == CCL_TZ - 1) return;
        ccl_border_voxel(mask, Z, Y, X, L, x, y, z);
    } else {
        const int tiles_x = (X + CCL_TX - 1) / CCL_TX;
        const int t = (int)blockIdx.x * 256 + (int)threadIdx.x, z = (int)blockIdx.z;
        const int f = t % (2 * tiles_x), y = t / (2 * tiles_x);
        const int x = (f >> 1) * CCL_TX + ((f & 1) ? CCL_TX - 1 : 0);
        if (y >= Y || x >= X) return;
        const int ly = y % CCL_TY;
        if ((z % CCL_TZ) == CCL_TZ - 1 || ly == 0 || ly == CCL_TY - 1) return;   // rows of modes 0 / 1
        ccl_border_voxel(mask, Z, Y, X, L, x, y, z);
    }
}

#define CCL_VPT 8
__global__ __launch_bounds__(256) void k_ccl_compress(size_t n, int* L, unsigned int* sizes, int* n_comp) {
    // Each thread resolves CCL_VPT voxels (256 apart, so the loads stay coalesced) and run-length merges their roots; the wave
    // then adds each distinct root's count with ONE atomic.  With one voxel per thread a single giant component (the inverted
    // body mask: 90 % of 134 M voxels) meant 2 M atomics on the same address, which alone took ~20 ms.
    const size_t base = (size_t)blockIdx.x * 256 * CCL_VPT + threadIdx.x;
    int roots[CCL_VPT];
    unsigned int cnts[CCL_VPT];
    int np = 0, ncomp = 0;
#pragma unroll
    for (int k = 0; k < CCL_VPT; ++k) {
        roots[k] = -1;
        cnts[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < CCL_VPT; ++k) {
        const size_t i = base + (size_t)k * 256;
        if (i < n && L[i] >= 0) {
            int root = (int)i;
            int p = L[root];
            while (p != root) {
                root = p;
                p = L[root];
            }
            L[i] = root;  // values only ever move towards the root: a concurrent walker through i just gets there sooner
            if (root == (int)i) ++ncomp;
            bool merged = false;
#pragma unroll
            for (int q = 0; q < CCL_VPT; ++q)
                if (!merged && q < np && roots[q] == root) {
                    ++cnts[q];
                    merged = true;
                }
            if (!merged) {
#pragma unroll
                for (int q = 0; q < CCL_VPT; ++q)
                    if (q == np) {
                        roots[q] = root;
                        cnts[q] = 1;
                    }
                ++np;
            }
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < CCL_VPT; ++q) {
        bool mine = q < np;
        const int root = roots[q];
        const unsigned int cnt = cnts[q];
        unsigned long long active = __ballot(mine);
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const int r = __shfl(root, leader);
            const bool match = mine && root == r;
            const unsigned long long same = __ballot(match);
            unsigned int v = match ? cnt : 0u;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            if (lane == leader) atomicAdd(&sizes[r], v);
            if (match) mine = false;
            active &= ~same;
        }
    }
    // number of components: wave sum, one atomic per wave
    unsigned int nc = (unsigned int)ncomp;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nc += __shfl_xor(nc, m);
    if (lane == 0 && nc) atomicAdd(n_comp, (int)nc);
}

// note on k_ccl_compress: path compression writes L[i] = root while other threads may still walk through i.
// A walker that reads the new value simply jumps to the root sooner: values only ever move towards the root.

extern "C" int boa_ccl26(boa_ctx* c, const uint8_t* dev_mask, int Z, int Y, int X, int32_t* dev_roots,
                         uint32_t* dev_sizes, int* host_n_components) {
    BOA_REQUIRE(c && dev_mask && dev_roots && dev_sizes && Z > 0 && Y > 0 && X > 0, "boa_ccl26: bad argument");
    const size_t n = (size_t)Z * Y * X;
    BOA_REQUIRE(n < (1ull << 31), "boa_ccl26: volume too large for int32 indices");
    // the component counter: a pooled 4-byte block; without host_n_components nothing is copied back and the call does not
    // synchronise (the BCA post-processing chains 16 of these per volume)
    int* d_count = nullptr;
    BOA_TRY(boa_malloc(c, sizeof(int), (void**)&d_count));
    static const bool flat = getenv("BOA_CCL_FLAT") != nullptr;  // the one-level version (A/B switch)
    {   // (an early return must hand the pooled counter back)
        hipError_t e0 = hipMemsetAsync(d_count, 0, sizeof(int), c->stream);
        if (e0 == hipSuccess && flat) e0 = hipMemsetAsync(dev_sizes, 0, n * sizeof(uint32_t), c->stream);
        if (e0 != hipSuccess) {
            boa_free(c, d_count);
            BOA_HIP_TRY(e0);
        }
    }
    unsigned grid = (unsigned)((n + 255) / 256);
    KernelTimer t(c, BOA_K_MORPH, 0, (double)n * 14.0);
    if (flat) {
        hipLaunchKernelGGL(k_ccl_init, dim3(grid), dim3(256), 0, c->stream, dev_mask, n, dev_roots);
        hipLaunchKernelGGL(k_ccl_merge, dim3(grid), dim3(256), 0, c->stream, dev_mask, Z, Y, X, dev_roots);
        hipLaunchKernelGGL(k_ccl_compress, dim3((unsigned)((n + 256 * CCL_VPT - 1) / (256 * CCL_VPT))), dim3(256), 0, c->stream, n, dev_roots,
                           dev_sizes, d_count);
    } else {
        const int tx = (X + CCL_TX - 1) / CCL_TX, ty = (Y + CCL_TY - 1) / CCL_TY, tz = (Z + CCL_TZ - 1) / CCL_TZ;
        hipLaunchKernelGGL(k_ccl_local, dim3((unsigned)((size_t)tx * ty * tz)), dim3(256), 0, c->stream, dev_mask, Z, Y, X, tx, ty, dev_roots,
                           dev_sizes);
        hipLaunchKernelGGL(k_ccl_border, dim3((unsigned)((X + 255) / 256), (unsigned)Y, (unsigned)tz), dim3(256), 0, c->stream, dev_mask, Z, Y, X,
                           dev_roots, 0);
        hipLaunchKernelGGL(k_ccl_border, dim3((unsigned)((X + 255) / 256), (unsigned)(2 * ty), (unsigned)Z), dim3(256), 0, c->stream, dev_mask, Z,
                           Y, X, dev_roots, 1);
        hipLaunchKernelGGL(k_ccl_border, dim3((unsigned)(((size_t)2 * tx * Y + 255) / 256), 1, (unsigned)Z), dim3(256), 0, c->stream, dev_mask, Z, Y,
                           X, dev_roots, 2);
        hipLaunchKernelGGL(k_ccl_resolve, dim3(grid), dim3(256), 0, c->stream, n, dev_roots, dev_sizes, d_count);
    }
    t.stop();
    hipError_t e = hipGetLastError();
    if (host_n_components && e == hipSuccess) {
        int cnt = 0;
        e = hipMemcpyAsync(&cnt, d_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        *host_n_components = cnt;
    }
    boa_free(c, d_count);  // (stream-ordered: the block is reused only by work queued after the kernels above)
    BOA_HIP_TRY(e);
    return BOA_OK;
}

__global__ __launch_bounds__(256) void k_ccl_best(const unsigned int* __restrict__ sizes, size_t n,
                                                  unsigned long long* best) {
    // grid-stride, one atomic per wave of a few thousand (one per 64 voxels was 2 M atomics on one word: 1.8 ms per 512^3 volume)
    unsigned long long key = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned int sz = sizes[i];
        if (sz) {
            const unsigned long long k = ((unsigned long long)sz << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            key = k > key ? k : key;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = __shfl_xor(key, m);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(best, key);
}

__global__ __launch_bounds__(256) void k_ccl_apply_largest(const int* __restrict__ roots, size_t n,
                                                           const unsigned long long* best, unsigned char* seg, int fill) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = *best;
    if (!b) return;
    const int best_root = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
    const int r = roots[i];
    if (r >= 0 && r != best_root) seg[i] = (unsigned char)fill;
}

extern "C" int boa_ccl_filter_largest(boa_ctx* c, const int32_t* dev_roots, const uint32_t* dev_sizes, size_t n,
                                      uint8_t* dev_seg, int fill_value) {
    BOA_REQUIRE(c && dev_roots && dev_sizes && dev_seg, "boa_ccl_filter_largest: NULL argument");
    if (n == 0) return BOA_OK;
    unsigned long long* d_best = nullptr;
    BOA_TRY(boa_malloc(c, sizeof(unsigned long long), (void**)&d_best));   // (pooled: no synchronisation around the two kernels)
    {
        const hipError_t e0 = hipMemsetAsync(d_best, 0, sizeof(unsigned long long), c->stream);
        if (e0 != hipSuccess) {   // (an early return must hand the pooled block back)
            boa_free(c, d_best);
            BOA_HIP_TRY(e0);
        }
    }
    unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_ccl_best, dim3(std::min<unsigned>(grid, (unsigned)c->cu_count * 8)), dim3(256), 0, c->stream, dev_sizes, n, d_best);
    hipLaunchKernelGGL(k_ccl_apply_largest, dim3(grid), dim3(256), 0, c->stream, dev_roots, n, d_best, dev_seg,
                       fill_value);
    hipError_t e = hipGetLastError();
    boa_free(c, d_best);
    BOA_HIP_TRY(e);
    return BOA_OK;
}

__global__ __launch_bounds__(256) void k_ccl_remove_small(const int* __restrict__ roots, const unsigned int* __restrict__ sizes,
                                                          size_t n, unsigned int max_size, unsigned char* mask) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = roots[i];
    if (r >= 0 && sizes[r] <= max_size) mask[i] = 0;
}

extern "C" int boa_ccl_remove_small(boa_ctx* c, const int32_t* dev_roots, const uint32_t* dev_sizes, size_t n,
                                    uint32_t max_size, uint8_t* dev_mask_inout) {
    BOA_REQUIRE(c && dev_roots && dev_sizes && dev_mask_inout, "boa_ccl_remove_small: NULL argument");
    if (n == 0) return BOA_OK;
    unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_ccl_remove_small, dim3(grid), dim3(256), 0, c->stream, dev_roots, dev_sizes, n, max_size,
                       dev_mask_inout);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}


// ------------------------------------------------------------------------------------------------------
// z-slab sharded connected components (SURVEY 8e "aggregation stages"): each rank labels its slab with boa_ccl26; the
// components that touch a slab interface are merged on the host over the exchanged boundary planes (boa_hip/agg_shard.py).
// These helpers move the small per-component tables between the device and the host.
__global__ __launch_bounds__(256) void k_ccl_list(const unsigned int* __restrict__ sizes, size_t n, int max_out, int* __restrict__ roots_out,
                                                  unsigned int* __restrict__ sizes_out, int* __restrict__ count) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned s = sizes[i];
    if (s == 0) return;
    const int k = atomicAdd(count, 1);
    if (k < max_out) {
        roots_out[k] = (int)i;
        sizes_out[k] = s;
    }
}

extern "C" int boa_ccl_list_components(boa_ctx* c, const uint32_t* dev_sizes, size_t n, int max_out, int32_t* host_roots,
                                       uint32_t* host_sizes, int* host_count) {
    BOA_REQUIRE(c && dev_sizes && host_roots && host_sizes && host_count && max_out >= 0, "boa_ccl_list_components: bad argument");
    int* d_cnt = nullptr;
    int* d_roots = nullptr;
    unsigned* d_sz = nullptr;
    BOA_TRY(boa_malloc(c, sizeof(int), (void**)&d_cnt));
    int rc = boa_malloc(c, (size_t)std::max(max_out, 1) * 4, (void**)&d_roots);
    if (!rc) rc = boa_malloc(c, (size_t)std::max(max_out, 1) * 4, (void**)&d_sz);
    if (!rc) {
        hipMemsetAsync(d_cnt, 0, sizeof(int), c->stream);
        if (n) hipLaunchKernelGGL(k_ccl_list, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dev_sizes, n, max_out, d_roots, d_sz, d_cnt);
        c->prof_break = true;
        hipError_t e = hipMemcpyAsync(host_count, d_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        const int m = std::min(*host_count, max_out);
        if (e == hipSuccess && m > 0) e = hipMemcpy(host_roots, d_roots, (size_t)m * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && m > 0) e = hipMemcpy(host_sizes, d_sz, (size_t)m * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            boa_set_error("boa_ccl_list_components: %s", hipGetErrorString(e));
            rc = BOA_EHIP;
        }
    }
    boa_free(c, d_cnt);
    if (d_roots) boa_free(c, d_roots);
    if (d_sz) boa_free(c, d_sz);
    return rc;
}

__global__ __launch_bounds__(256) void k_scatter_u32(const int* __restrict__ idx, const unsigned int* __restrict__ val, int m,
                                                     unsigned int* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) dst[idx[i]] = val[i];
}

extern "C" int boa_scatter_u32(boa_ctx* c, uint32_t* dev_dst, const int32_t* host_idx, const uint32_t* host_val, int m) {
    BOA_REQUIRE(c && dev_dst && (m == 0 || (host_idx && host_val)) && m >= 0, "boa_scatter_u32: bad argument");
    if (m == 0) return BOA_OK;
    int* d_i = nullptr;
    unsigned* d_v = nullptr;
    BOA_TRY(boa_malloc(c, (size_t)m * 4, (void**)&d_i));
    int rc = boa_malloc(c, (size_t)m * 4, (void**)&d_v);
    if (!rc) {
        c->prof_break = true;
        // (stream-ordered copies: d_i / d_v may be recycled blocks whose previous user still has work queued on the stream)
        hipError_t e = hipMemcpyAsync(d_i, host_idx, (size_t)m * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_v, host_val, (size_t)m * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_scatter_u32, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, d_i, d_v, m, dev_dst);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the host arrays are borrowed for the duration of the call
        if (e != hipSuccess) {
            boa_set_error("boa_scatter_u32: %s", hipGetErrorString(e));
            rc = BOA_EHIP;
        }
    }
    boa_free(c, d_i);
    if (d_v) boa_free(c, d_v);
    return rc;
}

__global__ __launch_bounds__(256) void k_ccl_fill_unmarked(const int* __restrict__ roots, const unsigned int* __restrict__ sizes, size_t n,
                                                           unsigned int mark, unsigned char* __restrict__ seg, int fill) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = roots[i];
    if (r >= 0 && sizes[r] != mark) seg[i] = (unsigned char)fill;
}

extern "C" int boa_ccl_fill_unmarked(boa_ctx* c, const int32_t* dev_roots, const uint32_t* dev_sizes, size_t n, uint32_t mark,
                                     uint8_t* dev_seg, int fill_value) {
    BOA_REQUIRE(c && dev_roots && dev_sizes && dev_seg, "boa_ccl_fill_unmarked: NULL argument");
    if (n == 0) return BOA_OK;
    KernelTimer t(c, BOA_K_MORPH, 0, (double)n * 6.0);
    hipLaunchKernelGGL(k_ccl_fill_unmarked, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dev_roots, dev_sizes, n, mark,
                       dev_seg, fill_value);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
