// Fused head of the sliding window: 1x1x1 conv + Gaussian weighting + fp16 accumulation over ALL covering tiles + normalisation +
// fold sum / mean + argmax + label remap in ONE pass over the volume ("gather" form of NN/inference/predict_from_raw_data.py:
// 560-631 + export_prediction.py:14-71 for identity resampling).
//
// The reference adds tile after tile into fp16 accumulators (`predicted_logits[sl] += prediction * gaussian`,
// `n_predictions[sl] += gaussian`, :611-614); the scatter form (k_head_mfma, one launch per tile in canonical order) therefore
// reads and writes the C + 1 accumulator planes once per tile that covers a voxel -- 104 of its 170 bytes per voxel and tile --
// and k_finalize_labels reads them once more.  Here the conv stack leaves the last decoder activation of EVERY tile of the
// volume in a stash (16.8 GB for the 125 tiles of a 512^3 part model: HBM has room), and each wave takes 32 consecutive z voxels
// of the volume, walks the tiles that cover them IN ASCENDING TILE INDEX (the reference's order, so every fp16 rounding happens
// in the same sequence), computes the tile's logits for its voxels with the same two MFMAs as k_head_mfma and keeps the C + 1
// running sums in registers: no accumulator planes exist at all, the only traffic is the stash read (66 B per voxel and covering
// tile) and one label byte per voxel.  Arithmetic per (voxel, tile), identical to k_head_mfma / k_finalize_labels:
//   x = lrelu(fma(act, scale, shift)) in packed fp16; logit = MFMA(w, x) + bias (fp32); pr = logit * g (fp32);
//   acc = half(float(acc) + pr); n = half(float(n) + g);   then  q = half(float(acc) / float(n)), fold sum / mean in half,
//   numpy argmax (first maximum, first NaN wins), lut / merge, crop.
// Several folds: one pass per fold, the normalised logits of a fold are added into a [C][voxels] fp16 buffer (`fold`), the last
// pass divides by the number of folds and takes the argmax (predict_from_raw_data.py:483-500).
#include "conv.h"

#ifndef GH_WAVES
#define GH_WAVES 6   // (4: 8.5 ms, 5: 7.5, 6: 7.2, 8: 7.2 per 512^3 part model -- latency-bound: occupancy pays more than the ~20 spilled registers cost)
#endif
typedef _Float16 gh2_t __attribute__((ext_vector_type(2)));

struct GatherArgs {
    const __half* act;            // stash [tile][2 planes][pv][16 halves]: raw output of the last decoder conv
    const unsigned* ssp;          // [tile][2 k-halves][16 words]: packed fp16 (scale, shift) of its InstanceNorm, k_head_mfma's layout
    const float* w;               // [C][32] head weights (this fold)
    const float* bias;            // [C]
    const unsigned short* gauss;  // [pv] fp16 or nullptr (weight 1)
    int C, P0, P1, P2, V0, V1, V2, n0, n1, n2;
    const int* tab;               // device: [tile origins per axis: n0 + n1 + n2][cover of x: V0][cover of y: V1][cover of the z runs]
                                  // cover word = first covering tile | count << 8 (host-built: the walk is scalar table look-ups)
    unsigned short* fold;         // [C][V0 V1 V2] fp16 (several folds) or nullptr
    int fold_mode;                // 0 single fold; 1 first of several (store); 2 middle (add); 3 last (add, / n_folds, argmax);
                                  // 4 raw partial sums (tile sharding): `fold` = the fp16 accumulator planes [C][V], `raw_n` = the weight plane [V];
                                  //   the running sums of planes [x_lo, x_hi) are WRITTEN there (started from the planes' contents when raw_init)
    int n_folds;
    int x_lo, x_hi;               // axis-0 planes this launch covers (the whole volume except in mode 4)
    unsigned short* raw_n;
    int raw_init;
    unsigned char* labels;
    int merge, crop, o0, o1, o2, c0, c1, c2;
    int* inf_flag;
    float slope;
    int ss_in_lds;                // the packed (scale, shift) table of all tiles fits the dynamic LDS allocation
    // split-precision mode (X3): act = fp32 octet planes [tile][4 planes][pv][8 floats], ssp = fp32 (scale, shift) [tile][32][2],
    // head weights split in the kernel as hi / lo parts of w * wscale (winv = 1 / wscale)
    float wscale, winv;
    unsigned char lut[256];
};

// fp16 += fp32 on a packed pair, as ONE pair: two conversions up (the second an SDWA form), one v_pk_add_f32, one v_cvt_pk_f16_f32 (RTNE) --
// the same fp32 additions and roundings as f2us(us2f(lo) + x) | f2us(us2f(hi) + y) << 16.  Written with shifts and masks hipcc paired the
// LOW halves of two neighbouring accumulator words (and the high halves) for its packed adds and paid three v_mov, a v_and, a v_lshl and two
// v_or_sdwa per four values to get the products and the results back into place: 19 instead of 12 instructions per four classes.
__device__ __forceinline__ unsigned gh_acc_add(unsigned acc, float x, float y) {
    typedef float ghp2_t __attribute__((ext_vector_type(2)));
    union {
        unsigned u;
        gh2_t v;
    } a, r;
    a.u = acc;
    const ghp2_t s = ghp2_t{(float)a.v[0], (float)a.v[1]} + ghp2_t{x, y};
    r.v = __builtin_convertvector(s, gh2_t);
    return r.u;
}

// fp32 (scale, shift) [tile][32][2] -> the packed layout the head reads: [tile][kh][step 0: sc x4, sh x4 | step 1: sc x4, sh x4]
__global__ void k_pack_head_ss(const float* __restrict__ ss, unsigned* __restrict__ out, int n_tiles) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles * 2) return;
    const int tile = t >> 1, k = t & 1;
    const float* s = ss + (size_t)tile * 64;
    unsigned* o = out + (size_t)t * 16;
    union {
        unsigned u;
        gh2_t v;
    } cv;
    for (int i = 0; i < 4; ++i) {
        const int c0 = 8 * k + 2 * i, c1 = 16 + 8 * k + 2 * i;
        cv.v = gh2_t{(_Float16)s[2 * c0], (_Float16)s[2 * c0 + 2]};
        o[i] = cv.u;
        cv.v = gh2_t{(_Float16)s[2 * c0 + 1], (_Float16)s[2 * c0 + 3]};
        o[4 + i] = cv.u;
        cv.v = gh2_t{(_Float16)s[2 * c1], (_Float16)s[2 * c1 + 2]};
        o[8 + i] = cv.u;
        cv.v = gh2_t{(_Float16)s[2 * c1 + 1], (_Float16)s[2 * c1 + 3]};
        o[12 + i] = cv.u;
    }
}

// MULTI: several folds (fold buffer read-modify-write, general epilogue); the single-fold instantiation carries none of that code
// (its 16 per-class plane pointers were the registers that spilled at six waves per SIMD)
// X3: the split-precision mode's head (precision 2): the stash holds fp32; a lane loads channels 8 c + 4 kh .. + 3 of its voxel for the
// four octets c, normalises in fp32, splits into hi / lo fp16 parts, and two v_permlane32_swap per octet hand the k-half-0 lane
// the hi parts of all 8 channels and the k-half-1 lane the lo parts: the B operand of [Wh | Wh] x [Xh ; Xl] + [Wl | Wl] x [Xh ; Xl]
// (8 MFMAs per covering tile instead of 2).  Everything after the logits is the same code.
// PF (fp16 mode): the stash records and the Gaussian of covering tile p + 1 are fetched by LDS-DMA (global_load_lds: no VGPRs, no
// ds_write) into a per-wave LDS slot while tile p is consumed, so every wave keeps loads in flight all the time -- the kernel was
// bound by HBM round trips of 1 KiB pieces at the occupancy its registers allow; holding the next pair in registers cost the
// occupancy it was meant to replace.
template <bool GAUSS, bool SSLDS, bool MULTI, bool X3, bool PF = false>
__device__ __forceinline__ void gather_head_body(const GatherArgs& p) {
    const int fold_mode = MULTI ? p.fold_mode : 0;
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int* __restrict__ sx = p.tab;
    const int* __restrict__ sy = p.tab + p.n0;
    const int* __restrict__ sz = p.tab + p.n0 + p.n1;
    const int* __restrict__ cvx = sz + p.n2;
    const int* __restrict__ cvy = cvx + p.V0;
    const int* __restrict__ cvz = cvy + p.V1;
    f16x8 a0, a1;
    f16x8 xah[X3 ? 4 : 1], xal[X3 ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a0[i] = l31 < p.C ? (_Float16)p.w[l31 * 32 + 8 * kh + i] : (_Float16)0.f;
        a1[i] = l31 < p.C ? (_Float16)p.w[l31 * 32 + 16 + 8 * kh + i] : (_Float16)0.f;
    }
    if constexpr (X3) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float wv = l31 < p.C ? p.w[l31 * 32 + 8 * c + i] * p.wscale : 0.f;
                const _Float16 h = (_Float16)wv;
                xah[c][i] = h;
                xal[c][i] = (_Float16)(wv - (float)h);
            }
    }
    // this lane's 16 biases (D-fragment rows 8 gq + 4 kh + e) live in LDS, re-read per tile pair: the kernel waits on HBM round
    // trips, so VGPRs (waves in flight) matter more than four ds_read_b128 per pair
    __shared__ __attribute__((aligned(16))) float s_bz[2][16];
    if (threadIdx.x < 32) {
        const int k = threadIdx.x >> 4, i = threadIdx.x & 15;
        const int c = 8 * (i >> 2) + 4 * k + (i & 3);
        s_bz[k][i] = c < p.C ? p.bias[c] : 0.f;
    }
    __syncthreads();
    const gh2_t sl = gh2_t{(_Float16)p.slope, (_Float16)p.slope};
    auto xform = [&](uint4 raw, uint4 scw, uint4 shw) {
        union {
            uint4 u;
            gh2_t v[4];
            f16x8 f;
        } x, sc, sh;
        x.u = raw;
        sc.u = scw;
        sh.u = shw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const gh2_t y = __builtin_elementwise_fma(x.v[i], sc.v[i], sh.v[i]);
            x.v[i] = __builtin_elementwise_max(y, y * sl);
        }
        return x.f;
    };
    const size_t pv = (size_t)p.P0 * p.P1 * p.P2;
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
    const int mpr = (p.V2 + 31) / 32;
    // (32-bit run indices: 64-bit divisions cost ~100 instructions each; the host checks that the run count fits)
    const unsigned n_mt = (unsigned)p.x_hi * (unsigned)p.V1 * (unsigned)mpr, mt_lo = (unsigned)p.x_lo * (unsigned)p.V1 * (unsigned)mpr;
    const unsigned gw = __builtin_amdgcn_readfirstlane((blockIdx.x * 256u + threadIdx.x) >> 6) + mt_lo, nw = gridDim.x * 4u;   // (wave-uniform: scalar tile walk)
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // every tile's packed (scale, shift) table in LDS (128 B per tile) when it fits (SSLDS): four ds_read_b128 per (run, tile)
    // pair instead of four L2 round trips
    extern __shared__ __attribute__((aligned(16))) unsigned s_ssp[];
    if (SSLDS) {
        const int n_tiles = p.n0 * p.n1 * p.n2;
        for (int i = threadIdx.x; i < n_tiles * (X3 ? 16 : 8); i += 256) ((uint4*)s_ssp)[i] = ((const uint4*)p.ssp)[i];
        __syncthreads();
    }
    typedef __attribute__((address_space(1))) const unsigned char* gptr_t;
    typedef unsigned gu4_t __attribute__((ext_vector_type(4)));
    auto gload4 = [](gptr_t b, unsigned off) {
        const gu4_t v = *(const __attribute__((address_space(1))) gu4_t*)(b + off);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    bool any_inf = false;
    for (unsigned mt = gw; mt < n_mt; mt += nw) {
        const unsigned row = mt / (unsigned)mpr;
        const int zb = (int)(mt - row * (unsigned)mpr) * 32;
        const int x = (int)(row / (unsigned)p.V1), y = (int)(row - (unsigned)(row / (unsigned)p.V1) * (unsigned)p.V1);
        const int z = zb + l31;
        const bool zvalid = z < p.V2;
        // the C running sums of this lane's 16 classes as 8 packed fp16 pairs (what the reference's fp16 accumulator planes hold)
        unsigned acch[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acch[i] = 0u;
        float nacc = 0.f;
        if (MULTI && fold_mode == 4 && p.raw_init) {   // tile sharding: continue the lower rank's partial sums (same fp16 += sequence)
            // (a wave-uniform branch with clamped per-lane addresses: behind a per-lane branch hipcc's backend rejected the scalar-base
            //  pins of the tile loop, "illegal VGPR to SGPR copy")
            const size_t vi0 = ((size_t)x * p.V1 + y) * p.V2 + (zvalid ? z : 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 8 * (i >> 2) + 4 * kh + (i & 3);
                const unsigned v = p.fold[(size_t)(c < p.C ? c : 0) * vv + vi0];
                acch[i >> 1] |= (c < p.C && zvalid ? v : 0u) << (16 * (i & 1));
            }
            nacc = zvalid ? us2f(p.raw_n[vi0]) : 0.f;
        }
        // covering tiles in ascending tile index: x outermost, z innermost (predict_from_raw_data.py:506-538)
        // (wave-uniform table look-ups: scalar loads; the first version scanned the origins in LDS and divided per pair -- the walk
        //  alone took 5 of the kernel's 9.6 ms per 512^3 part model)
        // (PF: the tables are read through the CONSTANT address space -- the DMA / s_waitcnt asm statements clobber "memory", after which
        //  hipcc no longer issues scalar loads for ordinary global pointers: it fell back to vector loads + v_readfirstlane whose
        //  vmcnt(0) waits also waited for the prefetch that had just been issued)
        typedef const int __attribute__((address_space(4))) * ctab_t;
        const ctab_t csx = (ctab_t)(unsigned long long)sx, csy = (ctab_t)(unsigned long long)sy, csz = (ctab_t)(unsigned long long)sz;
        const int wx = PF ? ((ctab_t)(unsigned long long)cvx)[x] : cvx[x], wy = PF ? ((ctab_t)(unsigned long long)cvy)[y] : cvy[y],
                  wz = PF ? ((ctab_t)(unsigned long long)cvz)[zb >> 5] : cvz[zb >> 5];
        const int fx = wx & 255, cx = wx >> 8, fy = wy & 255, cy = wy >> 8, fz = wz & 255, cz = wz >> 8;
        if constexpr (PF && !X3) {
            __shared__ __attribute__((aligned(16))) unsigned char s_pf[4][2][2560];   // [wave][buffer][plane 0 | plane 1 | gauss words]
            unsigned char* pfb = &s_pf[threadIdx.x >> 6][0][0];
            const unsigned pf_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pfb);
            const int np = cx * cy * cz;
            int jx = fx, jy = fy, jz = fz;   // the pair being issued
            auto dma16 = [](gptr_t base, unsigned voff, unsigned lds_off) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_off));
            };
            auto dma4 = [](gptr_t base, unsigned voff, unsigned lds_off) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_off));
            };
            auto issue = [&](int buf, unsigned& tile_o, int& in_o, unsigned& gsel_o) {
                const int dx = x - csx[jx], dy = y - csy[jy], tz0 = csz[jz];
                const bool in = zvalid && z >= tz0 && z < tz0 + p.P2;
                const unsigned tile = (unsigned)((jx * p.n1 + jy) * p.n2 + jz);
                const unsigned tvl = (unsigned)(dx * p.P1 + dy) * (unsigned)p.P2 + (unsigned)(in ? z - tz0 : 0);
                gptr_t ab = (gptr_t)p.act + (size_t)tile * 2 * pv * 32;
                asm volatile("" : "+s"(ab));
                const unsigned lo = pf_lds + (unsigned)buf * 2560u;
                const unsigned o0 = tvl * 32u + (unsigned)kh * 16u;
                dma16(ab, o0, lo);
                gptr_t ab1 = ab + pv * 32;
                asm volatile("" : "+s"(ab1));
                dma16(ab1, o0, lo + 1024u);
                if (GAUSS) {
                    gptr_t gb = (gptr_t)p.gauss;
                    asm volatile("" : "+s"(gb));
                    dma4(gb, (tvl * 2u) & ~3u, lo + 2048u);       // the aligned word that holds this voxel's fp16 weight
                }
                tile_o = tile;
                in_o = in ? 1 : 0;
                gsel_o = tvl & 1u;
                if (++jz == fz + cz) {
                    jz = fz;
                    if (++jy == fy + cy) {
                        jy = fy;
                        ++jx;
                    }
                }
            };
            unsigned tile_n = 0, gsel_n = 0;
            int in_n = 0;
            if (np > 0) issue(0, tile_n, in_n, gsel_n);
            for (int pi = 0; pi < np; ++pi) {
                const unsigned tile = tile_n, gsel = gsel_n;
                const bool in = in_n != 0;
                const bool more = pi + 1 < np;
                // The DMAs issued next overwrite buffer (pi + 1) & 1 = the buffer pair pi - 1's ordinary LDS reads came from.  The DMA asm
                // statements name no memory operand, so this compiler-only barrier is what keeps those reads (program order: previous
                // iteration) in front of them; the hardware side is in order by itself (the reads' values were consumed by that
                // iteration's MFMAs).  The walk tables live in the constant address space and are not pinned by it.
                asm volatile("" ::: "memory");
                if (more) issue((pi + 1) & 1, tile_n, in_n, gsel_n);
                // the DMAs of pair pi are older than the (2 + GAUSS) of pair pi + 1: in-order completion
                if (more)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + (GAUSS ? 1 : 0)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned char* cb = pfb + (pi & 1) * 2560;
                const uint4 r0 = *(const uint4*)(cb + lane * 16), r1 = *(const uint4*)(cb + 1024 + lane * 16);
                float g = 1.0f;
                if (GAUSS) {
                    const unsigned gw2 = *(const unsigned*)(cb + 2048 + lane * 4);
                    g = us2f((unsigned short)(gsel ? (gw2 >> 16) : (gw2 & 0xFFFFu)));
                }
                uint4 sc0, sh0, sc1, sh1;
                if (SSLDS) {
                    const uint4* sw = (const uint4*)(s_ssp + (tile * 2 + kh) * 16);
                    sc0 = sw[0]; sh0 = sw[1]; sc1 = sw[2]; sh1 = sw[3];
                } else {
                    const uint4* sw = (const uint4*)(p.ssp + ((size_t)tile * 2 + kh) * 16);
                    sc0 = sw[0]; sh0 = sw[1]; sc1 = sw[2]; sh1 = sw[3];
                }
                f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xform(r0, sc0, sh0), zero, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xform(r1, sc1, sh1), d, 0, 0, 0);
                if (in) {
                    typedef float gf2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const gf2_t sum = gf2_t{d[2 * i], d[2 * i + 1]} + gf2_t{s_bz[kh][2 * i], s_bz[kh][2 * i + 1]};
                        const gf2_t pr = GAUSS ? sum * gf2_t{g, g} : sum;
                        acch[i] = gh_acc_add(acch[i], pr.x, pr.y);
                    }
                    nacc = us2f(f2us(nacc + g));
                }
            }
        } else
        for (int ix = fx; ix < fx + cx; ++ix) {
            const int dx = x - sx[ix];
            for (int iy = fy; iy < fy + cy; ++iy) {
                const int dy = y - sy[iy];
                const unsigned rowoff = (unsigned)(dx * p.P1 + dy) * (unsigned)p.P2;   // tile-relative voxel index of (dx, dy, 0)
                const unsigned tile0 = (unsigned)((ix * p.n1 + iy) * p.n2);
                for (int iz = fz; iz < fz + cz; ++iz) {
                    const int tz0 = sz[iz];
                    const bool in = zvalid && z >= tz0 && z < tz0 + p.P2;
                    const unsigned tile = tile0 + (unsigned)iz;
                    const unsigned tvl = rowoff + (unsigned)(in ? z - tz0 : 0);
                    f32x16 d;
                    float g = 1.0f;
                    if constexpr (X3) {
                        gptr_t ab = (gptr_t)p.act + (size_t)tile * 4 * pv * 32;
                        asm volatile("" : "+s"(ab));
                        uint4 raw[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            unsigned oc = tvl * 32u + (unsigned)kh * 16u + (unsigned)c * ((unsigned)pv * 32u);
                            asm volatile("" : "+v"(oc));
                            raw[c] = gload4(ab, oc);
                        }
                        if (GAUSS) {
                            gptr_t gb = (gptr_t)p.gauss;
                            asm volatile("" : "+s"(gb));
                            unsigned og = tvl * 2u;
                            asm volatile("" : "+v"(og));
                            g = us2f(*(const __attribute__((address_space(1))) unsigned short*)(gb + og));
                        }
                        d = zero;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            // (scale, shift) of channels 8 c + 4 kh .. + 3: 8 floats
                            uint4 s01, s23;
                            if (SSLDS) {
                                const uint4* sw = (const uint4*)(s_ssp + tile * 64 + (8 * c + 4 * kh) * 2);
                                s01 = sw[0]; s23 = sw[1];
                            } else {
                                const uint4* sw = (const uint4*)(p.ssp + (size_t)tile * 64 + (8 * c + 4 * kh) * 2);
                                s01 = sw[0]; s23 = sw[1];
                            }
                            const float xs[4] = {__uint_as_float(raw[c].x), __uint_as_float(raw[c].y), __uint_as_float(raw[c].z), __uint_as_float(raw[c].w)};
                            const float sc[4] = {__uint_as_float(s01.x), __uint_as_float(s01.z), __uint_as_float(s23.x), __uint_as_float(s23.z)};
                            const float sh[4] = {__uint_as_float(s01.y), __uint_as_float(s01.w), __uint_as_float(s23.y), __uint_as_float(s23.w)};
                            float y[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float f = __builtin_fmaf(xs[e], sc[e], sh[e]);
                                y[e] = f > 0.f ? f : f * p.slope;
                            }
                            union {
                                gh2_t v;
                                unsigned u;
                            } h01, h23, l01, l23;
                            typedef float ghf2_t __attribute__((ext_vector_type(2)));
                            h01.v = __builtin_convertvector(ghf2_t{y[0], y[1]}, gh2_t);
                            h23.v = __builtin_convertvector(ghf2_t{y[2], y[3]}, gh2_t);
                            l01.v = __builtin_convertvector(ghf2_t{y[0] - (float)h01.v[0], y[1] - (float)h01.v[1]}, gh2_t);
                            l23.v = __builtin_convertvector(ghf2_t{y[2] - (float)h23.v[0], y[3] - (float)h23.v[1]}, gh2_t);
                            // A = hi words, B = lo words: the swap exchanges A's upper lanes with B's lower lanes ->
                            // k-half 0: (A, B) = (own hi ch 0-3, partner's hi ch 4-7); k-half 1: (partner's lo ch 0-3, own lo ch 4-7)
                            const auto s0 = __builtin_amdgcn_permlane32_swap(h01.u, l01.u, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(h23.u, l23.u, false, false);
                            union {
                                unsigned u[4];
                                f16x8 f;
                            } b;
                            b.u[0] = s0[0]; b.u[1] = s1[0]; b.u[2] = s0[1]; b.u[3] = s1[1];
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah[c], b.f, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(xal[c], b.f, d, 0, 0, 0);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) d[i] *= p.winv;
                    } else {
                    // wave-uniform 64-bit base + 32-bit lane offset (scalar-base global loads)
                    gptr_t ab = (gptr_t)p.act + (size_t)tile * 2 * pv * 32;
                    asm volatile("" : "+s"(ab));
                    unsigned o0 = tvl * 32u + (unsigned)kh * 16u;
                    asm volatile("" : "+v"(o0));
                    const uint4 r0 = gload4(ab, o0);
                    gptr_t ab1 = ab + pv * 32;
                    asm volatile("" : "+s"(ab1));
                    unsigned o1 = o0;
                    asm volatile("" : "+v"(o1));
                    const uint4 r1 = gload4(ab1, o1);
                    if (GAUSS) {
                        gptr_t gb = (gptr_t)p.gauss;
                        asm volatile("" : "+s"(gb));
                        unsigned og = tvl * 2u;
                        asm volatile("" : "+v"(og));
                        g = us2f(*(const __attribute__((address_space(1))) unsigned short*)(gb + og));
                    }
                    uint4 sc0, sh0, sc1, sh1;
                    if (SSLDS) {
                        const uint4* sw = (const uint4*)(s_ssp + (tile * 2 + kh) * 16);
                        sc0 = sw[0]; sh0 = sw[1]; sc1 = sw[2]; sh1 = sw[3];
                    } else {
                        const uint4* sw = (const uint4*)(p.ssp + ((size_t)tile * 2 + kh) * 16);
                        sc0 = sw[0]; sh0 = sw[1]; sc1 = sw[2]; sh1 = sw[3];
                    }
                    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xform(r0, sc0, sh0), zero, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xform(r1, sc1, sh1), d, 0, 0, 0);
                    }
                    if (in) {
                        // two entries per instruction (v_pk_add_f32 / v_pk_mul_f32 / v_cvt_pk_f16_f32): the same fp32 operations
                        // and RTNE roundings per entry -- the kernel is VALU-bound (~700 wave instructions per 32-voxel run)
                        typedef float gf2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const gf2_t sum = gf2_t{d[2 * i], d[2 * i + 1]} + gf2_t{s_bz[kh][2 * i], s_bz[kh][2 * i + 1]};
                            const gf2_t pr = GAUSS ? sum * gf2_t{g, g} : sum;                    // prediction *= gaussian (fp32)
                            // fp16 += fp32: fp32 add, RTNE to fp16; the sums stay packed between tiles (8 registers instead of 16: the
                            // kernel is latency-bound and wants waves, not fewer instructions).  v_fma_mix{lo,hi}_f16 with a multiplier of
                            // 1.0 would do add + conversion in one instruction but is NOT the same arithmetic in rare cases (8 of 134 M
                            // voxels of a 512^3 part model came out with another label than the scatter form: tests/test_gpu_fullsize.py)
                            acch[i] = gh_acc_add(acch[i], pr.x, pr.y);
                        }
                        nacc = us2f(f2us(nacc + g));
                    }
                }
            }
        }
        if (MULTI && fold_mode == 4) {   // the running sums as they are: the accumulator planes of the scatter form
            if (zvalid) {
                const size_t vi0 = ((size_t)x * p.V1 + y) * p.V2 + z;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = 8 * (i >> 2) + 4 * kh + (i & 3);
                    if (c < p.C) p.fold[(size_t)c * vv + vi0] = (unsigned short)(acch[i >> 1] >> (16 * (i & 1)));
                }
                if (kh == 0) p.raw_n[vi0] = f2us(nacc);
            }
            continue;
        }
        // normalise, fold sum / mean, argmax (k_finalize_labels)
        float acc[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[2 * i] = us2f((unsigned short)(acch[i] & 0xFFFFu));
            acc[2 * i + 1] = us2f((unsigned short)(acch[i] >> 16));
        }
        float best = 0.f;
        int bidx = 1 << 20;
        bool bnan = false, have = false;
        const size_t vi = ((size_t)x * p.V1 + y) * p.V2 + z;
        // Single fold (the part models): q(a) = half(a / n) is a monotone function of the running sum a (same positive divisor, RTNE
        // twice), so the argmax and the inf flag follow from the extremes -- two divisions per lane instead of sixteen (the
        // divisions were a third of the kernel's instructions at ~2 covering tiles per voxel).  first-maximum semantics: the winner
        // is the LOWEST class whose quotient equals the maximum's; only classes within 2^-8 of the largest sum can round to the
        // same half, and those few get their exact quotient.  NaN sums, quotients in the fp16-subnormal range and a zero maximum
        // take the general path (wave-uniform).
        bool fast = fold_mode == 0;
        float amax = 0.f, amin = 0.f, asum = 0.f;
        int imax = 1 << 20;
        if (fast) {
            bool first = true;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 8 * (i >> 2) + 4 * kh + (i & 3);
                if (c >= p.C) continue;
                const float a = acc[i];
                asum += a;
                if (first || a > amax) {   // (ascending classes, strict >: the lane's first maximum)
                    amax = a;
                    imax = c;
                }
                amin = first ? a : fminf(amin, a);
                first = false;
            }
            const unsigned short hmax = f2us(__fdiv_rn(amax, nacc)), hmin = f2us(__fdiv_rn(amin, nacc));
            const float qmax = us2f(hmax);
            const bool lane_has = imax < (1 << 20);
            // (lanes without classes -- kh = 1 when C <= 4 -- and runs beyond the volume do not vote)
            const bool slow = zvalid && lane_has && (asum != asum || !(fabsf(qmax) >= 1.3e-4f));
            if (__builtin_amdgcn_ballot_w64(slow) != 0) {
                fast = false;
            } else {   // (every lane runs this block: the shuffles and ballots below must stay convergent)
                if (zvalid && lane_has && ((hmax & 0x7FFF) == 0x7C00 || (hmin & 0x7FFF) == 0x7C00)) any_inf = true;
                // the voxel's maximum over both lanes (the other 16 classes live in lane ^ 32; a lane without classes -- kh = 1
                // when C <= 4 -- carries the sentinel index)
                const float oa = __shfl_xor(amax, 32), oq = __shfl_xor(qmax, 32);
                const int oi = __shfl_xor(imax, 32);
                float gmax = amax, gq = qmax;
                int gi = imax;
                if (oi < (1 << 20) && (!lane_has || oa > gmax || (oa == gmax && oi < gi))) {
                    gmax = oa;
                    gq = oq;
                    gi = oi;
                }
                // lower classes whose quotient ties with the maximum's
                const float lo = gmax - fabsf(gmax) * 0.00390625f;
                int ci = gi;
                // (one ballot for "any candidate at all" -- the common case has none -- instead of sixteen)
                unsigned cmask = 0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = 8 * (i >> 2) + 4 * kh + (i & 3);
                    const bool cand = zvalid && c < p.C && c < gi && acc[i] >= lo;
                    cmask |= cand ? (1u << i) : 0u;
                }
                if (__builtin_amdgcn_ballot_w64(cmask != 0u) != 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int c = 8 * (i >> 2) + 4 * kh + (i & 3);
                        const bool cand = (cmask >> i) & 1u;
                        if (__builtin_amdgcn_ballot_w64(cand) != 0) {
                            const float q = us2f(f2us(__fdiv_rn(acc[i], nacc)));
                            if (cand && q == gq) ci = min(ci, c);
                        }
                    }
                }
                bidx = min(ci, __shfl_xor(ci, 32));
                have = true;
            }
        }
        if (!fast) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 8 * (i >> 2) + 4 * kh + (i & 3);
            if (c >= p.C || !zvalid) continue;
            unsigned short h = f2us(__fdiv_rn(acc[i], nacc));   // torch.div(half, half): fp32 divide, RTNE to half
            if ((h & 0x7FFF) == 0x7C00) any_inf = true;
            if (fold_mode != 0) {
                unsigned short* fp = p.fold + (size_t)c * vv + vi;
                if (fold_mode >= 2) h = f2us(us2f(*fp) + us2f(h));                                  // prediction += fold
                if (fold_mode == 3 && p.n_folds > 1) h = f2us(__fdiv_rn(us2f(h), (float)p.n_folds));  // prediction /= n_folds
                if (fold_mode != 3) {
                    *fp = h;
                    continue;
                }
            }
            const float f = us2f(h);
            const bool isn = f != f;
            // numpy argmax over ascending classes: first maximum; the first NaN wins over everything
            if (!have) {
                best = f;
                bidx = c;
                bnan = isn;
                have = true;
            } else if (!bnan && (isn || f > best)) {
                best = f;
                bidx = c;
                bnan = isn;
            }
        }
        }
        if (fold_mode == 1 || fold_mode == 2) continue;
        // the voxel's other 16 classes live in lane ^ 32: combine (NaN first, then value, then the lower class index)
        if (!fast) {
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bidx, 32);
            const int on = __shfl_xor((int)bnan, 32);
            const int oh = __shfl_xor((int)have, 32);
            if (oh) {
                bool take;
                if (!have)
                    take = true;
                else if (bnan || on)
                    take = on && (!bnan || oi < bidx);
                else
                    take = ob > best || (ob == best && oi < bidx);
                if (take) {
                    best = ob;
                    bidx = oi;
                }
            }
        }
        if (kh == 0 && zvalid) {
            int ox = x, oy = y, oz = z;
            size_t oidx = vi;
            bool inside = true;
            if (p.crop) {
                ox -= p.o0;
                oy -= p.o1;
                oz -= p.o2;
                inside = ox >= 0 && oy >= 0 && oz >= 0 && ox < p.c0 && oy < p.c1 && oz < p.c2;
                oidx = ((size_t)ox * p.c1 + oy) * p.c2 + oz;
            }
            if (inside) {
                if (p.merge) {
                    if (bidx != 0) p.labels[oidx] = p.lut[bidx];
                } else {
                    p.labels[oidx] = p.lut[bidx];
                }
            }
        }
    }
    if (any_inf) atomicOr(p.inf_flag, 1);
}

template <bool GAUSS, bool SSLDS, bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GH_WAVES, 8))) void k_gather_head(GatherArgs p) {
    gather_head_body<GAUSS, SSLDS, MULTI, false>(p);
}

// prefetching variant of the fp16 head: 20 KB of LDS slots per block + the (scale, shift) table: four blocks per CU
#ifndef GH_PF_MULTI_WAVES
#define GH_PF_MULTI_WAVES 4   // experiment hook.  3 waves per EU (round 5, tools/gh_fold_time.py, five-fold 154 x 512 x 512): 157 VGPRs
                              // and no spills for the fold variants instead of 128 with 2 - 4 spilled, but 18.4 vs 17.3 ms per
                              // five fold launches: the occupancy is worth more than the spills, 4 stays.
#endif
template <bool GAUSS, bool SSLDS, bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MULTI ? GH_PF_MULTI_WAVES : 4, 8))) void k_gather_head_pf(GatherArgs p) {
    gather_head_body<GAUSS, SSLDS, MULTI, false, true>(p);
}

// (32 + 16 more registers for the split weights and the four fp32 octets: four waves per SIMD)
template <bool GAUSS, bool SSLDS, bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_gather_head_x3(GatherArgs p) {
    gather_head_body<GAUSS, SSLDS, MULTI, true>(p);
}

int launch_pack_head_ss(boa_ctx* ctx, const float* ss, unsigned* out, int n_tiles) {
    hipLaunchKernelGGL(k_pack_head_ss, dim3((unsigned)((n_tiles * 2 + 63) / 64)), dim3(64), 0, ctx->stream, ss, out, n_tiles);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_gather_head(boa_ctx* ctx, const __half* act, const unsigned* ssp, const float* w, const float* bias, const uint16_t* gauss,
                       int C, const int P[3], const int PV[3], const int ntile[3], const int* dev_tab, uint16_t* fold, int fold_mode,
                       int n_folds, const uint8_t* host_lut, int merge, uint8_t* labels, const int* crop_off, const int* crop_dims,
                       int* inf_flag, float slope, int tiles_total, bool x3, const int* x_range, uint16_t* raw_n, int raw_init) {
    BOA_REQUIRE(C >= 1 && C <= 32 && ntile[0] < 256 && ntile[1] < 256 && ntile[2] < 256, "gather head: C=%d / %d+%d+%d tiles per axis unsupported", C,
                ntile[0], ntile[1], ntile[2]);
    GatherArgs a;
    a.act = act; a.ssp = ssp; a.w = w; a.bias = bias; a.gauss = gauss; a.C = C;
    a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2]; a.V0 = PV[0]; a.V1 = PV[1]; a.V2 = PV[2];
    a.n0 = ntile[0]; a.n1 = ntile[1]; a.n2 = ntile[2]; a.tab = dev_tab;
    a.fold = fold; a.fold_mode = fold_mode; a.n_folds = n_folds; a.labels = labels; a.merge = merge;
    a.crop = crop_off != nullptr;
    a.o0 = a.o1 = a.o2 = 0;
    a.c0 = PV[0]; a.c1 = PV[1]; a.c2 = PV[2];
    if (crop_off) {
        a.o0 = crop_off[0]; a.o1 = crop_off[1]; a.o2 = crop_off[2];
        a.c0 = crop_dims[0]; a.c1 = crop_dims[1]; a.c2 = crop_dims[2];
    }
    a.inf_flag = inf_flag; a.slope = slope;
    a.x_lo = x_range ? x_range[0] : 0;
    a.x_hi = x_range ? x_range[1] : PV[0];
    a.raw_n = raw_n; a.raw_init = raw_init;
    BOA_REQUIRE(a.x_lo >= 0 && a.x_lo <= a.x_hi && a.x_hi <= PV[0] && (fold_mode != 4 || (fold && raw_n)), "gather head: plane range / raw buffers");
    a.wscale = X3_HEAD_WSCALE;
    a.winv = 1.0f / a.wscale;
    for (int i = 0; i < 256; ++i) a.lut[i] = host_lut ? host_lut[i] : (unsigned char)i;
    const long long n_mt = (long long)PV[0] * PV[1] * ((PV[2] + 31) / 32);
    BOA_REQUIRE(n_mt < (1ll << 31), "gather head: volume too large for 32-bit run indices");
    const long long n_mt_run = (long long)(a.x_hi - a.x_lo) * PV[1] * ((PV[2] + 31) / 32);
    const unsigned grid = (unsigned)std::min<long long>(std::max<long long>((n_mt_run + 3) / 4, 1), (long long)ctx->cu_count * 16);
    // algorithmic bytes: every tile's stash is read once (64 B activation + 2 B Gaussian per voxel), one label byte per voxel
    // (+ the fold buffer's read-modify-write)
    const double pvd = (double)P[0] * P[1] * P[2], vvd = (double)PV[0] * PV[1] * PV[2];
    const double bytes = fold_mode == 4 ? (double)n_mt_run * 32.0 * ((x3 ? 130.0 : 66.0) * 2.0 + 2.0 * (C + 1) * (raw_init ? 2 : 1))   // (~2 covering tiles per voxel)
                                        : (double)tiles_total * pvd * (x3 ? 130.0 : 66.0) + vvd * (fold_mode == 0 ? 1.0 : (fold_mode == 1 ? 2.0 * C : 4.0 * C));
    const size_t ss_bytes = (size_t)tiles_total * (x3 ? 256 : 128);
    a.ss_in_lds = ss_bytes <= 96 * 1024 ? 1 : 0;
    static bool once = (hipFuncSetAttribute((const void*)k_gather_head<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_pf<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_pf<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_pf<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_pf<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_x3<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_x3<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_x3<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                        hipFuncSetAttribute((const void*)k_gather_head_x3<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), true);
    (void)once;
    KernelTimer tm(ctx, BOA_K_HEAD_ACCUM, 2.0 * (double)tiles_total * pvd * 32 * C, bytes);
    const size_t lds = a.ss_in_lds ? ss_bytes : 0;
#define GH_LAUNCH(G, S, M)                                                                               \
    do {                                                                                                 \
        if (x3)                                                                                          \
            hipLaunchKernelGGL((k_gather_head_x3<G, S, M>), dim3(grid), dim3(256), lds, ctx->stream, a); \
        else if (pf)                                                                                     \
            hipLaunchKernelGGL((k_gather_head_pf<G, S, M>), dim3(grid), dim3(256), lds, ctx->stream, a); \
        else                                                                                             \
            hipLaunchKernelGGL((k_gather_head<G, S, M>), dim3(grid), dim3(256), lds, ctx->stream, a);    \
    } while (0)
    const bool multi = fold_mode != 0;
    static const bool pf = !(getenv("BOA_GH_PF") && atoi(getenv("BOA_GH_PF")) == 0);   // 0: the round-3 kernel without the LDS-DMA prefetch
    if (gauss && a.ss_in_lds) {
        if (multi) GH_LAUNCH(true, true, true); else GH_LAUNCH(true, true, false);
    } else if (gauss) {
        if (multi) GH_LAUNCH(true, false, true); else GH_LAUNCH(true, false, false);
    } else if (a.ss_in_lds) {
        if (multi) GH_LAUNCH(false, true, true); else GH_LAUNCH(false, true, false);
    } else {
        if (multi) GH_LAUNCH(false, false, true); else GH_LAUNCH(false, false, false);
    }
#undef GH_LAUNCH
    if (x3) ctx->counters[BOA_CNT_X3]++;
    ctx->counters[BOA_CNT_HEAD_GATHER]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
