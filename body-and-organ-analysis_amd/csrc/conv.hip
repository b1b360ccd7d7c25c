// 3-D convolution stack for gfx950: implicit-GEMM 3x3x3 / strided conv on f16 MFMA (32x32x16) with the
// producer's InstanceNorm+LeakyReLU applied while staging the LDS halo tile, InstanceNorm statistics reduced in
// the epilogue (deterministic per-block partials), first-layer fp32 VALU conv reading tiles out of the resident
// volume, transposed conv (kernel == stride) on MFMA, fused 1x1x1 head + Gaussian fp16 accumulation.
//
// Replaces `self.network(x)` (NN/inference/predict_from_raw_data.py:543), i.e. dynamic_network_architectures'
// PlainConvUNet as configured by NN/utilities/plans_handling/plans_handler.py:59-92.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include "conv.h"


// ======================================================================================================
// host-side weight packing
// A-operand image for v_mfma_f32_32x32x16_f16: lane l holds A[i = l & 31][k = 8 * (l >> 5) + j], j = 0..7, so a
// (tap, 16-channel chunk, k-half) fragment is 32 couts x 8 halves = 512 contiguous bytes.
// conv:  [cc = Cin/16][tap][khalf][Cout][8]
size_t conv_wpk_halves(int Cin_total, int Cout, const int k[3]) {
    return (size_t)(Cin_total / 16) * k[0] * k[1] * k[2] * 2 * Cout * 8;
}

void pack_conv_weights(const float* w, int Cin, int Cout, const int k[3], __half* dst) {
    const int taps = k[0] * k[1] * k[2];
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < taps; ++t) {
                int cc = ci / 16, kh = (ci % 16) / 8, j = ci % 8;
                size_t o = ((((size_t)cc * taps + t) * 2 + kh) * Cout + co) * 8 + j;
                dst[o] = __float2half_rn(w[((size_t)co * Cin + ci) * taps + t]);
            }
}

// convT: [tap][cc = Cin/16][khalf][Cout][8]
size_t convt_wpk_halves(int Cin, int Cout, const int s[3]) { return (size_t)s[0] * s[1] * s[2] * Cin * Cout; }

void pack_convt_weights(const float* w, int Cin, int Cout, const int s[3], __half* dst) {
    const int taps = s[0] * s[1] * s[2];
    const int ncc = Cin / 16;
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int t = 0; t < taps; ++t) {
                int cc = ci / 16, kh = (ci % 16) / 8, j = ci % 8;
                size_t o = ((((size_t)t * ncc + cc) * 2 + kh) * Cout + co) * 8 + j;
                dst[o] = __float2half_rn(w[((size_t)ci * Cout + co) * taps + t]);
            }
}

// ---- split-precision mode (precision 2) --------------------------------------------------------------
// One K = 16 MFMA step covers 8 real input channels: conv [cc = Cin/8][tap][part: hi, lo][Cout][8], convT
// [tap][cc = Cin/8][part][Cout][8]; every weight is scaled by `scale` (a power of two chosen per layer so that the largest
// magnitude sits just below 2^14: the lo parts of typical weights then stay out of the fp16 subnormal range) and split as
// hi = half(w s), lo = half(w s - hi).
float x3_weight_scale(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(w[i]));
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e;
    std::frexp(m, &e);               // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.f, std::max(-24, std::min(14 - e, 40)));   // m * scale in [2^13, 2^14)
}

static inline void x3_split(float v, __half* hi, __half* lo) {
    const __half h = __float2half_rn(v);
    *hi = h;
    *lo = __float2half_rn(v - __half2float(h));
}

size_t conv_wpk_halves_x3(int Cin_total, int Cout, const int k[3]) { return (size_t)(Cin_total / 8) * k[0] * k[1] * k[2] * 2 * Cout * 8; }

void pack_conv_weights_x3(const float* w, int Cin, int Cout, const int k[3], float scale, __half* dst) {
    const int taps = k[0] * k[1] * k[2];
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < taps; ++t) {
                const int cc = ci / 8, j = ci % 8;
                const size_t o = ((((size_t)cc * taps + t) * 2 + 0) * Cout + co) * 8 + j;
                x3_split(w[((size_t)co * Cin + ci) * taps + t] * scale, &dst[o], &dst[o + (size_t)Cout * 8]);
            }
}

size_t convt_wpk_halves_x3(int Cin, int Cout, const int s[3]) { return (size_t)s[0] * s[1] * s[2] * Cin * Cout * 2; }

void pack_convt_weights_x3(const float* w, int Cin, int Cout, const int s[3], float scale, __half* dst) {
    const int taps = s[0] * s[1] * s[2];
    const int ncc = Cin / 8;
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int t = 0; t < taps; ++t) {
                const int cc = ci / 8, j = ci % 8;
                const size_t o = ((((size_t)t * ncc + cc) * 2 + 0) * Cout + co) * 8 + j;
                x3_split(w[((size_t)ci * Cout + co) * taps + t] * scale, &dst[o], &dst[o + (size_t)Cout * 8]);
            }
}

// ======================================================================================================
// tile selection
static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

static size_t conv_lds_bytes(int HV, int taps) {
    size_t plane = (size_t)HV * 16 + 64;
    return 2 * plane + (size_t)taps * 1024 + (size_t)HV * 4 + 1024;
}

size_t conv_ws_lds_bytes(int HV, int taps, int ncc, int Cout);
bool conv_ws_supported(const int k[3], int HV);
bool conv_ws_resident(int HV, int taps, int ncc, int Cout);

static int conv_variant_override() {
    static int v = [] {
        const char* e = getenv("BOA_CONV_VARIANT");
        return e ? atoi(e) : -1;
    }();
    return v;
}

// Pick the wave M-tile shape, the block tile and the kernel variant.  Cost model (cycles per 32-voxel M-tile):
//   variant 1 (k_conv_ws): chunk time = max(MFMA time of the consumers, staging time of the producers) + barrier;
//   variant 0 (k_conv_mfma): MFMA and staging serialised inside a block, partly hidden by the second block on the CU.
bool choose_conv_tile(const ConvGeom& g, int cu_count, ConvTile* out, bool x3) {
    const int dims[3] = {g.Do, g.Ho, g.Wo};
    const int taps = g.k[0] * g.k[1] * g.k[2];
    const int ncc = x3 ? g.Cin / 8 : g.Cin / 16;   // split-precision mode: 8 real channels per staged chunk, two MFMAs per tap
    double best_cost = 1e30;
    bool found = false;
    const int force = x3 ? -1 : conv_variant_override();
    // (split-precision mode: the stride-2 layers take the N-split kernel too -- its stride-1 instantiation is not built for X3)
    if (force < 0 && conv_ns_applicable(g) && !(x3 && g.s[0] != 2) && !(x3 && getenv("BOA_X3_NO_NS"))) {  // N-split kernel (conv_ns.hip): stride-2 and deep layers, fixed tile shape
        conv_ns_tile(g, out);
        return true;
    }
    // wave M-tile shapes: 32 voxels = w0 x w1 x w2 (powers of two), contiguous axis as long as possible first
    // (second pass: extents so small that no wave tile fits without overhang -- e.g. a last axis of 2 -- take any shape; the
    // kernels mask the overhang)
    for (int relax = 0; relax < 2 && !found; ++relax)
    for (int w2 = 32; w2 >= 4; w2 >>= 1) {
        if (!relax && w2 > next_pow2(dims[2])) continue;
        for (int w1 = 1; w1 * w2 <= 32; w1 <<= 1) {
            const int w0 = 32 / (w2 * w1);
            if (!relax && (w1 > next_pow2(dims[1]) || w0 > next_pow2(dims[0]))) continue;
            const int w[3] = {w0, w1, w2};
            for (int variant : {1, 0}) {
                if ((force >= 0 && variant != force) || (x3 && variant != 1)) continue;
                for (int R : {4, 2, 1}) {
                    const int M = 4 * R;
                    for (int b0 = 1; b0 <= M; b0 *= 2)
                        for (int b1 = 1; b0 * b1 <= M; b1 *= 2) {
                            const int b2 = M / (b0 * b1);
                            if (b0 * b1 * b2 != M) continue;
                            const int b[3] = {b0, b1, b2};
                            int h[3], tl[3];
                            long long HV = 1, covered = 1, tiles = 1;
                            for (int d = 0; d < 3; ++d) {
                                const int ext = b[d] * w[d];
                                h[d] = (ext - 1) * g.s[d] + g.k[d];
                                tl[d] = ceil_div(dims[d], ext);
                                HV *= h[d];
                                covered *= (long long)tl[d] * ext;
                                tiles *= tl[d];
                            }
                            // k_conv_ws, M-tiles that span several x-planes (w0 > 1, rows of w2 = 8 / 16 lanes: the 16^3 ... 4^3 layers):
                            // lane row lx sits xs voxels = xs * 16 bytes behind row lx - 1, and with xs = h1 * h2 (180 / 100 at the 16^3 / 8^3
                            // layers: = 64 bytes mod 256) the rows of a ds_read_b128 lane group overlapped on the banks -- 35 - 44 % of these
                            // launches' LDS cycles were conflicts (profiles/r05_pmc_lds.txt).  The planes are padded to xs = w2 (mod 16): the
                            // rows of every lane group then tile a 256-byte bank row.  HV below counts the padded planes.
                            int xs = h[1] * h[2];
                            if (variant == 1 && w0 > 1 && w1 == 1 && w2 >= 8 && w2 <= 16 && g.s[0] == 1 && g.s[2] == 1 && !getenv("BOA_WS_NO_XPAD"))
                                while (xs % 16 != w2 % 16) ++xs;
                            if (variant == 1) HV = (long long)h[0] * xs;
                            if (variant == 1 && !conv_ws_supported(g.k, (int)HV)) continue;
                            const size_t lds = variant == 1 ? conv_ws_lds_bytes((int)HV, taps, ncc, g.Cout)
                                                            : conv_lds_bytes((int)HV, taps);
                            if (lds > 160 * 1024 - 2048) continue;   // (2 KiB stay free for the split-precision kernel's bias table)
                            const long long nblocks = tiles * (g.Cout / 32) * g.N;
                            // experiment hook: BOA_CONV_TILE="w0,w1,w2,b0,b1,b2" forces that shape wherever it is legal
                            static int ft[6] = {0, 0, 0, 0, 0, 0};
                            static const bool have_ft = getenv("BOA_CONV_TILE") &&
                                sscanf(getenv("BOA_CONV_TILE"), "%d,%d,%d,%d,%d,%d", ft, ft + 1, ft + 2, ft + 3, ft + 4, ft + 5) == 6;
                            const bool forced = have_ft && variant == 1 && g.s[0] == 1 && w0 == ft[0] && w1 == ft[1] && w2 == ft[2] &&
                                                b0 == ft[3] && b1 == ft[4] && b2 == ft[5];
                            const double valid = (double)dims[0] * dims[1] * dims[2];
                            const double waste = (double)covered / valid;
                            // LDS fragment reads are conflict-free when a wave row is contiguous (w2 lanes at the
                            // stride of the conv); short rows / strided rows cost extra LDS cycles
                            // (measured: an 8x8x8 block tile of 4x1x8 wave tiles stages 26 % fewer halo voxels than 4x4x32 of 1x1x32 wave
                            // tiles and is still 2-4 % slower: rows shorter than 32 lanes cost fragment-read conflicts)
                            const double lds_pen = (g.s[2] > 1 ? 1.25 : 1.0) * (w2 < 16 ? 1.3 : (w2 < 32 ? 1.15 : 1.0));
                            const double t_mfma = (double)taps * R * 32.0 * lds_pen * (x3 ? 2.0 : 1.0);
                            double t_chunk;
                            const double slots = cu_count;
                            if (variant == 1) {
                                const bool res = conv_ws_resident((int)HV, taps, ncc, g.Cout);
                                // measured (s_memtime stamps, 32->32 @128^3): a 1360-voxel chunk costs the producers ~6500
                                // cycles next to the consumers' MFMA stream -> ~4.3 cycles per halo voxel + fixed part
                                const double t_prod = (double)h[0] * h[1] * h[2] * 4.3 + (res ? 0.0 : taps * 64.0 / 256.0 * 60.0) + 800.0;   // (the halo's voxels, not the padded planes)
                                t_chunk = std::max(t_mfma * 1.15, t_prod * (g.s[2] > 1 ? 1.0 : lds_pen)) + 500.0;
                            } else {
                                const double items = (2.0 * HV + taps * 64.0) / 256.0;
                                const int bpc = lds <= 78 * 1024 ? 2 : 1;
                                t_chunk = (taps * R * 200.0 + items * 1500.0 + 400.0) / (bpc == 2 ? 1.8 : 1.0);
                            }
                            // epilogue + tile turnaround amortised over the chunks of a tile
                            const double t_tile = t_chunk * ncc + (variant == 1 ? 3000.0 : 6000.0);
                            const double rounds = std::ceil((double)nblocks / slots);
                            const double cost = forced ? -1.0 : t_tile * rounds * waste / (double)M * (slots / (double)nblocks);
                            if (cost < best_cost) {
                                best_cost = cost;
                                found = true;
                                out->variant = variant;
                                out->R = R;
                                for (int d = 0; d < 3; ++d) {
                                    out->w[d] = w[d];
                                    out->b[d] = b[d];
                                    out->h[d] = h[d];
                                    out->tiles[d] = tl[d];
                                }
                                out->lds_bytes = lds;
                                out->xs = xs;
                            }
                        }
                }
            }
        }
    }
    return found;
}

// number of per-(n, cout) partial-statistics entries the conv kernel writes
int conv_nblk(const ConvTile& t, int cu_count, int Cout) {
    const int spatial = t.tiles[0] * t.tiles[1] * t.tiles[2];
    if (t.variant == 2) return conv_ws_nslots(spatial * conv_ns_ncy(Cout), cu_count);
    return t.variant == 1 ? conv_ws_nslots(spatial * (Cout / 32), cu_count) : spatial;
}

// ======================================================================================================
// MFMA conv kernel
template <int R>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    const int taps = p.k0 * p.k1 * p.k2;
    const int HV = p.h0 * p.h1 * p.h2;
    const int plane = HV * 16 + 64;
    unsigned char* lds_in = smem;
    unsigned char* lds_w = smem + 2 * plane;
    int* lds_tab = (int*)(lds_w + taps * 1024);
    float* lds_red = (float*)(lds_tab + HV);

    const int n = blockIdx.z;
    const int cout0 = blockIdx.y * 32;
    int bt = blockIdx.x;
    const int tz = bt % p.t2;
    bt /= p.t2;
    const int ty = bt % p.t1;
    const int tx = bt / p.t1;
    const int ox0 = tx * p.b0 * p.w0, oy0 = ty * p.b1 * p.w1, oz0 = tz * p.b2 * p.w2;
    const int ix0 = ox0 * p.s0 - p.p0, iy0 = oy0 * p.s1 - p.p1, iz0 = oz0 * p.s2 - p.p2;

    // halo voxel -> input voxel index table (-1 = zero padding), reused by every channel chunk
    for (int v = tid; v < HV; v += 256) {
        int hz = v % p.h2;
        int t = v / p.h2;
        int hy = t % p.h1;
        int hx = t / p.h1;
        int ix = ix0 + hx, iy = iy0 + hy, iz = iz0 + hz;
        bool inb = ix >= 0 && ix < p.Di && iy >= 0 && iy < p.Hi && iz >= 0 && iz < p.Wi;
        lds_tab[v] = inb ? (ix * p.Hi + iy) * p.Wi + iz : -1;
    }

    // per-lane voxel of each of this wave's M-tiles
    const int lz = l31 % p.w2;
    const int ly = (l31 / p.w2) % p.w1;
    const int lx = l31 / (p.w2 * p.w1);
    int hoff[R];
    int ovox[R];  // output voxel index or -1
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int m = wave * R + r;
        int mz = m % p.b2;
        int t = m / p.b2;
        int my = t % p.b1;
        int mx = t / p.b1;
        int tx_ = mx * p.w0 + lx, ty_ = my * p.w1 + ly, tz_ = mz * p.w2 + lz;
        hoff[r] = ((tx_ * p.s0) * p.h1 + ty_ * p.s1) * p.h2 + tz_ * p.s2;
        int ox = ox0 + tx_, oy = oy0 + ty_, oz = oz0 + tz_;
        ovox[r] = (ox < p.Do && oy < p.Ho && oz < p.Wo) ? (ox * p.Ho + oy) * p.Wo + oz : -1;
    }

    f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;

    const size_t in_vox = (size_t)p.Di * p.Hi * p.Wi;
    const int ncc = (p.C0 + p.C1) / 16;
    const int oct = tid & 1;  // this thread always stages the same channel octet of a chunk

    for (int cc = 0; cc < ncc; ++cc) {
        __syncthreads();  // previous chunk fully consumed (also orders the table writes before first use)
        // ---- stage the halo tile of 16 channels, applying the producer's norm + LeakyReLU -------------
        {
            int cg = cc * 16 + oct * 8;
            const __half* base;
            const float* ss;
            int C;
            // chunk-planar activations [N][C/16][voxel][16]: plane of this chunk + this thread's octet
            if (cg < p.C0) {
                base = p.src0 + ((size_t)n * p.C0 + (cg & ~15)) * in_vox + (cg & 15);
                ss = p.ss0 ? p.ss0 + ((size_t)n * p.C0 + cg) * 2 : nullptr;
                C = p.C0;
            } else {
                cg -= p.C0;
                base = p.src1 + ((size_t)n * p.C1 + (cg & ~15)) * in_vox + (cg & 15);
                ss = p.ss1 ? p.ss1 + ((size_t)n * p.C1 + cg) * 2 : nullptr;
                C = p.C1;
            }
            (void)C;
            float sc[8], sh[8];
            if (ss) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sc[j] = ss[2 * j];
                    sh[j] = ss[2 * j + 1];
                }
            }
            unsigned char* dstp = lds_in + oct * plane;
            for (int i = tid; i < 2 * HV; i += 256) {
                int v = i >> 1;
                int gi = lds_tab[v];
                uint4 val = make_uint4(0, 0, 0, 0);
                if (gi >= 0) {
                    val = *(const uint4*)(base + (size_t)gi * 16);
                    if (ss) val = norm_act8(val, sc, sh, p.slope);
                }
                *(uint4*)(dstp + v * 16) = val;
            }
        }
        // ---- stage this chunk's weights for the block's 32 output channels ----------------------------
        {
            const __half* wsrc = p.wpk + ((size_t)cc * taps * 2) * p.Cout * 8;
            for (int i = tid; i < taps * 64; i += 256) {
                int seg = i >> 5, co = i & 31;
                *(uint4*)(lds_w + i * 16) = *(const uint4*)(wsrc + ((size_t)seg * p.Cout + cout0 + co) * 8);
            }
        }
        __syncthreads();
        // ---- MFMA over the taps -----------------------------------------------------------------------
        const unsigned char* bbase = lds_in + kh * plane;
        const unsigned char* abase = lds_w + (kh * 32 + l31) * 16;
        int tap = 0;
        for (int dx = 0; dx < p.k0; ++dx)
            for (int dy = 0; dy < p.k1; ++dy) {
                const int rowoff = (dx * p.h1 + dy) * p.h2;
#pragma unroll 3
                for (int dz = 0; dz < p.k2; ++dz, ++tap) {
                    f16x8 a = *(const f16x8*)(abase + tap * 1024);
                    const int toff = rowoff + dz;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f16x8 b = *(const f16x8*)(bbase + (hoff[r] + toff) * 16);
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r], 0, 0, 0);
                    }
                }
            }
    }

    // ---- epilogue: + bias, fp16 store, InstanceNorm partial statistics ------------------------------
    // D layout: lane holds column (voxel) l31, rows (couts) (i & 3) + 8 * (i >> 2) + 4 * kh
    float s[16], q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = q[i] = 0.f;
    float bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bv[i] = p.bias[cout0 + (i & 3) + 8 * (i >> 2) + 4 * kh];
    const size_t out_vox = (size_t)p.Do * p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (ovox[r] < 0) continue;
        // couts cout0 + 8 g + 4 kh + j -> plane cout0 / 16 + g / 2, offset 8 (g % 2) + 4 kh + j
        __half* op = p.out + ((size_t)n * p.Cout + cout0) * out_vox + (size_t)ovox[r] * 16 + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            union {
                uint2 u;
                __half h[4];
            } pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[r][g * 4 + j] + bv[g * 4 + j];
                __half hv = __float2half_rn(v);
                pk.h[j] = hv;
                float vr = __half2float(hv);
                s[g * 4 + j] += vr;
                q[g * 4 + j] = __builtin_fmaf(vr, vr, q[g * 4 + j]);
            }
            *(uint2*)(op + (size_t)(g >> 1) * 16 * out_vox + 8 * (g & 1)) = pk.u;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            s[i] += __shfl_xor(s[i], m);
            q[i] += __shfl_xor(q[i], m);
        }
    }
    __syncthreads();
    if (l31 == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
            lds_red[(wave * 32 + row) * 2 + 0] = s[i];
            lds_red[(wave * 32 + row) * 2 + 1] = q[i];
        }
    }
    __syncthreads();
    if (tid < 64) {
        int row = tid >> 1, j = tid & 1;
        float v = lds_red[(0 * 32 + row) * 2 + j] + lds_red[(1 * 32 + row) * 2 + j];
        v += lds_red[(2 * 32 + row) * 2 + j];
        v += lds_red[(3 * 32 + row) * 2 + j];
        const int nblk = gridDim.x;
        p.partials[(((size_t)n * p.Cout + cout0 + row) * 2 + j) * nblk + blockIdx.x] = v;
    }
}

int launch_conv_mfma(boa_ctx* ctx, const ActSrc& s0, const ActSrc& s1, const ConvGeom& g, const ConvTile& t,
                     const __half* wpk, const float* bias, float slope, __half* out, float* partials) {
    BOA_REQUIRE(s0.C % 16 == 0 && s1.C % 16 == 0 && s0.C > 0, "conv: input channels (%d,%d) must be multiples of 16",
                s0.C, s1.C);
    BOA_REQUIRE(g.Cout % 32 == 0, "conv: Cout=%d must be a multiple of 32", g.Cout);
    ConvArgs a;
    a.src0 = s0.data; a.src1 = s1.data; a.ss0 = s0.ss; a.ss1 = s1.ss; a.C0 = s0.C; a.C1 = s1.C;
    a.ss16_0 = s0.ss16; a.ss16_1 = s1.ss16;
    BOA_REQUIRE((s0.ss == nullptr) == (s0.ss16 == nullptr) && (s1.ss == nullptr) == (s1.ss16 == nullptr),
                "conv: ss and ss16 must be given together");
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Cout;
    a.k0 = g.k[0]; a.k1 = g.k[1]; a.k2 = g.k[2]; a.s0 = g.s[0]; a.s1 = g.s[1]; a.s2 = g.s[2];
    a.p0 = (g.k[0] - 1) / 2; a.p1 = (g.k[1] - 1) / 2; a.p2 = (g.k[2] - 1) / 2;
    a.w0 = t.w[0]; a.w1 = t.w[1]; a.w2 = t.w[2]; a.b0 = t.b[0]; a.b1 = t.b[1]; a.b2 = t.b[2];
    a.h0 = t.h[0]; a.h1 = t.h[1]; a.h2 = t.h[2]; a.t0 = t.tiles[0]; a.t1 = t.tiles[1]; a.t2 = t.tiles[2];
    a.xs = t.xs > 0 ? t.xs : t.h[1] * t.h[2];
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    a.lw1 = ilog2(t.w[1]); a.lw2 = ilog2(t.w[2]); a.lb1 = ilog2(t.b[1]); a.lb2 = ilog2(t.b[2]);
    a.wpk = wpk; a.bias = bias; a.out = out; a.partials = partials; a.slope = slope;
    dim3 grid(t.tiles[0] * t.tiles[1] * t.tiles[2], g.Cout / 32, g.N);
    const int taps = g.k[0] * g.k[1] * g.k[2];
    const double vox = (double)g.N * g.Do * g.Ho * g.Wo;
    const double flops = 2.0 * vox * taps * (s0.C + s1.C) * g.Cout;
    const double bytes = 2.0 * ((double)g.N * g.Di * g.Hi * g.Wi * (s0.C + s1.C) + vox * g.Cout);
    if (t.variant == 2) return launch_conv_ns(ctx, a, t, flops, bytes);
    if (t.variant == 1) return launch_conv_ws(ctx, a, t, flops, bytes);
    ctx->counters[BOA_CNT_CONV_SIMPLE]++;
    KernelTimer tm(ctx, BOA_K_CONV_MFMA, flops, bytes);
    switch (t.R) {
        case 4: {
            static bool once = (hipFuncSetAttribute((const void*)k_conv_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
            (void)once;
            hipLaunchKernelGGL(k_conv_mfma<4>, grid, dim3(256), t.lds_bytes, ctx->stream, a);
            break;
        }
        case 2: {
            static bool once = (hipFuncSetAttribute((const void*)k_conv_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
            (void)once;
            hipLaunchKernelGGL(k_conv_mfma<2>, grid, dim3(256), t.lds_bytes, ctx->stream, a);
            break;
        }
        case 1: {
            static bool once = (hipFuncSetAttribute((const void*)k_conv_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
            (void)once;
            hipLaunchKernelGGL(k_conv_mfma<1>, grid, dim3(256), t.lds_bytes, ctx->stream, a);
            break;
        }
        default:
            boa_set_error("conv: unsupported R=%d", t.R);
            return BOA_EINVAL;
    }
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// split-precision conv (precision 2): k_conv_ws<..., X3>.  Sources / output are fp32 octet planes [N][C/8][voxel][8]; `ss` the
// producer's fp32 (scale, shift) table [N][C][2] or nullptr (raw source).  The kernel sees 2-byte channel units (2 C).
int launch_conv_x3(boa_ctx* ctx, const float* src0, const float* ss0, int C0, const float* src1, const float* ss1, int C1,
                   const ConvGeom& g, const ConvTile& t, const __half* wpk, float wscale, const float* bias, float slope, float* out,
                   float* partials) {
    BOA_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0 && C0 > 0, "conv_x3: input channels (%d,%d) must be multiples of 8", C0, C1);
    BOA_REQUIRE(g.Cout % 32 == 0, "conv_x3: Cout=%d must be a multiple of 32", g.Cout);
    BOA_REQUIRE(t.variant == 1 || t.variant == 2, "conv_x3: tile variant %d", t.variant);
    ConvArgs a;
    a.src0 = (const __half*)src0; a.src1 = (const __half*)src1; a.ss0 = ss0; a.ss1 = ss1; a.C0 = 2 * C0; a.C1 = 2 * C1;
    a.ss16_0 = (const unsigned*)ss0; a.ss16_1 = (const unsigned*)ss1;   // read as 16 fp32 words per 8-channel chunk
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Cout;
    a.k0 = g.k[0]; a.k1 = g.k[1]; a.k2 = g.k[2]; a.s0 = g.s[0]; a.s1 = g.s[1]; a.s2 = g.s[2];
    a.p0 = (g.k[0] - 1) / 2; a.p1 = (g.k[1] - 1) / 2; a.p2 = (g.k[2] - 1) / 2;
    a.w0 = t.w[0]; a.w1 = t.w[1]; a.w2 = t.w[2]; a.b0 = t.b[0]; a.b1 = t.b[1]; a.b2 = t.b[2];
    a.h0 = t.h[0]; a.h1 = t.h[1]; a.h2 = t.h[2]; a.t0 = t.tiles[0]; a.t1 = t.tiles[1]; a.t2 = t.tiles[2];
    a.xs = t.xs > 0 ? t.xs : t.h[1] * t.h[2];
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    a.lw1 = ilog2(t.w[1]); a.lw2 = ilog2(t.w[2]); a.lb1 = ilog2(t.b[1]); a.lb2 = ilog2(t.b[2]);
    a.wpk = wpk; a.bias = bias; a.out = (__half*)out; a.partials = partials; a.slope = slope;
    a.wscale = wscale; a.winv = 1.0f / wscale;
    const int taps = g.k[0] * g.k[1] * g.k[2];
    const double vox = (double)g.N * g.Do * g.Ho * g.Wo;
    const double flops = 2.0 * vox * taps * (C0 + C1) * g.Cout;
    const double bytes = 4.0 * ((double)g.N * g.Di * g.Hi * g.Wi * (C0 + C1) + vox * g.Cout);
    if (t.variant == 2) return launch_conv_ns(ctx, a, t, flops, bytes, true);
    return launch_conv_ws(ctx, a, t, flops, bytes, true);
}

// ======================================================================================================
// first conv: fp32 VALU, tiles gathered from the resident volume
// Stage 1: k_gather_patches copies the N tiles out of the resident volume into a zero-padded dense fp32 buffer
// [N][Cin][PX][PY][PZ] (conv padding + pad_nd_image zeros + tile overhang), so that stage 2 has no bounds logic.
// Stage 2: k_conv_first<K0,K1,K2>: each thread computes FV consecutive voxels along the contiguous axis x 32 output
// channels, so one LDS read of a weight quad feeds 4 * FV FMAs and one input value feeds up to 3 taps x 32
// channels.  Weights live in LDS ([Cin][tap][32] floats, broadcast reads); a block covers FT0 x FT1 x FT2 voxels.
#define FT0 4
#define FT1 4
#define FT2 64
#define FV 4

__global__ __launch_bounds__(256) void k_gather_patches(const float* __restrict__ vol, const int* __restrict__ origins,
                                                        int V0, int V1, int V2, int o0, int o1, int o2, int Cin, int P0,
                                                        int P1, int P2, int pad0, int pad1, int pad2, int PX, int PY, int PZ,
                                                        int flip, float* __restrict__ out) {
    const int n = blockIdx.z, ci = blockIdx.y;
    const unsigned pvol = (unsigned)(PX * PY * PZ);  // (a padded tile is far below 2^31 voxels: 32-bit index divisions)
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pvol) return;
    const unsigned r = i / (unsigned)PZ;
    const int z = (int)(i - r * (unsigned)PZ), x = (int)(r / (unsigned)PY), y = (int)(r - (unsigned)x * (unsigned)PY);
    const int px = x - pad0, py = y - pad1, pz = z - pad2;  // patch coordinates
    float v = 0.f;
    if (px >= 0 && px < P0 && py >= 0 && py < P1 && pz >= 0 && pz < P2) {
        // test-time mirroring (predict_from_raw_data.py:541-557): the network sees torch.flip(tile, axes)
        const int qx = (flip & 1) ? P0 - 1 - px : px, qy = (flip & 2) ? P1 - 1 - py : py, qz = (flip & 4) ? P2 - 1 - pz : pz;
        const int vx = origins[n * 3 + 0] + qx - o0, vy = origins[n * 3 + 1] + qy - o1, vz = origins[n * 3 + 2] + qz - o2;
        if (vx >= 0 && vx < V0 && vy >= 0 && vy < V1 && vz >= 0 && vz < V2)
            v = vol[(size_t)ci * V0 * V1 * V2 + ((size_t)vx * V1 + vy) * V2 + vz];
    }
    out[((size_t)n * Cin + ci) * pvol + i] = v;
}

struct FirstArgs {
    const float* padded;  // [N][Cin][PX][PY][PZ]
    int PX, PY, PZ;
    int N, Cin, P0, P1, P2, Cout;
    const float* w;  // [Cin][taps][Cout]
    const float* bias;
    __half* out;
    float* partials;
    int t0, t1, t2;
    int nblk;  // stride of the partials table (>= number of entries a launch writes)
    float* out32;  // F32OUT: fp32 octet planes [N][Cout/8][voxel][8] (split-precision mode)
};

template <int K0, int K1, int K2, bool F32OUT>
__global__ __launch_bounds__(256) void k_conv_first(FirstArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const int cout0 = blockIdx.y * 32;
    int bt = blockIdx.x;
    const int tz = bt % p.t2;
    bt /= p.t2;
    const int ty = bt % p.t1;
    const int tx = bt / p.t1;
    constexpr int h0 = FT0 + K0 - 1, h1 = FT1 + K1 - 1, h2 = FT2 + K2 - 1;
    constexpr int HV = h0 * h1 * h2;
    constexpr int taps = K0 * K1 * K2;
    float* lds_w = (float*)smem;                // [Cin][taps][32]
    float* lds_in = lds_w + p.Cin * taps * 32;  // [Cin][HV]
    float* lds_red = lds_in + ((p.Cin * HV + 3) & ~3);
    for (int i = tid; i < p.Cin * taps * 32; i += 256) lds_w[i] = p.w[(size_t)(i >> 5) * p.Cout + cout0 + (i & 31)];
    {
        // halo tile from the padded buffer: always in bounds, compile-time index arithmetic, 4 loads in flight
        const size_t pvol = (size_t)p.PX * p.PY * p.PZ;
        const float* src = p.padded + (size_t)n * p.Cin * pvol;
        const int total = p.Cin * HV;
        for (int i0 = tid; i0 < total; i0 += 256 * 4) {
            float v[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = min(i0 + b * 256, total - 1);
                const int ci = i / HV, r = i % HV;
                const int hz = r % h2, hy = (r / h2) % h1, hx = r / (h2 * h1);
                v[b] = src[(size_t)ci * pvol + ((size_t)(tx * FT0 + hx) * p.PY + (ty * FT1 + hy)) * p.PZ + tz * FT2 + hz];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) lds_in[min(i0 + b * 256, total - 1)] = v[b];
        }
    }
    __syncthreads();
    const int lzq = tid % (FT2 / FV);
    const int ly = (tid / (FT2 / FV)) % FT1;
    const int lx = tid / ((FT2 / FV) * FT1);
    const int lz = lzq * FV;
    const int ox = tx * FT0 + lx, oy = ty * FT1 + ly, oz = tz * FT2 + lz;
    float acc[FV][32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const float b = p.bias[cout0 + c];
#pragma unroll
        for (int v = 0; v < FV; ++v) acc[v][c] = b;
    }
    // The weight reads are wave-uniform; hipcc would scalarise all 27 x 32 of them (v_readfirstlane into SGPRs,
    // ~2700 SGPR spills, occupancy 1).  A lane-opaque zero keeps them as plain broadcast LDS reads.
    int lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    for (int ci = 0; ci < p.Cin; ++ci) {
        for (int dx = 0; dx < K0; ++dx)
            for (int dy = 0; dy < K1; ++dy) {
                const float* row = lds_in + ci * HV + ((lx + dx) * h1 + (ly + dy)) * h2 + lz;
                float xin[FV + K2 - 1];
#pragma unroll
                for (int j = 0; j < FV + K2 - 1; ++j) xin[j] = row[j];
                const float* wrow = lds_w + ((ci * taps) + (dx * K1 + dy) * K2) * 32 + lane_zero;
#pragma unroll
                for (int dz = 0; dz < K2; ++dz) {
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const float4 w4 = *(const float4*)(wrow + dz * 32 + c4 * 4);  // broadcast read
#pragma unroll
                        for (int v = 0; v < FV; ++v) {
                            acc[v][c4 * 4 + 0] = __builtin_fmaf(xin[v + dz], w4.x, acc[v][c4 * 4 + 0]);
                            acc[v][c4 * 4 + 1] = __builtin_fmaf(xin[v + dz], w4.y, acc[v][c4 * 4 + 1]);
                            acc[v][c4 * 4 + 2] = __builtin_fmaf(xin[v + dz], w4.z, acc[v][c4 * 4 + 2]);
                            acc[v][c4 * 4 + 3] = __builtin_fmaf(xin[v + dz], w4.w, acc[v][c4 * 4 + 3]);
                        }
                    }
                }
            }
    }
    float s[32], q[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) s[c] = q[c] = 0.f;
#pragma unroll
    for (int v = 0; v < FV; ++v) {
        if (F32OUT && ox < p.P0 && oy < p.P1 && oz + v < p.P2) {
            // split-precision mode: the fp32 sums are stored as they are (statistics of the stored values)
            const size_t pvox = (size_t)p.P0 * p.P1 * p.P2;
            float* op = p.out32 + ((size_t)n * p.Cout + cout0) * pvox + ((((size_t)ox) * p.P1 + oy) * (size_t)p.P2 + (oz + v)) * 8;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                s[c] += acc[v][c];
                q[c] = __builtin_fmaf(acc[v][c], acc[v][c], q[c]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *(float4*)(op + (size_t)(j >> 1) * 8 * pvox + 4 * (j & 1)) = make_float4(acc[v][4 * j], acc[v][4 * j + 1], acc[v][4 * j + 2], acc[v][4 * j + 3]);
        } else if (!F32OUT && ox < p.P0 && oy < p.P1 && oz + v < p.P2) {
            const size_t pvox = (size_t)p.P0 * p.P1 * p.P2;
            __half* op = p.out + ((size_t)n * p.Cout + cout0) * pvox + ((((size_t)ox) * p.P1 + oy) * (size_t)p.P2 + (oz + v)) * 16;
            union {
                uint4 u[4];
                __half h[32];
            } pk;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                __half hv = __float2half_rn(acc[v][c]);
                pk.h[c] = hv;
                float vr = __half2float(hv);
                s[c] += vr;
                q[c] = __builtin_fmaf(vr, vr, q[c]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) *(uint4*)(op + (size_t)(j >> 1) * 16 * pvox + 8 * (j & 1)) = pk.u[j];  // chunk-planar
        }
    }
    // recursive-halving reduction over the wave: after step m a lane keeps half of its channels, summed with its
    // partner's copy; 16 + 8 + ... shuffles per quantity instead of 6 x 32.  Lane l ends up owning channel
    // 16*b0 + 8*b1 + 4*b2 + 2*b3 + b4 (b_i = bit i of the lane id).
    const int lane_ = tid & 63;
#define HALVE_STEP(M, HALF)                                                                  \
    {                                                                                        \
        const bool up = (lane_ & (M)) != 0;                                                  \
        _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                                 \
            const float ks = up ? s[i + (HALF)] : s[i], ss_ = up ? s[i] : s[i + (HALF)];     \
            const float kq = up ? q[i + (HALF)] : q[i], sq_ = up ? q[i] : q[i + (HALF)];     \
            s[i] = ks + __shfl_xor(ss_, (M));                                                \
            q[i] = kq + __shfl_xor(sq_, (M));                                                \
        }                                                                                    \
    }
    HALVE_STEP(1, 16)
    HALVE_STEP(2, 8)
    HALVE_STEP(4, 4)
    HALVE_STEP(8, 2)
    HALVE_STEP(16, 1)
#undef HALVE_STEP
    s[0] += __shfl_xor(s[0], 32);
    q[0] += __shfl_xor(q[0], 32);
    const int wave = tid >> 6;
    if (lane_ < 32) {
        const int c = ((lane_ & 1) << 4) | ((lane_ & 2) << 2) | (lane_ & 4) | ((lane_ & 8) >> 2) | ((lane_ & 16) >> 4);
        lds_red[(wave * 32 + c) * 2 + 0] = s[0];
        lds_red[(wave * 32 + c) * 2 + 1] = q[0];
    }
    __syncthreads();
    if (tid < 64) {
        int row = tid >> 1, j = tid & 1;
        float v = lds_red[(0 * 32 + row) * 2 + j] + lds_red[(1 * 32 + row) * 2 + j];
        v += lds_red[(2 * 32 + row) * 2 + j];
        v += lds_red[(3 * 32 + row) * 2 + j];
        p.partials[(((size_t)n * p.Cout + cout0 + row) * 2 + j) * p.nblk + blockIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------------------
// First conv on the matrix cores (Cin == 1, 3x3x3, Cout == 32): the 27 taps are the K dimension (padded to 32 = two
// v_mfma_f32_32x32x16_f16 steps).  A = weights [cout][k] in registers for the whole kernel, B = im2col fragment gathered
// from an fp16 halo tile in LDS (lane (voxel z, k-half) reads its 8 taps with 2-byte LDS loads), D[cout][voxel] goes
// through the same epilogue as k_conv_ws (bias, fp32 InstanceNorm partial sums, v_permlane32_swap transpose, two 16-byte
// stores per lane).  864 fp32 FMAs per voxel become 2 MFMAs per 32 voxels: the kernel is bound by the 64 B/voxel store.
// Persistent blocks walk block tiles of MF0 x MF1 x 32 voxels; M-tile = 32 consecutive z at fixed (x, y).
#define MF0 4
#define MF1 8
#define MF2 32

struct FirstMfmaArgs {
    const float* padded;  // [N][PX][PY][PZ] fp32, conv padding included
    int PX, PY, PZ;
    int N, P0, P1, P2;
    const float* w;  // [27][32] fp32
    const float* bias;
    __half* out;      // [N][P0][P1][P2][32]
    float* partials;  // [N][32][2][nslots]
    int nslots;
    int t0, t1, t2;  // block tiles per axis
    int vw;          // virtual workgroups per sample
    // fused tile gather (vol != NULL): the halo is read straight out of the resident volume -- tile origin, pad_nd_image zeros,
    // conv padding, tile overhang and the test-time flip resolved per element -- instead of from the dense `padded` copy that
    // k_gather_patches made (one kernel and a 4.3 B / voxel round trip less per batch)
    const float* vol;
    const int* origins;  // [N][3]
    int V0, V1, V2, o0, o1, o2, flip;
    float* out32;        // X3: fp32 octet planes [N][4][voxel][8]
    float wscale, winv;  // X3: power-of-two scale of the split weights
};

// X3 (split-precision mode): the fp32 input is staged as hi / lo fp16 halo tiles, a K step covers 8 taps ([Wh | Wh] x [Xh ; Xl] +
// [Wl | Wl] x [Xh ; Xl], 4 steps = 8 MFMAs per 32 voxels), the epilogue stores fp32 octet planes (128 B per voxel: the kernel
// stays store-bound).
template <bool X3>
__global__ __launch_bounds__(256) void k_conv_first_mfma(FirstMfmaArgs p) {
    constexpr int H0 = MF0 + 2, H1 = MF1 + 2, H2 = MF2 + 2, HV = H0 * H1 * H2;
    __shared__ _Float16 halo[X3 ? 4 : 2][HV + 8];   // [buffer][X3: hi, lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    // A fragments: lane (cout = l31, kh) holds k = 8 kh + i (step 0) and 16 + 8 kh + i (step 1); taps >= 27 are zero
    f16x8 a0, a1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k0 = 8 * kh + i, k1 = 16 + 8 * kh + i;
        a0[i] = (_Float16)p.w[k0 * 32 + l31];
        a1[i] = k1 < 27 ? (_Float16)p.w[k1 * 32 + l31] : (_Float16)0.f;
    }
    f16x8 xah[X3 ? 4 : 1], xal[X3 ? 4 : 1];
    if constexpr (X3) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 8 * st + i;
                const float wv = k < 27 ? p.w[k * 32 + l31] * p.wscale : 0.f;
                const _Float16 h = (_Float16)wv;
                xah[st][i] = h;
                xal[st][i] = (_Float16)(wv - (float)h);
            }
    }
    // MFMA column (lane l31) <-> voxel lv of the 32-voxel row: even lanes take voxels 0-15, odd lanes 16-31, so that after the
    // register transpose the lane pair (2m, 2m + 1) can exchange one 16-byte piece and ONE store instruction writes the complete
    // 32-byte records of voxels 0-15 (the next one 16-31): whole 64-byte lines per instruction in this write-bound kernel
    // (column = voxel made every instruction write bytes [0, 16) or [16, 32) of all 32 records)
    const int lv = X3 ? l31 : (l31 >> 1) + ((l31 & 1) << 4);   // (X3: column = voxel, 16-byte stores of a half-wave are 1 KiB contiguous)
    const bool odd = (l31 & 1) != 0;
    // LDS offsets (in halves) of this lane's 16 taps relative to the M-tile's first halo voxel
    int toff[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int t = (i < 8 ? 8 * kh + i : 16 + 8 * kh + (i - 8));
        t = t < 27 ? t : 26;  // padded taps: any finite value (their weights are zero)
        toff[i] = ((t / 9) * H1 + (t / 3) % 3) * H2 + t % 3 + lv;
    }
    int xoff[X3 ? 32 : 1];   // X3: taps 8 st + i for both k-halves (the k-half selects the hi / lo tile)
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = i < 27 ? i : 26;
            xoff[i] = ((t / 9) * H1 + (t / 3) % 3) * H2 + t % 3 + lv;
        }
    }
    float4 bq[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bq[gq] = *(const float4*)(p.bias + 8 * gq + 4 * kh);
    float st_s[16], st_q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st_s[i] = st_q[i] = 0.f;
    int st_n = -1;
    auto flush = [&](int slot) {
        if (st_n < 0) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int mm = 1; mm < 32; mm <<= 1) {
                st_s[i] += __shfl_xor(st_s[i], mm);
                st_q[i] += __shfl_xor(st_q[i], mm);
            }
        }
        if (l31 == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 8 * (i >> 2) + 4 * kh + (i & 3);
                float* pp = p.partials + (((size_t)st_n * 32 + row) * 2) * p.nslots + slot;
                pp[0] = st_s[i];
                pp[p.nslots] = st_q[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) st_s[i] = st_q[i] = 0.f;
    };
    const int nsp = p.t0 * p.t1 * p.t2;
    const size_t pvol = (size_t)p.PX * p.PY * p.PZ, ovox = (size_t)p.P0 * p.P1 * p.P2;
    // Tile sequence: the spatial tiles of ONE sample are dealt out to vw = min(nsp, CUs) virtual workgroups (j takes
    // sp = j, j + vw, ...), whose partial sums go to statistics slot 4 j + wave of that sample -- a function of the sample
    // alone, not of the batch it shares the launch with (batch-invariant results, as in k_conv_ws).  Physical workgroup b
    // executes the virtual workgroups b, b + G, ...
    const int vw = p.vw, nvirt = p.N * vw;
    struct Seq {
        int v, n, j, sp;
    };
    auto seq_valid = [&](const Seq& q) { return q.v < nvirt; };
    auto seq_first = [&]() {
        Seq q;
        q.v = (int)blockIdx.x;
        q.n = q.v / vw;
        q.j = q.v - q.n * vw;
        q.sp = q.j;
        return q;
    };
    auto seq_next = [&](Seq q) {
        q.sp += vw;
        if (q.sp >= nsp) {
            q.v += (int)gridDim.x;
            q.n = q.v / vw;
            q.j = q.v - q.n * vw;
            q.sp = q.j;
        }
        return q;
    };
    // halo staging is split in two halves so that the global round trip of the NEXT tile overlaps this tile's compute:
    // fetch() issues the loads into registers, commit() converts and writes them to the other LDS buffer afterwards
    constexpr int NPRE = (HV + 255) / 256;
    float pre[NPRE];
    // this thread's halo voxels (tile independent): packed coordinates x | y << 8 | z << 16 and the voxel's linear offset in the
    // volume relative to the halo origin -- the per-tile gather is then three range tests and one add per element (the first
    // version decomposed the index and rebuilt a 64-bit address per element and tile: ~80 instructions each, as much as the
    // tile's MFMA + epilogue work)
    int hc[NPRE], hrel[NPRE];
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int i = min(tid + 256 * j, HV - 1);
        const int z = i % H2, r = i / H2, y = r % H1, x = r / H1;
        hc[j] = x | (y << 8) | (z << 16);
        hrel[j] = (x * p.V1 + y) * p.V2 + z;
    }
    auto fetch = [&](const Seq& q) {
        int sp = q.sp;
        const int tz = sp % p.t2;
        sp /= p.t2;
        const int ty = sp % p.t1, tx = sp / p.t1;
        if (p.vol && p.flip == 0) {
            // halo origin in patch coordinates (conv padding 1) and in the volume; valid halo range per axis: inside the patch
            // (conv / tile padding reads zero) and inside the volume (pad_nd_image zeros)
            const int bx = tx * MF0 - 1, by = ty * MF1 - 1, bz = tz * MF2 - 1;
            const int ox = p.origins[q.n * 3 + 0] - p.o0 + bx, oy = p.origins[q.n * 3 + 1] - p.o1 + by, oz = p.origins[q.n * 3 + 2] - p.o2 + bz;
            const int xl = max(-bx, -ox), xh = min(p.P0 - bx, p.V0 - ox);      // halo x valid iff xl <= x < xh
            const int yl = max(-by, -oy), yh = min(p.P1 - by, p.V1 - oy);
            const int zl = max(-bz, -oz), zh = min(p.P2 - bz, p.V2 - oz);
            const float* base = p.vol + ((ptrdiff_t)ox * p.V1 + oy) * p.V2 + oz;
#pragma unroll
            for (int j = 0; j < NPRE; ++j) {
                const int x = hc[j] & 255, y = (hc[j] >> 8) & 255, z = hc[j] >> 16;
                const bool ok = (unsigned)(x - xl) < (unsigned)max(xh - xl, 0) && (unsigned)(y - yl) < (unsigned)max(yh - yl, 0) &&
                                (unsigned)(z - zl) < (unsigned)max(zh - zl, 0);
                pre[j] = ok ? base[hrel[j]] : 0.f;
            }
            return;
        }
        if (p.vol) {
            // patch coordinates of the halo origin (conv padding 1) and the tile's position in the volume
            const int bx = tx * MF0 - 1, by = ty * MF1 - 1, bz = tz * MF2 - 1;
            const int ox = p.origins[q.n * 3 + 0] - p.o0, oy = p.origins[q.n * 3 + 1] - p.o1, oz = p.origins[q.n * 3 + 2] - p.o2;
#pragma unroll
            for (int j = 0; j < NPRE; ++j) {
                const int i = min(tid + 256 * j, HV - 1);
                const int z = i % H2, r = i / H2, y = r % H1, x = r / H1;
                const int px = bx + x, py = by + y, pz = bz + z;
                float v = 0.f;
                if ((unsigned)px < (unsigned)p.P0 && (unsigned)py < (unsigned)p.P1 && (unsigned)pz < (unsigned)p.P2) {
                    // test-time mirroring (predict_from_raw_data.py:541-557): the network sees torch.flip(tile, axes)
                    const int qx = (p.flip & 1) ? p.P0 - 1 - px : px, qy = (p.flip & 2) ? p.P1 - 1 - py : py,
                              qz = (p.flip & 4) ? p.P2 - 1 - pz : pz;
                    const int vx = ox + qx, vy = oy + qy, vz = oz + qz;
                    if ((unsigned)vx < (unsigned)p.V0 && (unsigned)vy < (unsigned)p.V1 && (unsigned)vz < (unsigned)p.V2)
                        v = p.vol[((size_t)vx * p.V1 + vy) * p.V2 + vz];
                }
                pre[j] = v;
            }
            return;
        }
        const float* src = p.padded + (size_t)q.n * pvol + ((size_t)(tx * MF0) * p.PY + ty * MF1) * p.PZ + tz * MF2;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = min(tid + 256 * j, HV - 1);
            const int z = i % H2, r = i / H2, y = r % H1, x = r / H1;
            pre[j] = src[((size_t)x * p.PY + y) * p.PZ + z];
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            if constexpr (X3) {
                if (i < HV) {
                    const _Float16 h = (_Float16)pre[j];
                    halo[2 * buf][i] = h;
                    halo[2 * buf + 1][i] = (_Float16)(pre[j] - (float)h);
                }
            } else {
                if (i < HV) halo[buf][i] = (_Float16)pre[j];
            }
        }
    };
    Seq cur = seq_first();
    if (seq_valid(cur)) {
        fetch(cur);
        commit(0);
    }
    __syncthreads();
    int st_v = -1, slot = 0;
    for (int it = 0; seq_valid(cur); ++it) {
        const int buf = it & 1;
        const Seq nxt = seq_next(cur);
        const bool more = seq_valid(nxt);
        if (more) fetch(nxt);
        const int n = cur.n;
        int sp = cur.sp;
        const int tz = sp % p.t2;
        sp /= p.t2;
        const int ty = sp % p.t1, tx = sp / p.t1;
        if (cur.v != st_v) {
            flush(slot);
            st_v = cur.v;
            st_n = n;
            slot = cur.j * 4 + wave;
        }
        const _Float16* hb = X3 ? halo[2 * buf + kh] : halo[buf];
#pragma unroll 2
        for (int r = 0; r < (MF0 * MF1) / 4; ++r) {
            const int row = wave * ((MF0 * MF1) / 4) + r;  // (x, y) row of the block tile
            const int x = row / MF1, y = row % MF1;
            const _Float16* hr = hb + (x * H1 + y) * H2;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc;
            if constexpr (X3) {
                acc = zero;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    f16x8 b;
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] = hr[xoff[8 * st + i]];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah[st], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xal[st], b, acc, 0, 0, 0);
                }
            } else {
                f16x8 b0, b1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    b0[i] = hr[toff[i]];
                    b1[i] = hr[toff[8 + i]];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, zero, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
            }
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                v[gq * 4 + 0] = (X3 ? acc[gq * 4 + 0] * p.winv : acc[gq * 4 + 0]) + bq[gq].x;
                v[gq * 4 + 1] = (X3 ? acc[gq * 4 + 1] * p.winv : acc[gq * 4 + 1]) + bq[gq].y;
                v[gq * 4 + 2] = (X3 ? acc[gq * 4 + 2] * p.winv : acc[gq * 4 + 2]) + bq[gq].z;
                v[gq * 4 + 3] = (X3 ? acc[gq * 4 + 3] * p.winv : acc[gq * 4 + 3]) + bq[gq].w;
            }
            // packed fp32 statistics (v_pk_add_f32 / v_pk_fma_f32: the same operations per entry, two entries per instruction)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                typedef float cf2 __attribute__((ext_vector_type(2)));
                const cf2 vv = cf2{v[2 * i], v[2 * i + 1]};
                cf2 s2 = cf2{st_s[2 * i], st_s[2 * i + 1]}, q2 = cf2{st_q[2 * i], st_q[2 * i + 1]};
                s2 = s2 + vv;
                q2 = __builtin_elementwise_fma(vv, vv, q2);
                st_s[2 * i] = s2.x; st_s[2 * i + 1] = s2.y;
                st_q[2 * i] = q2.x; st_q[2 * i + 1] = q2.y;
            }
            if constexpr (X3) {
                // fp32 octet planes [N][4][voxel][8]: entries 4 gq .. + 3 = couts 8 gq + 4 kh .. + 3 of voxel l31
                float* dst32 = p.out32 + ((size_t)n * 32 * ovox + (((size_t)(tx * MF0 + x) * p.P1 + ty * MF1 + y) * p.P2 + tz * MF2 + l31) * 8) + 4 * kh;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) *(float4*)(dst32 + (size_t)gq * 8 * ovox) = make_float4(v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]);
                continue;
            }
            unsigned w8[8];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float lo4[4], hi4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[pr * 4 + e]), __float_as_uint(v[(pr + 2) * 4 + e]), false, false);
                    lo4[e] = __uint_as_float(sw[0]);
                    hi4[e] = __uint_as_float(sw[1]);
                }
                // one v_cvt_pk_f16_f32 (RTNE, same rounding as __float2half_rn) per output word
                typedef float cvf2 __attribute__((ext_vector_type(2)));
                typedef _Float16 cvh2 __attribute__((ext_vector_type(2)));
                auto pk = [](float a, float b) {
                    union {
                        cvh2 v;
                        unsigned u;
                    } c;
                    c.v = __builtin_convertvector(cvf2{a, b}, cvh2);
                    return c.u;
                };
                w8[pr * 4 + 0] = pk(lo4[0], lo4[1]);
                w8[pr * 4 + 1] = pk(lo4[2], lo4[3]);
                w8[pr * 4 + 2] = pk(hi4[0], hi4[1]);
                w8[pr * 4 + 3] = pk(hi4[2], hi4[3]);
            }
            // chunk-planar [N][2][voxel][16]: lane pair (2m, 2m + 1) of plane kh writes voxel m's record, then voxel 16 + m's
            unsigned wa[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned n0 = (unsigned)__builtin_amdgcn_mov_dpp((int)w8[i], 0xB1, 0xF, 0xF, true);       // neighbour's piece 0
                const unsigned n1 = (unsigned)__builtin_amdgcn_mov_dpp((int)w8[4 + i], 0xB1, 0xF, 0xF, true);   // neighbour's piece 1
                wa[i] = odd ? n1 : w8[i];
                wb[i] = odd ? w8[4 + i] : n0;
            }
            __half* dst = p.out + ((size_t)(n * 2 + kh) * ovox + ((size_t)(tx * MF0 + x) * p.P1 + ty * MF1 + y) * p.P2 + tz * MF2 + (l31 >> 1)) * 16 + (odd ? 8 : 0);
            // (streaming `nt` stores were measured: 0 ... -10 %)
            *(uint4*)dst = make_uint4(wa[0], wa[1], wa[2], wa[3]);
            *(uint4*)(dst + 16 * 16) = make_uint4(wb[0], wb[1], wb[2], wb[3]);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        cur = nxt;
    }
    flush(slot);
}

static bool first_mfma_ok(int Cin, const int P[3], const int k[3], int Cout) {
    static const bool off = getenv("BOA_FIRST_MFMA") && atoi(getenv("BOA_FIRST_MFMA")) == 0;
    return !off && Cin == 1 && Cout == 32 && k[0] == 3 && k[1] == 3 && k[2] == 3 && P[0] % MF0 == 0 && P[1] % MF1 == 0 && P[2] % MF2 == 0;
}

int conv_first_nblk(const int P[3], int cu_count) {
    return std::max(ceil_div(P[0], FT0) * ceil_div(P[1], FT1) * ceil_div(P[2], FT2), cu_count * 4);
}

// dims of the zero-padded gather buffer for a patch P and kernel k
void conv_first_padded_dims(const int P[3], const int k[3], int out[3]) {
    const int ft[3] = {FT0, FT1, FT2};
    for (int a = 0; a < 3; ++a) out[a] = ceil_div(P[a], ft[a]) * ft[a] + (k[a] - 1);
}

int launch_conv_first(boa_ctx* ctx, const float* volume, const int V[3], const int vol_off[3], const int* dev_origins,
                      int N, int Cin, const int P[3], const int k[3], int Cout, const float* w, const float* bias,
                      float* padded_scratch, __half* out, float* partials, int* nblk_out, int flip_mask, float* out32) {
    BOA_REQUIRE(Cout % 32 == 0, "first conv: Cout=%d must be a multiple of 32", Cout);
    BOA_REQUIRE(Cin >= 1 && Cin <= 4, "first conv: Cin=%d unsupported (1..4)", Cin);
    const bool k333 = k[0] == 3 && k[1] == 3 && k[2] == 3, k133 = k[0] == 1 && k[1] == 3 && k[2] == 3;
    BOA_REQUIRE(k333 || k133, "first conv: kernel %dx%dx%d not instantiated", k[0], k[1], k[2]);
    int PD[3];
    conv_first_padded_dims(P, k, PD);
    const size_t pvol = (size_t)PD[0] * PD[1] * PD[2];
    const double vox = (double)N * P[0] * P[1] * P[2];
    KernelTimer tm(ctx, BOA_K_CONV_FIRST, 2.0 * vox * k[0] * k[1] * k[2] * Cin * Cout, vox * (4.0 * Cin + 2.0 * Cout));
    static const bool fuse_gather = !(getenv("BOA_FIRST_GATHER") && atoi(getenv("BOA_FIRST_GATHER")) == 1);  // 1: separate gather kernel
    const bool fused = fuse_gather && first_mfma_ok(Cin, P, k, Cout);
    if (!fused)
    hipLaunchKernelGGL(k_gather_patches, dim3((unsigned)((pvol + 255) / 256), Cin, N), dim3(256), 0, ctx->stream, volume,
                       dev_origins, V[0], V[1], V[2], vol_off ? vol_off[0] : 0, vol_off ? vol_off[1] : 0,
                       vol_off ? vol_off[2] : 0, Cin, P[0], P[1], P[2], (k[0] - 1) / 2, (k[1] - 1) / 2, (k[2] - 1) / 2, PD[0],
                       PD[1], PD[2], flip_mask, padded_scratch);
    const int nblk_tab = conv_first_nblk(P, ctx->cu_count);
    if (nblk_out) *nblk_out = nblk_tab;
    if (first_mfma_ok(Cin, P, k, Cout)) {
        FirstMfmaArgs m;
        m.out32 = out32;
        m.wscale = X3_HEAD_WSCALE;   // (first-conv weights are O(0.1 .. 1) like the head's: one fixed power of two)
        m.winv = 1.0f / X3_HEAD_WSCALE;
        m.padded = padded_scratch; m.PX = PD[0]; m.PY = PD[1]; m.PZ = PD[2];
        m.vol = fused ? volume : nullptr; m.origins = dev_origins; m.flip = flip_mask;
        m.V0 = V[0]; m.V1 = V[1]; m.V2 = V[2];
        m.o0 = vol_off ? vol_off[0] : 0; m.o1 = vol_off ? vol_off[1] : 0; m.o2 = vol_off ? vol_off[2] : 0;
        m.N = N; m.P0 = P[0]; m.P1 = P[1]; m.P2 = P[2];
        m.w = w; m.bias = bias; m.out = out; m.partials = partials; m.nslots = nblk_tab;
        m.t0 = P[0] / MF0; m.t1 = P[1] / MF1; m.t2 = P[2] / MF2;
        m.vw = std::min(m.t0 * m.t1 * m.t2, ctx->cu_count);
        // (physical workgroup b runs the virtual workgroups b, b + G, ...: any G gives the same results; more than one workgroup
        //  per CU hides the halo gather's and the stores' latency -- the kernel is a 3.4 GB write per 25 tiles)
        static const int grid_mult = getenv("BOA_FIRST_GRID") ? std::max(1, atoi(getenv("BOA_FIRST_GRID"))) : 4;
        const dim3 fgrid((unsigned)std::min<long long>((long long)m.vw * N, (long long)ctx->cu_count * grid_mult));
        if (out32) {
            hipLaunchKernelGGL(k_conv_first_mfma<true>, fgrid, dim3(256), 0, ctx->stream, m);
            ctx->counters[BOA_CNT_X3]++;
        } else {
            hipLaunchKernelGGL(k_conv_first_mfma<false>, fgrid, dim3(256), 0, ctx->stream, m);
            ctx->counters[BOA_CNT_FIRST_MFMA]++;
        }
        tm.stop();
        BOA_HIP_TRY(hipGetLastError());
        return BOA_OK;
    }
    FirstArgs a;
    a.nblk = nblk_tab;
    a.padded = padded_scratch; a.PX = PD[0]; a.PY = PD[1]; a.PZ = PD[2];
    a.N = N; a.Cin = Cin; a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2]; a.Cout = Cout;
    a.w = w; a.bias = bias; a.out = out; a.partials = partials; a.out32 = out32;
    a.t0 = ceil_div(P[0], FT0); a.t1 = ceil_div(P[1], FT1); a.t2 = ceil_div(P[2], FT2);
    const int nblk = a.t0 * a.t1 * a.t2;
    const int HV = (FT0 + k[0] - 1) * (FT1 + k[1] - 1) * (FT2 + k[2] - 1);
    const size_t lds = ((size_t)Cin * k[0] * k[1] * k[2] * 32 + (((size_t)Cin * HV + 3) & ~(size_t)3)) * 4 + 1024;
    if (out32) {
        if (k333)
            hipLaunchKernelGGL((k_conv_first<3, 3, 3, true>), dim3(nblk, Cout / 32, N), dim3(256), lds, ctx->stream, a);
        else
            hipLaunchKernelGGL((k_conv_first<1, 3, 3, true>), dim3(nblk, Cout / 32, N), dim3(256), lds, ctx->stream, a);
        ctx->counters[BOA_CNT_X3]++;
    } else {
        if (k333)
            hipLaunchKernelGGL((k_conv_first<3, 3, 3, false>), dim3(nblk, Cout / 32, N), dim3(256), lds, ctx->stream, a);
        else
            hipLaunchKernelGGL((k_conv_first<1, 3, 3, false>), dim3(nblk, Cout / 32, N), dim3(256), lds, ctx->stream, a);
        ctx->counters[BOA_CNT_FIRST_VALU]++;
    }
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// InstanceNorm finalize: deterministic fp64 reduction of the per-block partials
// T = 256: one block per (n, c).  T = 64 (layers with at most 64 slots per (n, c): the 8^3 / 4^3 layers, whose N x 320 blocks of 256 mostly
// idle threads took four rounds of dependent HBM round trips to get through the CUs): one wave per (n, c) -- the same additions in the same
// order (with <= 64 slots only wave 0 of the 256-thread form holds non-zero terms), so the same bits.
template <int T>
__global__ __launch_bounds__(T) void k_norm_finalize(float* __restrict__ partials, int nblk, int C, double count, int clear,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, float* __restrict__ ss,
                                                    unsigned short* __restrict__ ss16) {
    const int c = blockIdx.x, n = blockIdx.y;
    float* ps = partials + (((size_t)n * C + c) * 2 + 0) * nblk;
    float* pq = partials + (((size_t)n * C + c) * 2 + 1) * nblk;
    __shared__ double red[8];
    // (loaded up front: behind the reduction they were one more dependent round trip per block)
    const float gam = gamma[c], bet = beta[c];
    double s = 0.0, q = 0.0;
    // (four slots per thread in flight: the loop was a chain of dependent global round trips -- 16 for the 4 096 slots of a 128^3
    //  layer; the additions keep their order)
    int i = threadIdx.x;
    for (; i + 768 < nblk; i += 1024) {
        const float a0 = ps[i], a1 = ps[i + 256], a2 = ps[i + 512], a3 = ps[i + 768];
        const float b0 = pq[i], b1 = pq[i + 256], b2 = pq[i + 512], b3 = pq[i + 768];
        s += (double)a0; q += (double)b0;
        s += (double)a1; q += (double)b1;
        s += (double)a2; q += (double)b2;
        s += (double)a3; q += (double)b3;
        if (clear) {  // k_conv_ws only writes the slots of waves that worked on (n, c): leave the table zeroed for the next launch
            ps[i] = 0.f; ps[i + 256] = 0.f; ps[i + 512] = 0.f; ps[i + 768] = 0.f;
            pq[i] = 0.f; pq[i + 256] = 0.f; pq[i + 512] = 0.f; pq[i + 768] = 0.f;
        }
    }
    for (; i < nblk; i += 256) {
        s += (double)ps[i];
        q += (double)pq[i];
        if (clear) {
            ps[i] = 0.f;
            pq[i] = 0.f;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m);
        q += __shfl_xor(q, m);
    }
    if (T > 64) {
        if ((threadIdx.x & 63) == 0) {
            red[(threadIdx.x >> 6) * 2] = s;
            red[(threadIdx.x >> 6) * 2 + 1] = q;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (T > 64) {
            s = (red[0] + red[2]) + (red[4] + red[6]);  // fixed order: deterministic
            q = (red[1] + red[3]) + (red[5] + red[7]);
        }
        double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        double inv = 1.0 / sqrt(var + (double)eps);
        float scale = (float)((double)gam * inv);
        float shift = (float)((double)bet - mean * (double)gam * inv);
        ss[((size_t)n * C + c) * 2 + 0] = scale;
        ss[((size_t)n * C + c) * 2 + 1] = shift;
        if (ss16) {  // packed fp16 copy for k_conv_ws: per channel pair {s_c, s_c+1, t_c, t_c+1}
            unsigned short* q16 = ss16 + ((size_t)n * C + (c & ~1)) * 2;
            q16[c & 1] = f2us(scale);
            q16[2 + (c & 1)] = f2us(shift);
        }
    }
}

int launch_norm_finalize(boa_ctx* ctx, float* partials, int nblk, int N, int C, double count,
                         const float* gamma, const float* beta, float eps, float* ss_out, unsigned* ss16_out, int clear) {
    KernelTimer tm(ctx, BOA_K_NORM_FINALIZE, 0, (double)N * C * nblk * 8.0);
    if (nblk <= 64)
        hipLaunchKernelGGL(k_norm_finalize<64>, dim3(C, N), dim3(64), 0, ctx->stream, partials, nblk, C, count, clear, gamma,
                           beta, eps, ss_out, (unsigned short*)ss16_out);
    else
        hipLaunchKernelGGL(k_norm_finalize<256>, dim3(C, N), dim3(256), 0, ctx->stream, partials, nblk, C, count, clear, gamma,
                           beta, eps, ss_out, (unsigned short*)ss16_out);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// transposed conv, kernel == stride: out[o] = sum_ci y[o / s][ci] * W[ci][co][o % s] + b
// ---- the transposed convs' per-wave LDS slab (D fragments -> 16-byte pieces of the output voxels' records) -------------------------
// Logical layout [plane][output voxel ov = l31 * TZ + t][16 couts]: lane (l31, kh) writes the 8-byte piece q = 2 (gq & 1) + kh of its
// voxel's 32-byte record, the wave then reads 16-byte pieces `lane + 64 k` and stores 1 KiB runs.  With TZ = 2 the writing lanes sit
// 64 bytes apart: every 16-lane group of the ds_write_b64 fell on two banks' worth of one 128-byte row -- 8-way conflicts, 72 - 76 % of
// the kernels' LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_pmc_lds.txt).  Physical layout for TZ = 2: record
// (t, l31) at t * 32 + (l31 ^ 4 t) (lanes 32 bytes apart; the xor keeps the two taps of a voxel pair off the same bank row for the
// reads), piece q at q ^ ((l31 >> 2) & 3) (the four lanes of a group that share a 32-byte window take its four pieces): writes and
// reads are conflict-free; a reader whose record has an odd swizzle finds the two 8-byte pieces of its half swapped and swaps them back.
template <int TZ>
__device__ __forceinline__ int convt_slab_waddr(int gq, int l31, int t, int kh) {
    if constexpr (TZ == 2) {
        const int rec = t * 32 + (l31 ^ (4 * t));
        const int q = ((gq & 1) * 2 + kh) ^ ((l31 >> 2) & 3);
        return ((gq >> 1) * 64 + rec) * 32 + q * 8;
    } else {
        return ((gq >> 1) * 32 * TZ + l31 * TZ + t) * 32 + (8 * (gq & 1) + 4 * kh) * 2;
    }
}
template <int TZ>
__device__ __forceinline__ uint4 convt_slab_read(const unsigned char* slab, int pl, int piece) {
    if constexpr (TZ == 2) {
        const int ov = piece >> 1, h = piece & 1;
        const int j = ov >> 1, tz = ov & 1;
        const int sw = (j >> 2) & 3;
        const uint4 d = *(const uint4*)(slab + (pl * 64 + tz * 32 + (j ^ (4 * tz))) * 32 + (h ^ (sw >> 1)) * 16);
        return (sw & 1) ? make_uint4(d.z, d.w, d.x, d.y) : d;
    } else {
        return *(const uint4*)(slab + pl * 32 * TZ * 32 + piece * 16);
    }
}

struct ConvTArgs {
    const __half* src;
    const float* ss;
    const unsigned* ss16;  // packed fp16 (scale, shift) pairs of the input's deferred norm (preferred), or NULL
    int Cin, Cout, N, Di, Hi, Wi, s0, s1, s2;
    const __half* wpk;  // [tap][Cin/16][2][Cout][8]
    const float* bias;
    __half* out;
    float slope;
};

// One wave = 32 consecutive (flattened) input voxels.  Per (tx, ty, cout chunk) it computes BOTH z taps (TZ = s2
// accumulators): in the chunk-planar output the voxels 2 iz and 2 iz + 1 of a row are neighbours, so the wave's result for
// one 16-cout plane is one run of 2 KiB (TZ = 2) of consecutive bytes.  The D fragments go through a per-wave LDS slab
// [2 planes][32 TZ voxels][16 couts] and leave as 16-byte pieces, lane L taking pieces L, L + 64, ...: every store
// instruction writes 1 KiB of consecutive bytes (the one-tap-per-pass form wrote 32-byte pieces 64 bytes apart).
typedef _Float16 ct_h2 __attribute__((ext_vector_type(2)));

// deferred InstanceNorm + LeakyReLU on 8 channels in packed fp16 (the same one-rounding evaluation as k_conv_ws's producers):
// w = 4 x {packed scales, packed shifts} of the four channel pairs
__device__ __forceinline__ uint4 convt_norm_act8_pk(uint4 raw, const uint4& w0, const uint4& w1, unsigned slope2) {
    union {
        uint4 u;
        ct_h2 v[4];
    } x;
    union {
        unsigned u;
        ct_h2 v;
    } s, t, sl;
    const unsigned w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    x.u = raw;
    sl.u = slope2;
    ct_h2 y[4], z[4];
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

template <int TZ>
__global__ __launch_bounds__(256) void k_convt_mfma(ConvTArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    const int ncc = p.Cin / 16;
    // (32-bit index arithmetic: N * in_vox < 2^31, checked on the host -- the 64-bit divisions of the first version were
    //  ~40 % of the kernel's instructions; the kernel is instruction-bound: with every memory access switched off it still
    //  took half of its time)
    const unsigned in_vox = (unsigned)(p.Di * p.Hi * p.Wi);
    const unsigned total = (unsigned)p.N * in_vox;
    constexpr int SLAB = 2 * 32 * TZ * 32;  // bytes: [2 planes][32 * TZ output voxels][16 halves]
    unsigned char* lds = smem + (size_t)wave * (ncc * 1024 + SLAB);  // [cc][khalf][32 voxels][8 halves] + slab
    unsigned char* slab = lds + ncc * 1024;
    const unsigned g0 = ((unsigned)blockIdx.x * 4 + wave) * 32;  // first flattened (n, voxel) of this wave
    const unsigned g = g0 + l31;
    const bool valid = g < total;
    const unsigned n = valid ? g / in_vox : 0;
    const unsigned vi = valid ? g - n * in_vox : 0;
    // stage this wave's 32 voxels: lane (l31, kh) moves octet kh of every 16-channel chunk; loads batched by 4
    {
        union {
            unsigned u;
            ct_h2 v;
        } sl2;
        sl2.v = ct_h2{(_Float16)p.slope, (_Float16)p.slope};
        const __half* src_l = p.src + ((size_t)n * ncc * in_vox + vi) * 16 + kh * 8;   // + cc * in_vox * 16 per chunk
        const unsigned* ss16_l = p.ss16 ? p.ss16 + ((size_t)n * p.Cin + kh * 8) : nullptr;  // + cc * 16 words per chunk
        for (int c0 = 0; c0 < ncc; c0 += 4) {
            uint4 val[4], w0[4], w1[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int cc = min(c0 + b, ncc - 1);
                val[b] = *(const uint4*)(src_l + (size_t)cc * in_vox * 16);  // chunk-planar
                if (ss16_l) {
                    w0[b] = *(const uint4*)(ss16_l + cc * 16);
                    w1[b] = *(const uint4*)(ss16_l + cc * 16 + 4);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int cc = min(c0 + b, ncc - 1);
                uint4 o = val[b];
                if (ss16_l) {
                    o = convt_norm_act8_pk(o, w0[b], w1[b], sl2.u);
                } else if (p.ss) {  // (callers without the packed table: fp32 evaluation)
                    float sc[8], sh[8];
                    const float* ss = p.ss + ((size_t)n * p.Cin + cc * 16 + kh * 8) * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        sc[j] = ss[2 * j];
                        sh[j] = ss[2 * j + 1];
                    }
                    o = norm_act8(o, sc, sh, p.slope);
                }
                if (!valid) o = make_uint4(0, 0, 0, 0);
                *(uint4*)(lds + ((cc * 2 + kh) * 32 + l31) * 16) = o;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int Ho = p.Hi * p.s1, Wo = p.Wi * p.s2;
    const size_t ovox = (size_t)(p.Di * p.s0) * Ho * Wo;
    const int nco = p.Cout / 32;
    const int npairs = p.s0 * p.s1 * nco;  // (tx, ty, cout chunk); every pair covers the TZ z taps
    // store side: the slab of one plane holds 32 * TZ output voxels = 64 * TZ pieces of 16 bytes; this lane takes pieces
    // lane + 64 k (k < TZ) of each plane: output voxel ov = piece / 2 -> input voxel j = ov / TZ, z tap ov % TZ.
    // optr[k]: this lane's piece in plane 0 of the sample at tap (0, 0); a pass adds ((co * 2 + pl) * ovox + toff) * 32 bytes
    unsigned char* optr[TZ];
    bool ovalid[TZ];
#pragma unroll
    for (int k = 0; k < TZ; ++k) {
        const int ov = (lane + 64 * k) >> 1;
        const int j = ov / TZ, tz = ov % TZ;
        const unsigned gg = g0 + j;
        ovalid[k] = gg < total;
        const unsigned nn = ovalid[k] ? gg / in_vox : 0;
        const unsigned v2 = ovalid[k] ? gg - nn * in_vox : 0;
        const unsigned r2 = v2 / (unsigned)p.Wi;
        const int iz = (int)(v2 - r2 * (unsigned)p.Wi);
        const int ix = (int)(r2 / (unsigned)p.Hi), iy = (int)(r2 - (unsigned)ix * (unsigned)p.Hi);
        const size_t ospat = ((size_t)(ix * p.s0) * Ho + (size_t)(iy * p.s1)) * Wo + (size_t)(iz * p.s2 + tz);
        optr[k] = (unsigned char*)p.out + ((size_t)nn * (p.Cout / 16) * ovox + ospat) * 32 + 16 * (lane & 1);
    }
    // weights: wave-uniform part of the address per (tap, chunk, cout chunk) + this lane's (kh, cout) offset
    const unsigned char* wbase = (const unsigned char*)p.wpk;
    const unsigned wlane = ((unsigned)kh * (unsigned)p.Cout + (unsigned)l31) * 16u;
    const unsigned wstep_cc = 2u * (unsigned)p.Cout * 16u;  // bytes per (tap, chunk)
    const unsigned char* bfrag = lds + (kh * 32 + l31) * 16;  // + cc * 1024
    int bias_co = -1;
    f32x16 biasv;
#pragma unroll
    for (int i = 0; i < 16; ++i) biasv[i] = 0.f;
    for (int pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
        const int txy = pr / nco, co = pr - txy * nco;
        const int ty = txy % p.s1, tx = txy / p.s1;
        if (co != bias_co) {  // this lane's 16 biases of the cout chunk (entry 4 gq + e <-> cout 8 gq + 4 kh + e)
            bias_co = co;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bq = *(const float4*)(p.bias + co * 32 + 8 * gq + 4 * kh);
                biasv[gq * 4 + 0] = bq.x; biasv[gq * 4 + 1] = bq.y; biasv[gq * 4 + 2] = bq.z; biasv[gq * 4 + 3] = bq.w;
            }
        }
        f32x16 acc[TZ];
        const int tap0 = (tx * p.s1 + ty) * p.s2;
        const unsigned char* wpass = wbase + ((size_t)tap0 * ncc * wstep_cc + (size_t)co * 32 * 16);  // uniform
        // groups of 4 chunks: the group's TZ x 4 weight fragments are loaded as one batch (the thin deep layers wait on L2 for
        // them: 8 loads in flight per wave), then 4 x TZ MFMAs.  The very first MFMA takes an inline-zero C operand instead
        // of zeroed accumulators.
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto group = [&](int c0, bool first) {
            f16x8 a[TZ][4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int t = 0; t < TZ; ++t)
                    a[t][b] = *(const f16x8*)(wpass + (size_t)(t * ncc + min(c0 + b, ncc - 1)) * wstep_cc + wlane);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b == 0 || c0 + b < ncc) {  // (wave-uniform)
                    const f16x8 bf = *(const f16x8*)(bfrag + (c0 + b) * 1024);
#pragma unroll
                    for (int t = 0; t < TZ; ++t)
                        acc[t] = (first && b == 0) ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][b], bf, zero, 0, 0, 0)
                                                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][b], bf, acc[t], 0, 0, 0);
                }
            }
        };
        group(0, true);
        for (int c0 = 4; c0 < ncc; c0 += 4) group(c0, false);
        // fp16, into the slab: lane (voxel l31, kh) holds couts 8 gq + 4 kh + e -> plane gq / 2, offset 8 (gq % 2) + 4 kh
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
            for (int t = 0; t < TZ; ++t) {
                union {
                    uint2 u;
                    __half h[4];
                } pk;
                pk.h[0] = __float2half_rn(acc[t][gq * 4 + 0] + biasv[gq * 4 + 0]);
                pk.h[1] = __float2half_rn(acc[t][gq * 4 + 1] + biasv[gq * 4 + 1]);
                pk.h[2] = __float2half_rn(acc[t][gq * 4 + 2] + biasv[gq * 4 + 2]);
                pk.h[3] = __float2half_rn(acc[t][gq * 4 + 3] + biasv[gq * 4 + 3]);
                *(uint2*)(slab + convt_slab_waddr<TZ>(gq, l31, t, kh)) = pk.u;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const size_t poff = ((size_t)(co * 2) * ovox + ((size_t)tx * Ho + ty) * Wo) * 32;  // uniform; plane pl adds ovox * 32
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int k = 0; k < TZ; ++k) {
                const int piece = lane + 64 * k;
                const uint4 d = convt_slab_read<TZ>(slab, pl, piece);
                if (ovalid[k]) *(uint4*)(optr[k] + poff + (size_t)pl * ovox * 32) = d;
            }
        __builtin_amdgcn_wave_barrier();
    }
}

// Register-weights variant for Cin = 16 NCC <= 128 (the 32^3 -> 64^3 and 64^3 -> 128^3 transposed convs, 75 % of the class's
// time): a wave covers G groups of 32 input voxels, staged once into its LDS slice, and per (x tap, y tap, cout chunk) pass loads
// the pass's TZ x NCC weight fragments ONCE into registers for all G groups.  k_convt_mfma re-reads them from L2 for every 32
// voxels: 1 KiB of weights per input voxel of the 64 -> 32 layer against 640 bytes of activations moved -- the kernel was bound
// by L2 -> CU weight traffic, not by HBM.  Same arithmetic, same output order.
template <int TZ, int NCC, int G>
__global__ __launch_bounds__(256) void k_convt_mfma_rw(ConvTArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    const unsigned in_vox = (unsigned)(p.Di * p.Hi * p.Wi);
    const unsigned total = (unsigned)p.N * in_vox;
    constexpr int SLAB = 2 * 32 * TZ * 32;  // bytes: [2 planes][32 * TZ output voxels][16 halves]
    unsigned char* lds = smem + (size_t)wave * (G * NCC * 1024 + SLAB);  // [g][cc][khalf][32 voxels][8 halves] + slab
    unsigned char* slab = lds + G * NCC * 1024;
    const unsigned g0 = ((unsigned)blockIdx.x * 4 + wave) * (32 * G);  // first flattened (n, voxel) of this wave
    union {
        unsigned u;
        ct_h2 v;
    } sl2;
    sl2.v = ct_h2{(_Float16)p.slope, (_Float16)p.slope};
    // stage the wave's G x 32 voxels with the deferred norm applied
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const unsigned gv = g0 + 32 * g + l31;
        const bool valid = gv < total;
        const unsigned n = valid ? gv / in_vox : 0;
        const unsigned vi = valid ? gv - n * in_vox : 0;
        const __half* src_l = p.src + ((size_t)n * NCC * in_vox + vi) * 16 + kh * 8;
        const unsigned* ss16_l = p.ss16 ? p.ss16 + ((size_t)n * p.Cin + kh * 8) : nullptr;
        uint4 val[NCC];
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) val[cc] = *(const uint4*)(src_l + (size_t)cc * in_vox * 16);
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            uint4 o = val[cc];
            if (ss16_l) {
                const uint4 w0 = *(const uint4*)(ss16_l + cc * 16), w1 = *(const uint4*)(ss16_l + cc * 16 + 4);
                o = convt_norm_act8_pk(o, w0, w1, sl2.u);
            }
            if (!valid) o = make_uint4(0, 0, 0, 0);
            *(uint4*)(lds + (((g * NCC + cc) * 2 + kh) * 32 + l31) * 16) = o;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int Ho = p.Hi * p.s1, Wo = p.Wi * p.s2;
    const size_t ovox = (size_t)(p.Di * p.s0) * Ho * Wo;
    const int nco = p.Cout / 32;
    const int npairs = p.s0 * p.s1 * nco;
    // store side (per group): this lane's pieces lane + 64 k of each plane of the slab (see k_convt_mfma)
    unsigned char* optr[G][TZ];
    unsigned ovalid = 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < TZ; ++k) {
            const int ov = (lane + 64 * k) >> 1;
            const int j = ov / TZ, tz = ov % TZ;
            const unsigned gg = g0 + 32 * g + j;
            const bool ok = gg < total;
            ovalid |= ok ? (1u << (g * TZ + k)) : 0u;
            const unsigned nn = ok ? gg / in_vox : 0;
            const unsigned v2 = ok ? gg - nn * in_vox : 0;
            const unsigned r2 = v2 / (unsigned)p.Wi;
            const int iz = (int)(v2 - r2 * (unsigned)p.Wi);
            const int ix = (int)(r2 / (unsigned)p.Hi), iy = (int)(r2 - (unsigned)ix * (unsigned)p.Hi);
            const size_t ospat = ((size_t)(ix * p.s0) * Ho + (size_t)(iy * p.s1)) * Wo + (size_t)(iz * p.s2 + tz);
            optr[g][k] = (unsigned char*)p.out + ((size_t)nn * (p.Cout / 16) * ovox + ospat) * 32 + 16 * (lane & 1);
        }
    const unsigned char* wbase = (const unsigned char*)p.wpk;
    const unsigned wlane = ((unsigned)kh * (unsigned)p.Cout + (unsigned)l31) * 16u;
    const unsigned wstep_cc = 2u * (unsigned)p.Cout * 16u;  // bytes per (tap, chunk)
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
        const int txy = pr / nco, co = pr - txy * nco;
        const int ty = txy % p.s1, tx = txy / p.s1;
        float4 bq[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bq[gq] = *(const float4*)(p.bias + co * 32 + 8 * gq + 4 * kh);
        const int tap0 = (tx * p.s1 + ty) * p.s2;
        const unsigned char* wpass = wbase + ((size_t)tap0 * NCC * wstep_cc + (size_t)co * 32 * 16);  // uniform
        f16x8 a[TZ][NCC];   // the pass's weights, once for all G groups
#pragma unroll
        for (int t = 0; t < TZ; ++t)
#pragma unroll
            for (int cc = 0; cc < NCC; ++cc) a[t][cc] = *(const f16x8*)(wpass + (size_t)(t * NCC + cc) * wstep_cc + wlane);
        const size_t poff = ((size_t)(co * 2) * ovox + ((size_t)tx * Ho + ty) * Wo) * 32;  // uniform; plane pl adds ovox * 32
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x16 acc[TZ];
            const unsigned char* bfrag = lds + ((g * NCC * 2 + kh) * 32 + l31) * 16;  // + cc * 1024
#pragma unroll
            for (int cc = 0; cc < NCC; ++cc) {
                const f16x8 bf = *(const f16x8*)(bfrag + cc * 1024);
#pragma unroll
                for (int t = 0; t < TZ; ++t)
                    acc[t] = cc == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][cc], bf, zero, 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][cc], bf, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                for (int t = 0; t < TZ; ++t) {
                    union {
                        uint2 u;
                        __half h[4];
                    } pk;
                    pk.h[0] = __float2half_rn(acc[t][gq * 4 + 0] + bq[gq].x);
                    pk.h[1] = __float2half_rn(acc[t][gq * 4 + 1] + bq[gq].y);
                    pk.h[2] = __float2half_rn(acc[t][gq * 4 + 2] + bq[gq].z);
                    pk.h[3] = __float2half_rn(acc[t][gq * 4 + 3] + bq[gq].w);
                    *(uint2*)(slab + convt_slab_waddr<TZ>(gq, l31, t, kh)) = pk.u;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int k = 0; k < TZ; ++k) {
                    const int piece = lane + 64 * k;
                    const uint4 d = convt_slab_read<TZ>(slab, pl, piece);
                    if ((ovalid >> (g * TZ + k)) & 1u) *(uint4*)(optr[g][k] + poff + (size_t)pl * ovox * 32) = d;
                }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Deep transposed convs (Cin = 16 NCC >= 256: 4^3 ... 16^3 inputs, round 4).  These layers are not HBM-bound at all -- a pass's
// weights (2 NCC KiB per (x tap, y tap, cout chunk)) outweigh the activations, and k_convt_mfma streams them from L2 once per WAVE:
// 230 / 138 / 41 us per 25 tiles for 84 / 19 / 3 MB of tensor traffic.  Here the block shares them: a wave keeps the B fragments of
// its MT x 32 input voxels (deferred norm applied) in REGISTERS for the whole kernel (one wave per SIMD: 512 VGPRs), the pass's
// weight fragments are moved L2 -> LDS once per BLOCK by LDS-DMA (double-buffered: the next pass's weights arrive under this
// pass's MFMAs) and every wave reads its A fragments from LDS: 0.5 KiB of LDS reads per MFMA, no weight traffic per wave.
// Same arithmetic as k_convt_mfma (chunk order, fp32 accumulation, bias add, RTNE to fp16), same slab interleave for the stores.
template <int NCC>
__global__ __launch_bounds__(256) void k_convt_deep(ConvTArgs p) {
    constexpr int TZ = 2, MT = 2;
    constexpr int WB = TZ * NCC * 1024;          // bytes of one pass's weights in LDS
    constexpr int SLAB = 2 * 32 * TZ * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [weights 0][weights 1][4 slabs]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned in_vox = (unsigned)(p.Di * p.Hi * p.Wi);
    const unsigned total = (unsigned)p.N * in_vox;
    unsigned char* slab = smem + 2 * WB + wave * SLAB;
    float* lbias = (float*)(smem + 2 * WB + 4 * SLAB);   // the bias vector in LDS: a global load inside the pass loop would make hipcc wait
    for (int i = tid; i < p.Cout; i += 256) lbias[i] = p.bias[i];   // with vmcnt(0) -- i.e. also for the next pass's weight DMA
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const int Ho = p.Hi * p.s1, Wo = p.Wi * p.s2;
    const size_t ovox = (size_t)(p.Di * p.s0) * Ho * Wo;
    const int nco = p.Cout / 32;
    const int npairs = p.s0 * p.s1 * nco;
    // weights of pass `pr` -> LDS buffer `buf`: fragment f = tz * NCC + cc is one wave-wide LDS-DMA (64 lanes x 16 B: k-half
    // lane / 32, cout lane % 32); the four waves take f = wave, wave + 4, ...
    const unsigned wvoff = ((unsigned)kh * (unsigned)p.Cout + (unsigned)l31) * 16u;
    auto dma_pass = [&](int pr, int buf) {
        const int txy = pr / nco, co = pr - txy * nco;
        const int ty = txy % p.s1, tx = txy / p.s1;
        const int tap0 = (tx * p.s1 + ty) * p.s2;
        for (int f = wave; f < TZ * NCC; f += 4) {
            const int t = f / NCC, cc = f - t * NCC;
            const size_t woff = ((size_t)((tap0 + t) * NCC + cc) * 2 * p.Cout + (size_t)co * 32) * 16;
            // (wave-uniform by construction; readfirstlane makes it so for the compiler: the SGPR operands of the DMA)
            const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)woff), whi = __builtin_amdgcn_readfirstlane((unsigned)(woff >> 32));
            const unsigned char* src = (const unsigned char*)p.wpk + (((size_t)whi << 32) | wlo);
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_w + (unsigned)(buf * WB + f * 1024));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wvoff), "s"(src), "s"(m0v) : "memory");
        }
    };
    int pr = blockIdx.y;
    if (pr < npairs) dma_pass(pr, 0);
    // this wave's B fragments (registers) and store pointers
    union {
        unsigned u;
        ct_h2 v;
    } sl2;
    sl2.v = ct_h2{(_Float16)p.slope, (_Float16)p.slope};
    const unsigned g0 = ((unsigned)blockIdx.x * 4 + wave) * (32 * MT);
    f16x8 b[MT][NCC];
    unsigned char* optr[MT][TZ];
    unsigned ovalid = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const unsigned gv = g0 + 32 * m + l31;
        const bool valid = gv < total;
        const unsigned n = valid ? gv / in_vox : 0;
        const unsigned vi = valid ? gv - n * in_vox : 0;
        const __half* src_l = p.src + ((size_t)n * NCC * in_vox + vi) * 16 + kh * 8;
        const unsigned* ss16_l = p.ss16 + ((size_t)n * p.Cin + kh * 8);
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            uint4 o = *(const uint4*)(src_l + (size_t)cc * in_vox * 16);
            const uint4 w0 = *(const uint4*)(ss16_l + cc * 16), w1 = *(const uint4*)(ss16_l + cc * 16 + 4);
            o = convt_norm_act8_pk(o, w0, w1, sl2.u);
            if (!valid) o = make_uint4(0, 0, 0, 0);
            union {
                uint4 u;
                f16x8 f;
            } cv;
            cv.u = o;
            b[m][cc] = cv.f;
        }
#pragma unroll
        for (int k = 0; k < TZ; ++k) {
            const int ov = (lane + 64 * k) >> 1;
            const int j = ov / TZ, tz = ov % TZ;
            const unsigned gg = g0 + 32 * m + j;
            const bool ok = gg < total;
            ovalid |= ok ? (1u << (m * TZ + k)) : 0u;
            const unsigned nn = ok ? gg / in_vox : 0;
            const unsigned v2 = ok ? gg - nn * in_vox : 0;
            const unsigned r2 = v2 / (unsigned)p.Wi;
            const int iz = (int)(v2 - r2 * (unsigned)p.Wi);
            const int ix = (int)(r2 / (unsigned)p.Hi), iy = (int)(r2 - (unsigned)ix * (unsigned)p.Hi);
            const size_t ospat = ((size_t)(ix * p.s0) * Ho + (size_t)(iy * p.s1)) * Wo + (size_t)(iz * p.s2 + tz);
            optr[m][k] = (unsigned char*)p.out + ((size_t)nn * (p.Cout / 16) * ovox + ospat) * 32 + 16 * (lane & 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; pr < npairs; pr += gridDim.y, ++it) {
        const int buf = it & 1;
        if (pr + (int)gridDim.y < npairs) dma_pass(pr + (int)gridDim.y, buf ^ 1);
        const int txy = pr / nco, co = pr - txy * nco;
        const int ty = txy % p.s1, tx = txy / p.s1;
        float4 bq[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bq[gq] = *(const float4*)(lbias + co * 32 + 8 * gq + 4 * kh);
        const unsigned char* wl = smem + buf * WB + (kh * 32 + l31) * 16;
        f32x16 acc[TZ][MT];
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
            for (int t = 0; t < TZ; ++t) {
                const f16x8 a = *(const f16x8*)(wl + (t * NCC + cc) * 1024);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[t][m] = cc == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[m][cc], zero, 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[m][cc], acc[t][m], 0, 0, 0);
            }
        const size_t poff = ((size_t)(co * 2) * ovox + ((size_t)tx * Ho + ty) * Wo) * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int t = 0; t < TZ; ++t) {
                    union {
                        uint2 u;
                        __half h[4];
                    } pk;
                    pk.h[0] = __float2half_rn(acc[t][m][gq * 4 + 0] + bq[gq].x);
                    pk.h[1] = __float2half_rn(acc[t][m][gq * 4 + 1] + bq[gq].y);
                    pk.h[2] = __float2half_rn(acc[t][m][gq * 4 + 2] + bq[gq].z);
                    pk.h[3] = __float2half_rn(acc[t][m][gq * 4 + 3] + bq[gq].w);
                    *(uint2*)(slab + convt_slab_waddr<TZ>(gq, l31, t, kh)) = pk.u;
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int k = 0; k < TZ; ++k) {
                    const int piece = lane + 64 * k;
                    const uint4 d = convt_slab_read<TZ>(slab, pl, piece);
                    if ((ovalid >> (m * TZ + k)) & 1u) *(uint4*)(optr[m][k] + poff + (size_t)pl * ovox * 32) = d;
                }
            __builtin_amdgcn_wave_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next pass's weights have landed (and this pass's stores are out)
        __syncthreads();
    }
}

int launch_convt_mfma(boa_ctx* ctx, const ActSrc& src, int N, const int din[3], const int s[3], int Cout,
                      const __half* wpk, const float* bias, float slope, __half* out) {
    BOA_REQUIRE(src.C % 16 == 0 && Cout % 32 == 0, "convT: channels %d -> %d unsupported", src.C, Cout);
    ConvTArgs a;
    a.src = src.data; a.ss = src.ss; a.ss16 = src.ss16; a.Cin = src.C; a.Cout = Cout; a.N = N;
    a.Di = din[0]; a.Hi = din[1]; a.Wi = din[2]; a.s0 = s[0]; a.s1 = s[1]; a.s2 = s[2];
    a.wpk = wpk; a.bias = bias; a.out = out; a.slope = slope;
    const int gy_mult = 2;  // (8 and 32 measured slower: every y-slice re-stages the block's input voxels)
    size_t total = (size_t)N * din[0] * din[1] * din[2];
    int gx = (int)((total + 127) / 128);
    // split the (tap, cout-chunk) pairs over gridDim.y only as far as needed to fill the chip: every y-slice
    // re-stages the block's input voxels
    BOA_REQUIRE(s[2] == 1 || s[2] == 2, "convT: stride %d along the contiguous axis is not instantiated (1 or 2)", s[2]);
    BOA_REQUIRE((double)total < 2147483648.0 - 256.0, "convT: %zu input voxels exceed the 32-bit index range", total);
    const int npairs = s[0] * s[1] * (Cout / 32);   // (tx, ty, cout chunk); a pair covers the s2 z taps
    int gy = std::min(npairs, std::max(1, ceil_div(gy_mult * ctx->cu_count, gx)));
    size_t lds = (size_t)4 * ((src.C / 16) * 1024 + 2 * 32 * s[2] * 32);
    BOA_REQUIRE(lds <= 160 * 1024, "convT: Cin=%d needs %zu bytes of LDS", src.C, lds);
    static bool once = (hipFuncSetAttribute((const void*)k_convt_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                        hipFuncSetAttribute((const void*)k_convt_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    const double taps = (double)s[0] * s[1] * s[2];
    KernelTimer tm(ctx, BOA_K_CONVT, 2.0 * total * taps * src.C * Cout, 2.0 * total * (src.C + taps * Cout));
    static const bool no_rw = getenv("BOA_CONVT_NO_RW") != nullptr;
    const bool rw = !no_rw && s[2] == 2 && src.ss16 != nullptr && (src.C == 64 || src.C == 128);
    static const bool no_deep = getenv("BOA_CONVT_NO_DEEP") != nullptr;
    static const bool deep128 = getenv("BOA_CONVT_DEEP128") != nullptr;   // experiment: the 32^3 -> 64^3 layer on k_convt_deep too
    const bool deep = !no_deep && s[0] == 2 && s[1] == 2 && s[2] == 2 && src.ss16 != nullptr && (src.C == 256 || src.C == 320 || (deep128 && src.C == 128));
    if (deep) {
        const int ncc = src.C / 16;
        const int gxd = (int)((total + 255) / 256);    // 4 waves x 2 M-tiles x 32 voxels per block
        // the (x tap, y tap, cout chunk) passes are spread over gridDim.y until there is about one block per CU (one fits: 64-80 KiB of
        // weight buffers), at least two passes per block so that the weight DMA overlaps (measured at 25 tiles: 16^3 124 / 140 / 151 /
        // 189 us at 1 / 2 / 4 / 8 slices, 8^3 139 / 81 / 48 / 58, 4^3 166 / 94 / 52 / 34 and 22 at 20)
        static const int gy_force = getenv("BOA_CONVT_DEEP_GY") ? atoi(getenv("BOA_CONVT_DEEP_GY")) : 0;
        const int gyd = gy_force > 0 ? std::min(gy_force, npairs) : std::max(1, std::min(npairs / 2, ctx->cu_count / std::max(gxd, 1)));
        const size_t ldsd = (size_t)2 * 2 * ncc * 1024 + 4 * (2 * 32 * 2 * 32) + (size_t)Cout * sizeof(float);
        static bool od = (hipFuncSetAttribute((const void*)k_convt_deep<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                          hipFuncSetAttribute((const void*)k_convt_deep<20>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                          hipFuncSetAttribute((const void*)k_convt_deep<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
        (void)od;
        if (ncc == 8)
            hipLaunchKernelGGL(k_convt_deep<8>, dim3(gxd, gyd), dim3(256), ldsd, ctx->stream, a);
        else if (ncc == 16)
            hipLaunchKernelGGL(k_convt_deep<16>, dim3(gxd, gyd), dim3(256), ldsd, ctx->stream, a);
        else
            hipLaunchKernelGGL(k_convt_deep<20>, dim3(gxd, gyd), dim3(256), ldsd, ctx->stream, a);
    } else if (rw) {
        // register-weights variant: G groups of 32 voxels per wave (G x 128 voxels per block)
        static const int g128 = getenv("BOA_CONVT_G128") ? atoi(getenv("BOA_CONVT_G128")) : 2;   // (32^3 -> 64^3: 125 -> 105 us per 8 tiles: two workgroups per CU)
        static const int g64 = getenv("BOA_CONVT_G64") ? atoi(getenv("BOA_CONVT_G64")) : 2;
        const int G = src.C == 64 ? g64 : g128;
        const int gxr = (int)((total + 128 * G - 1) / (128 * G));
        const int gyr = std::min(npairs, std::max(1, ceil_div(gy_mult * ctx->cu_count, gxr)));
        const size_t ldsr = (size_t)4 * ((size_t)G * (src.C / 16) * 1024 + 2 * 32 * 2 * 32);
        static bool o1 = (hipFuncSetAttribute((const void*)k_convt_mfma_rw<2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                          hipFuncSetAttribute((const void*)k_convt_mfma_rw<2, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                          hipFuncSetAttribute((const void*)k_convt_mfma_rw<2, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                          hipFuncSetAttribute((const void*)k_convt_mfma_rw<2, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
        (void)o1;
        if (src.C == 64 && G == 4)
            hipLaunchKernelGGL((k_convt_mfma_rw<2, 4, 4>), dim3(gxr, gyr), dim3(256), ldsr, ctx->stream, a);
        else if (src.C == 64)
            hipLaunchKernelGGL((k_convt_mfma_rw<2, 4, 2>), dim3(gxr, gyr), dim3(256), ldsr, ctx->stream, a);
        else if (G == 4)
            hipLaunchKernelGGL((k_convt_mfma_rw<2, 8, 4>), dim3(gxr, gyr), dim3(256), ldsr, ctx->stream, a);
        else
            hipLaunchKernelGGL((k_convt_mfma_rw<2, 8, 2>), dim3(gxr, gyr), dim3(256), ldsr, ctx->stream, a);
    } else if (s[2] == 2)
        hipLaunchKernelGGL(k_convt_mfma<2>, dim3(gx, gy), dim3(256), lds, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_convt_mfma<1>, dim3(gx, gy), dim3(256), lds, ctx->stream, a);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// 1x1x1 head (+ Gaussian weighting + fp16 accumulate)
struct HeadArgs {
    const __half* act;
    const float* ss;
    int F0, P0, P1, P2, C;
    const float* w;  // [C][F0]
    const float* bias;
    float slope;
    float* logits;
    const unsigned short* gauss;
    unsigned short* acc;
    unsigned short* nacc;
    int V0, V1, V2, s0, s1, s2;
    size_t plane_stride;  // voxels between the 16-channel planes of `act` (the tile's voxel count; a stash: its own)
};

template <int F0, int VPT>
__global__ __launch_bounds__(256) void k_head(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lw = (float*)smem;        // [C][F0]
    float* lb = lw + p.C * F0;       // [C]
    float* lss = lb + p.C;           // [F0][2]
    for (int i = threadIdx.x; i < p.C * F0; i += 256) lw[i] = p.w[i];
    for (int i = threadIdx.x; i < p.C; i += 256) lb[i] = p.bias[i];
    for (int i = threadIdx.x; i < 2 * F0; i += 256) lss[i] = p.ss[i];
    __syncthreads();
    const size_t pv = (size_t)p.P0 * p.P1 * p.P2;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VPT;  // first of VPT consecutive voxels along z
    if (i >= pv) return;
    float y[VPT][F0];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
#pragma unroll
        for (int v = 0; v < F0 / 8; ++v) {
            union {
                uint4 u4;
                __half h[8];
            } x;
            x.u4 = *(const uint4*)(p.act + ((size_t)(v >> 1) * p.plane_stride + (i + u)) * 16 + 8 * (v & 1));  // chunk-planar
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = v * 8 + j;
                float f = __builtin_fmaf(__half2float(x.h[j]), lss[2 * c], lss[2 * c + 1]);
                y[u][c] = f > 0.f ? f : f * p.slope;
            }
        }
    }
    if (p.logits) {
        for (int c = 0; c < p.C; ++c) {
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                float sum = lb[c];
#pragma unroll
                for (int k = 0; k < F0; ++k) sum = __builtin_fmaf(lw[c * F0 + k], y[u][k], sum);
                p.logits[(size_t)c * pv + i + u] = sum;
            }
        }
        return;
    }
    const int p2 = (int)(i % p.P2);
    const int p1 = (int)((i / p.P2) % p.P1);
    const int p0 = (int)(i / ((size_t)p.P2 * p.P1));
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
    const size_t vi = ((size_t)(p.s0 + p0) * p.V1 + (p.s1 + p1)) * p.V2 + (p.s2 + p2);
    float g[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) g[u] = p.gauss ? us2f(p.gauss[i + u]) : 1.0f;
    for (int c = 0; c < p.C; ++c) {
        float sum[VPT];
#pragma unroll
        for (int u = 0; u < VPT; ++u) sum[u] = lb[c];
#pragma unroll
        for (int k = 0; k < F0; ++k) {
            const float wk = lw[c * F0 + k];
#pragma unroll
            for (int u = 0; u < VPT; ++u) sum[u] = __builtin_fmaf(wk, y[u][k], sum[u]);
        }
        unsigned short* ap = p.acc + (size_t)c * vv + vi;
        if (VPT == 2) {
            union {
                unsigned u32;
                unsigned short h[2];
            } a;
            a.u32 = *(const unsigned*)ap;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pr = p.gauss ? sum[u] * g[u] : sum[u];  // prediction *= gaussian (fp32)
                a.h[u] = f2us(us2f(a.h[u]) + pr);             // fp16 += fp32 (fp32 add, RTNE to fp16)
            }
            *(unsigned*)ap = a.u32;
        } else {
            float pr = p.gauss ? sum[0] * g[0] : sum[0];
            *ap = f2us(us2f(*ap) + pr);
        }
    }
#pragma unroll
    for (int u = 0; u < VPT; ++u) p.nacc[vi + u] = f2us(us2f(p.nacc[vi + u]) + g[u]);
}

// Head on the matrix cores (F0 == 32, C <= 32, accumulate mode): per 32 consecutive z voxels two
// v_mfma_f32_32x32x16_f16 (K = 32 channels) replace 32 x C fp32 FMAs per voxel.  B fragments are the voxels' channel
// records straight from global memory (16 B per lane and step) with the deferred InstanceNorm + LeakyReLU applied in
// packed fp16, A = the head weights in registers.  D[class][voxel] -> + bias, x Gaussian (fp32), fp16 `+=` into the
// accumulators exactly as k_head does it (fp32 add, one RTNE rounding): lanes 0-31 / 32-63 update two classes of the
// same 32 voxels per instruction (64 contiguous bytes each).
typedef _Float16 hh2_t __attribute__((ext_vector_type(2)));

// LOGITS = true: the same MFMA / bias / transpose path, but the fp32 logits [C][P0][P1][P2] are written out instead of
// being accumulated (boa_net_forward, and the seam that proves the accumulate arithmetic of THIS kernel bit-exact: the
// logits it writes are the values its accumulate mode multiplies by the Gaussian and adds).
template <bool LOGITS>
__global__ __launch_bounds__(256, 5) void k_head_mfma(HeadArgs p) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    f16x8 a0, a1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a0[i] = l31 < p.C ? (_Float16)p.w[l31 * 32 + 8 * kh + i] : (_Float16)0.f;
        a1[i] = l31 < p.C ? (_Float16)p.w[l31 * 32 + 16 + 8 * kh + i] : (_Float16)0.f;
    }
    // The per-lane constants -- packed (scale, shift) of this lane's 16 input channels and the 16 biases of its D rows --
    // depend on the k-half only; they live in LDS (2 x 32 words) and are re-read per M-tile instead of occupying 32 VGPRs:
    // the kernel waits on HBM round trips, and 114 -> ~80 VGPRs doubles the waves in flight (3 -> 6 per SIMD).
    __shared__ __attribute__((aligned(16))) unsigned s_ss[2][16];   // [kh][step 0: sc x4, sh x4 | step 1: sc x4, sh x4]
    __shared__ __attribute__((aligned(16))) float s_bz[2][16];
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        union {
            unsigned u;
            hh2_t v;
        } cv;
        for (int i = 0; i < 4; ++i) {
            const int c0 = 8 * k + 2 * i, c1 = 16 + 8 * k + 2 * i;
            cv.v = hh2_t{(_Float16)p.ss[2 * c0], (_Float16)p.ss[2 * c0 + 2]};
            s_ss[k][i] = cv.u;
            cv.v = hh2_t{(_Float16)p.ss[2 * c0 + 1], (_Float16)p.ss[2 * c0 + 3]};
            s_ss[k][4 + i] = cv.u;
            cv.v = hh2_t{(_Float16)p.ss[2 * c1], (_Float16)p.ss[2 * c1 + 2]};
            s_ss[k][8 + i] = cv.u;
            cv.v = hh2_t{(_Float16)p.ss[2 * c1 + 1], (_Float16)p.ss[2 * c1 + 3]};
            s_ss[k][12 + i] = cv.u;
        }
        for (int i = 0; i < 16; ++i) {
            const int c = 8 * (i >> 2) + 4 * k + (i & 3);
            s_bz[k][i] = c < p.C ? p.bias[c] : 0.f;
        }
    }
    __syncthreads();
    const hh2_t sl = hh2_t{(_Float16)p.slope, (_Float16)p.slope};
    auto xform = [&](uint4 raw, int step) {
        union {
            uint4 u;
            hh2_t v[4];
            f16x8 f;
        } x, sc, sh;
        x.u = raw;
        sc.u = *(const uint4*)&s_ss[kh][8 * step];
        sh.u = *(const uint4*)&s_ss[kh][8 * step + 4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const hh2_t y = __builtin_elementwise_fma(x.v[i], sc.v[i], sh.v[i]);
            x.v[i] = __builtin_elementwise_max(y, y * sl);
        }
        return x.f;
    };
    // per-wave LDS slab for the [class][voxel] transpose: 33 rows (32 classes + pad) x 36 floats (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) float slab_all[4][32 * 36];
    float* slab = slab_all[threadIdx.x >> 6];
    const int mpr = p.P2 / 32;                     // M-tiles per (x, y) row
    const int n_mt = p.P0 * p.P1 * mpr;
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
    const size_t pv = (size_t)p.P0 * p.P1 * p.P2;
    const int gw = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), nw = (int)(gridDim.x * 4);
    const int n_items = LOGITS ? p.C * 4 : (p.C + 1) * 4;  // (class, group of 8 voxels); class index C = the n_predictions row
    for (int mt = gw; mt < n_mt; mt += nw) {
        const int zb = (mt % mpr) * 32, row = mt / mpr, p1 = row % p.P1, p0 = row / p.P1;
        const size_t t0 = ((size_t)p0 * p.P1 + p1) * p.P2 + zb;       // first voxel of the M-tile within the tile
        const size_t v0 = ((size_t)(p.s0 + p0) * p.V1 + (p.s1 + p1)) * p.V2 + (p.s2 + zb);  // ... within the volume
        // chunk-planar: plane 0 = channels 0-15 (MFMA step 0 takes its octet kh), plane 1 = channels 16-31 (step 1); a wave
        // reads 1 KiB of consecutive bytes per plane
        const uint4 r0 = *(const uint4*)(p.act + (t0 + l31) * 16 + kh * 8);
        const uint4 r1 = *(const uint4*)(p.act + (p.plane_stride + t0 + l31) * 16 + kh * 8);
        // the RMW operands of this lane's items: issued before the MFMAs so that their latency overlaps
        uint4 gq8[2], old8[2];
        if (!LOGITS) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = lane + 64 * it;
                const int c = item >> 2, grp = item & 3;
                gq8[it] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);  // 1.0 (no Gaussian)
                old8[it] = make_uint4(0, 0, 0, 0);
                if (item < n_items) {
                    if (p.gauss) gq8[it] = *(const uint4*)(p.gauss + t0 + 8 * grp);
                    const unsigned short* src = (c < p.C ? p.acc + (size_t)c * vv : p.nacc) + v0 + 8 * grp;
                    old8[it] = *(const uint4*)src;
                }
            }
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xform(r0, 0), zero, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xform(r1, 1), d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) slab[(8 * (i >> 2) + 4 * kh + (i & 3)) * 36 + l31] = d[i] + s_bz[kh][i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = lane + 64 * it;
            const int c = item >> 2, grp = item & 3;
            if (item < n_items) {
                float4 lo = make_float4(1.f, 1.f, 1.f, 1.f), hi = lo;  // the n_predictions row adds the Gaussian itself
                if (c < p.C) {
                    lo = *(const float4*)(slab + c * 36 + 8 * grp);
                    hi = *(const float4*)(slab + c * 36 + 8 * grp + 4);
                }
                if (LOGITS) {
                    float* dst = p.logits + (size_t)c * pv + t0 + 8 * grp;
                    *(float4*)dst = lo;
                    *(float4*)(dst + 4) = hi;
                } else {
                    union {
                        uint4 u;
                        unsigned short h[8];
                    } g, o;
                    g.u = gq8[it];
                    o.u = old8[it];
                    const float sum[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gg = us2f(g.h[e]);
                        const float pr = (p.gauss || c >= p.C) ? sum[e] * gg : sum[e];  // prediction *= gaussian (fp32); n += g
                        o.h[e] = f2us(us2f(o.h[e]) + pr);                               // fp16 += fp32 (fp32 add, RTNE)
                    }
                    unsigned short* dst = (c < p.C ? p.acc + (size_t)c * vv : p.nacc) + v0 + 8 * grp;
                    *(uint4*)dst = o.u;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_head(boa_ctx* ctx, const __half* act, const float* ss, int F0, const int P[3], int C, const float* w,
                const float* bias, float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc,
                uint16_t* nacc, const int PV[3], const int start[3], size_t plane_stride) {
    BOA_REQUIRE(F0 == 32 || F0 == 64, "head: features[0]=%d unsupported (32 or 64)", F0);
    HeadArgs a;
    a.act = act; a.ss = ss; a.F0 = F0; a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2]; a.C = C; a.w = w; a.bias = bias;
    a.slope = slope; a.logits = logits_out; a.gauss = gauss; a.acc = acc; a.nacc = nacc;
    a.plane_stride = plane_stride ? plane_stride : (size_t)P[0] * P[1] * P[2];
    bool pair = (P[2] % 2 == 0) && F0 == 32;
    if (!logits_out) {
        for (int d = 0; d < 3; ++d)
            BOA_REQUIRE(start[d] >= 0 && start[d] + P[d] <= PV[d], "head: tile [%d,%d) outside accumulator dim %d (%d)",
                        start[d], start[d] + P[d], d, PV[d]);
        a.V0 = PV[0]; a.V1 = PV[1]; a.V2 = PV[2]; a.s0 = start[0]; a.s1 = start[1]; a.s2 = start[2];
        pair = pair && (PV[2] % 2 == 0) && (start[2] % 2 == 0) && (((uintptr_t)acc) % 4 == 0);
    } else {
        a.V0 = a.V1 = a.V2 = a.s0 = a.s1 = a.s2 = 0;
    }
    size_t pv = (size_t)P[0] * P[1] * P[2];
    size_t lds = ((size_t)C * F0 + C + 2 * F0) * 4;
    double bytes = (double)pv * (2.0 * F0 + (logits_out ? 4.0 * C : (4.0 * (C + 1) + 2.0)));
    KernelTimer tm(ctx, BOA_K_HEAD_ACCUM, 2.0 * pv * F0 * C, bytes);
    static const bool mfma_off = getenv("BOA_HEAD_MFMA") && atoi(getenv("BOA_HEAD_MFMA")) == 0;
    const bool mfma_shape = !mfma_off && F0 == 32 && C <= 31 && P[2] % 32 == 0 && ((uintptr_t)act) % 16 == 0;
    const unsigned mfma_grid = (unsigned)std::min<size_t>(std::max<size_t>(pv / 32 / 4, 1), (size_t)ctx->cu_count * 8);
    // 16-byte accumulator accesses: the tile's z origin, the volume's z extent and the buffers must be 8-voxel aligned
    if (!logits_out && mfma_shape && start[2] % 8 == 0 && PV[2] % 8 == 0 && ((uintptr_t)acc) % 16 == 0 &&
        ((uintptr_t)nacc) % 16 == 0 && (!gauss || ((uintptr_t)gauss) % 16 == 0)) {
        hipLaunchKernelGGL(k_head_mfma<false>, dim3(mfma_grid), dim3(256), 0, ctx->stream, a);
        ctx->counters[BOA_CNT_HEAD_MFMA]++;
    } else if (logits_out && mfma_shape && ((uintptr_t)logits_out) % 16 == 0) {
        hipLaunchKernelGGL(k_head_mfma<true>, dim3(mfma_grid), dim3(256), 0, ctx->stream, a);
        ctx->counters[BOA_CNT_HEAD_MFMA]++;
    } else if (!logits_out && mfma_shape) {
        // accumulators that do not allow the 16-byte read-modify-write (tile z origin / volume z extent not 8-aligned: the common
        // case for real CT sizes): the SAME MFMA logits (logits mode) into a scratch buffer, then the reference's accumulate step on
        // them (k_accumulate_tile: identical arithmetic, tests/test_gpu_head.py) -- so that every tile's logits come from the same
        // kernel whatever its alignment, and the logits API agrees bit for bit with the label path (gather head).  (Round 2 fell
        // back to an fp32 VALU head here, whose logits differ in the last bits.)
        float* tmp = nullptr;
        if (boa_malloc(ctx, (size_t)C * pv * sizeof(float), (void**)&tmp) != BOA_OK) {
            tm.stop();
            return BOA_ENOMEM;
        }
        HeadArgs al = a;
        al.logits = tmp;
        hipLaunchKernelGGL(k_head_mfma<true>, dim3(mfma_grid), dim3(256), 0, ctx->stream, al);
        tm.stop();
        ctx->counters[BOA_CNT_HEAD_MFMA]++;
        const int rc = boa_accumulate_tile(ctx, tmp, gauss, acc, nacc, C, P, PV, start);
        boa_free(ctx, tmp);
        if (rc) return rc;
        BOA_HIP_TRY(hipGetLastError());
        return BOA_OK;
    } else {
        if (pair)
            hipLaunchKernelGGL((k_head<32, 2>), dim3((unsigned)((pv / 2 + 255) / 256)), dim3(256), lds, ctx->stream, a);
        else if (F0 == 32)
            hipLaunchKernelGGL((k_head<32, 1>), dim3((unsigned)((pv + 255) / 256)), dim3(256), lds, ctx->stream, a);
        else
            hipLaunchKernelGGL((k_head<64, 1>), dim3((unsigned)((pv + 255) / 256)), dim3(256), lds, ctx->stream, a);
        ctx->counters[BOA_CNT_HEAD_VALU]++;
    }
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// layout helpers
// PyTorch [N][C][vox] fp32 -> the engine's chunk-planar fp16 layout [N][C/16][vox][16]
__global__ void k_nchw_to_ndhwc_f16(const float* __restrict__ in, int C, size_t vox, __half* __restrict__ out) {
    const int n = blockIdx.y;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over C * vox in the output order
    if (i >= vox * C) return;
    const int j = (int)(i % 16);
    const size_t v = (i / 16) % vox;
    const int c = (int)(i / (16 * vox)) * 16 + j;
    out[(size_t)n * vox * C + i] = __float2half_rn(in[((size_t)n * C + c) * vox + v]);
}

int launch_nchw_to_ndhwc_f16(boa_ctx* ctx, const float* in, int N, int C, size_t vox, __half* out) {
    size_t tot = vox * C;
    hipLaunchKernelGGL(k_nchw_to_ndhwc_f16, dim3((unsigned)((tot + 255) / 256), N), dim3(256), 0, ctx->stream, in, C,
                       vox, out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

__global__ void k_ndhwc_to_nchw_f32(const __half* __restrict__ in, const float* __restrict__ ss, float slope, int C,
                                    size_t vox, float* __restrict__ out) {
    const int n = blockIdx.y;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over C * vox, voxel fastest
    if (i >= vox * C) return;
    size_t v = i % vox;
    int c = (int)(i / vox);
    float f = __half2float(in[(((size_t)n * (C / 16) + c / 16) * vox + v) * 16 + (c & 15)]);  // chunk-planar
    if (ss) {
        f = __builtin_fmaf(f, ss[((size_t)n * C + c) * 2], ss[((size_t)n * C + c) * 2 + 1]);
        f = f > 0.f ? f : f * slope;
    }
    out[((size_t)n * C + c) * vox + v] = f;
}

int launch_ndhwc_to_nchw_f32(boa_ctx* ctx, const __half* in, const float* ss, float slope, int N, int C, size_t vox,
                             float* out) {
    size_t tot = vox * C;
    hipLaunchKernelGGL(k_ndhwc_to_nchw_f32, dim3((unsigned)((tot + 255) / 256), N), dim3(256), 0, ctx->stream, in, ss,
                       slope, C, vox, out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
