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
bool choose_conv_tile(const ConvGeom& g, int cu_count, ConvTile* out) {
    const int dims[3] = {g.Do, g.Ho, g.Wo};
    int w[3];
    w[2] = std::min(32, next_pow2(dims[2]));
    w[1] = std::min(32 / w[2], next_pow2(dims[1]));
    w[0] = 32 / (w[2] * w[1]);
    const int taps = g.k[0] * g.k[1] * g.k[2];
    double best_cost = 1e30;
    bool found = false;
    const int force = conv_variant_override();
    for (int variant : {1, 0}) {
        if (force >= 0 && variant != force) continue;
        for (int R : {4, 2, 1}) {
            const int M = 4 * R;
            for (int b0 = 1; b0 <= M; b0 *= 2)
                for (int b1 = 1; b0 * b1 <= M; b1 *= 2) {
                    int b2 = M / (b0 * b1);
                    if (b0 * b1 * b2 != M) continue;
                    int b[3] = {b0, b1, b2};
                    int h[3], tl[3];
                    long long HV = 1, covered = 1, tiles = 1;
                    for (int d = 0; d < 3; ++d) {
                        int ext = b[d] * w[d];
                        h[d] = (ext - 1) * g.s[d] + g.k[d];
                        tl[d] = ceil_div(dims[d], ext);
                        HV *= h[d];
                        covered *= (long long)tl[d] * ext;
                        tiles *= tl[d];
                    }
                    if (variant == 1 && !conv_ws_supported(g.k, (int)HV)) continue;
                    const int ncc = g.Cin / 16;
                    size_t lds = variant == 1 ? conv_ws_lds_bytes((int)HV, taps, ncc, g.Cout) : conv_lds_bytes((int)HV, taps);
                    if (lds > 160 * 1024) continue;
                    const long long nblocks = tiles * (g.Cout / 32) * g.N;
                    const double valid = (double)dims[0] * dims[1] * dims[2];
                    const double waste = (double)covered / valid;
                    const double t_mfma = (double)taps * R * 32.0;
                    double items = (2.0 * HV + taps * 64.0) / 256.0;
                    double t_chunk, slots;
                    if (variant == 1) {
                        if (conv_ws_resident((int)HV, taps, ncc, g.Cout)) items = 2.0 * HV / 256.0;
                        t_chunk = std::max(t_mfma * 1.1, items * 130.0) + 400.0;
                        slots = cu_count;
                    } else {
                        const int bpc = lds <= 78 * 1024 ? 2 : 1;
                        t_chunk = (t_mfma + items * 200.0 + 400.0) / (bpc == 2 ? 1.6 : 1.0);
                        slots = cu_count;
                    }
                    const double rounds = std::ceil((double)nblocks / slots);  // quantisation on a part-filled chip
                    const double cost = t_chunk * rounds * waste / (double)(M) * ((double)slots / (double)nblocks);
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
                    }
                }
        }
    }
    return found;
}

// number of per-(n, cout) partial-statistics entries the conv kernel writes
int conv_nblk(const ConvTile& t) { return t.tiles[0] * t.tiles[1] * t.tiles[2] * (t.variant == 1 ? 4 : 1); }

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
            if (cg < p.C0) {
                base = p.src0 + (size_t)n * in_vox * p.C0 + cg;
                ss = p.ss0 ? p.ss0 + ((size_t)n * p.C0 + cg) * 2 : nullptr;
                C = p.C0;
            } else {
                cg -= p.C0;
                base = p.src1 + (size_t)n * in_vox * p.C1 + cg;
                ss = p.ss1 ? p.ss1 + ((size_t)n * p.C1 + cg) * 2 : nullptr;
                C = p.C1;
            }
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
                    val = *(const uint4*)(base + (size_t)gi * C);
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
        __half* op = p.out + ((size_t)n * out_vox + ovox[r]) * p.Cout + cout0 + 4 * kh;
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
            *(uint2*)(op + 8 * g) = pk.u;
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
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    a.lw1 = ilog2(t.w[1]); a.lw2 = ilog2(t.w[2]); a.lb1 = ilog2(t.b[1]); a.lb2 = ilog2(t.b[2]);
    a.wpk = wpk; a.bias = bias; a.out = out; a.partials = partials; a.slope = slope;
    dim3 grid(t.tiles[0] * t.tiles[1] * t.tiles[2], g.Cout / 32, g.N);
    const int taps = g.k[0] * g.k[1] * g.k[2];
    const double vox = (double)g.N * g.Do * g.Ho * g.Wo;
    const double flops = 2.0 * vox * taps * (s0.C + s1.C) * g.Cout;
    const double bytes = 2.0 * ((double)g.N * g.Di * g.Hi * g.Wi * (s0.C + s1.C) + vox * g.Cout);
    if (t.variant == 1) return launch_conv_ws(ctx, a, t, flops, bytes);
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

// ======================================================================================================
// first conv: fp32 VALU, tiles gathered from the resident volume
#define FT0 4
#define FT1 4
#define FT2 16
struct FirstArgs {
    const float* vol;
    const int* origins;
    int V0, V1, V2, o0, o1, o2;  // volume dims and position of the volume inside the padded grid
    int N, Cin, P0, P1, P2, k0, k1, k2, Cout;
    const float* w;  // [Cin][taps][Cout]
    const float* bias;
    __half* out;
    float* partials;
    int t0, t1, t2;
};

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
    const int pad0 = (p.k0 - 1) / 2, pad1 = (p.k1 - 1) / 2, pad2 = (p.k2 - 1) / 2;
    const int h0 = FT0 + p.k0 - 1, h1 = FT1 + p.k1 - 1, h2 = FT2 + p.k2 - 1;
    const int HV = h0 * h1 * h2;
    float* lds_in = (float*)smem;  // [Cin][HV]
    float* lds_red = lds_in + p.Cin * HV;
    const int org0 = p.origins[n * 3 + 0], org1 = p.origins[n * 3 + 1], org2 = p.origins[n * 3 + 2];
    const int x0 = tx * FT0 - pad0, y0 = ty * FT1 - pad1, z0 = tz * FT2 - pad2;  // tile-local halo origin
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
    for (int i = tid; i < p.Cin * HV; i += 256) {
        int ci = i / HV;
        int v = i % HV;
        int hz = v % h2;
        int t = v / h2;
        int hy = t % h1;
        int hx = t / h1;
        int px = x0 + hx, py = y0 + hy, pz = z0 + hz;  // patch coordinates
        float val = 0.f;
        if (px >= 0 && px < p.P0 && py >= 0 && py < p.P1 && pz >= 0 && pz < p.P2) {
            int vx = org0 + px - p.o0, vy = org1 + py - p.o1, vz = org2 + pz - p.o2;  // volume coordinates
            if (vx >= 0 && vx < p.V0 && vy >= 0 && vy < p.V1 && vz >= 0 && vz < p.V2)
                val = p.vol[(size_t)ci * vv + ((size_t)vx * p.V1 + vy) * p.V2 + vz];
        }
        lds_in[i] = val;
    }
    __syncthreads();
    const int lz = tid % FT2;
    const int ly = (tid / FT2) % FT1;
    const int lx = tid / (FT2 * FT1);
    const int ox = tx * FT0 + lx, oy = ty * FT1 + ly, oz = tz * FT2 + lz;
    const bool valid = ox < p.P0 && oy < p.P1 && oz < p.P2;
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = p.bias[cout0 + c];
    const int taps = p.k0 * p.k1 * p.k2;
    for (int ci = 0; ci < p.Cin; ++ci) {
        int tap = 0;
        for (int dx = 0; dx < p.k0; ++dx)
            for (int dy = 0; dy < p.k1; ++dy)
                for (int dz = 0; dz < p.k2; ++dz, ++tap) {
                    float x = lds_in[ci * HV + ((lx + dx) * h1 + (ly + dy)) * h2 + lz + dz];
                    const float* wr = p.w + ((size_t)ci * taps + tap) * p.Cout + cout0;  // wave-uniform -> scalar loads
#pragma unroll
                    for (int c = 0; c < 32; ++c) acc[c] = __builtin_fmaf(x, wr[c], acc[c]);
                }
    }
    float s[32], q[32];
    if (valid) {
        __half* op = p.out + (((size_t)n * p.P0 + ox) * p.P1 + oy) * (size_t)p.P2 * p.Cout + (size_t)oz * p.Cout + cout0;
        union {
            uint4 u[4];
            __half h[32];
        } pk;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            __half hv = __float2half_rn(acc[c]);
            pk.h[c] = hv;
            float vr = __half2float(hv);
            s[c] = vr;
            q[c] = vr * vr;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ((uint4*)op)[j] = pk.u[j];
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) s[c] = q[c] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            s[c] += __shfl_xor(s[c], m);
            q[c] += __shfl_xor(q[c], m);
        }
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            lds_red[(wave * 32 + c) * 2 + 0] = s[c];
            lds_red[(wave * 32 + c) * 2 + 1] = q[c];
        }
    }
    __syncthreads();
    if (tid < 64) {
        int row = tid >> 1, j = tid & 1;
        float v = lds_red[(0 * 32 + row) * 2 + j] + lds_red[(1 * 32 + row) * 2 + j];
        v += lds_red[(2 * 32 + row) * 2 + j];
        v += lds_red[(3 * 32 + row) * 2 + j];
        p.partials[(((size_t)n * p.Cout + cout0 + row) * 2 + j) * gridDim.x + blockIdx.x] = v;
    }
}

int conv_first_nblk(const int P[3]) { return ceil_div(P[0], FT0) * ceil_div(P[1], FT1) * ceil_div(P[2], FT2); }

int launch_conv_first(boa_ctx* ctx, const float* volume, const int V[3], const int vol_off[3], const int* dev_origins,
                      int N, int Cin, const int P[3], const int k[3], int Cout, const float* w, const float* bias,
                      __half* out, float* partials, int* nblk_out) {
    BOA_REQUIRE(Cout % 32 == 0, "first conv: Cout=%d must be a multiple of 32", Cout);
    BOA_REQUIRE(Cin >= 1 && Cin <= 8, "first conv: Cin=%d unsupported (1..8)", Cin);
    FirstArgs a;
    a.vol = volume; a.origins = dev_origins;
    a.V0 = V[0]; a.V1 = V[1]; a.V2 = V[2];
    a.o0 = vol_off ? vol_off[0] : 0; a.o1 = vol_off ? vol_off[1] : 0; a.o2 = vol_off ? vol_off[2] : 0;
    a.N = N; a.Cin = Cin; a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2]; a.k0 = k[0]; a.k1 = k[1]; a.k2 = k[2]; a.Cout = Cout;
    a.w = w; a.bias = bias; a.out = out; a.partials = partials;
    a.t0 = ceil_div(P[0], FT0); a.t1 = ceil_div(P[1], FT1); a.t2 = ceil_div(P[2], FT2);
    int nblk = a.t0 * a.t1 * a.t2;
    if (nblk_out) *nblk_out = nblk;
    int HV = (FT0 + k[0] - 1) * (FT1 + k[1] - 1) * (FT2 + k[2] - 1);
    size_t lds = (size_t)Cin * HV * 4 + 1024;
    const double vox = (double)N * P[0] * P[1] * P[2];
    KernelTimer tm(ctx, BOA_K_CONV_FIRST, 2.0 * vox * k[0] * k[1] * k[2] * Cin * Cout, vox * (4.0 * Cin + 2.0 * Cout));
    hipLaunchKernelGGL(k_conv_first, dim3(nblk, Cout / 32, N), dim3(256), lds, ctx->stream, a);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// InstanceNorm finalize: deterministic fp64 reduction of the per-block partials
__global__ __launch_bounds__(64) void k_norm_finalize(const float* __restrict__ partials, int nblk, int C, double count,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float* __restrict__ ss,
                                                      unsigned short* __restrict__ ss16) {
    const int c = blockIdx.x, n = blockIdx.y;
    const float* ps = partials + (((size_t)n * C + c) * 2 + 0) * nblk;
    const float* pq = partials + (((size_t)n * C + c) * 2 + 1) * nblk;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) {
        s += (double)ps[i];
        q += (double)pq[i];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m);
        q += __shfl_xor(q, m);
    }
    if (threadIdx.x == 0) {
        double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        double inv = 1.0 / sqrt(var + (double)eps);
        float scale = (float)((double)gamma[c] * inv);
        float shift = (float)((double)beta[c] - mean * (double)gamma[c] * inv);
        ss[((size_t)n * C + c) * 2 + 0] = scale;
        ss[((size_t)n * C + c) * 2 + 1] = shift;
        if (ss16) {  // packed fp16 copy for k_conv_ws: per channel pair {s_c, s_c+1, t_c, t_c+1}
            unsigned short* q16 = ss16 + ((size_t)n * C + (c & ~1)) * 2;
            q16[c & 1] = f2us(scale);
            q16[2 + (c & 1)] = f2us(shift);
        }
    }
}

int launch_norm_finalize(boa_ctx* ctx, const float* partials, int nblk, int N, int C, double count,
                         const float* gamma, const float* beta, float eps, float* ss_out, unsigned* ss16_out) {
    KernelTimer tm(ctx, BOA_K_NORM_FINALIZE, 0, (double)N * C * nblk * 8.0);
    hipLaunchKernelGGL(k_norm_finalize, dim3(C, N), dim3(64), 0, ctx->stream, partials, nblk, C, count, gamma, beta,
                       eps, ss_out, (unsigned short*)ss16_out);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// transposed conv, kernel == stride: out[o] = sum_ci y[o / s][ci] * W[ci][co][o % s] + b
struct ConvTArgs {
    const __half* src;
    const float* ss;
    int Cin, Cout, N, Di, Hi, Wi, s0, s1, s2;
    const __half* wpk;  // [tap][Cin/16][2][Cout][8]
    const float* bias;
    __half* out;
    float slope;
};

__global__ __launch_bounds__(256) void k_convt_mfma(ConvTArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    const int ncc = p.Cin / 16;
    const size_t in_vox = (size_t)p.Di * p.Hi * p.Wi;
    const size_t total = (size_t)p.N * in_vox;
    unsigned char* lds = smem + (size_t)wave * ncc * 1024;  // [cc][khalf][32 voxels][8 halves]
    const size_t g = ((size_t)blockIdx.x * 4 + wave) * 32 + l31;  // flattened (n, voxel)
    const bool valid = g < total;
    const int n = valid ? (int)(g / in_vox) : 0;
    const size_t vi = valid ? g % in_vox : 0;
    // stage this wave's 32 voxels: lane (l31, kh) moves octet kh of every 16-channel chunk
    for (int cc = 0; cc < ncc; ++cc) {
        uint4 val = make_uint4(0, 0, 0, 0);
        if (valid) {
            int cg = cc * 16 + kh * 8;
            val = *(const uint4*)(p.src + ((size_t)n * in_vox + vi) * p.Cin + cg);
            if (p.ss) {
                float sc[8], sh[8];
                const float* ss = p.ss + ((size_t)n * p.Cin + cg) * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sc[j] = ss[2 * j];
                    sh[j] = ss[2 * j + 1];
                }
                val = norm_act8(val, sc, sh, p.slope);
            }
        }
        *(uint4*)(lds + ((cc * 2 + kh) * 32 + l31) * 16) = val;
    }
    __syncthreads();
    const int iz = (int)(vi % p.Wi);
    const int iy = (int)((vi / p.Wi) % p.Hi);
    const int ix = (int)(vi / ((size_t)p.Wi * p.Hi));
    const int Do = p.Di * p.s0, Ho = p.Hi * p.s1, Wo = p.Wi * p.s2;
    const int taps = p.s0 * p.s1 * p.s2;
    const int nco = p.Cout / 32;
    for (int tap = 0; tap < taps; ++tap) {
        const int tz = tap % p.s2, ty = (tap / p.s2) % p.s1, tx = tap / (p.s2 * p.s1);
        const size_t ovox = ((size_t)(ix * p.s0 + tx) * Ho + (iy * p.s1 + ty)) * Wo + (iz * p.s2 + tz);
        for (int co = 0; co < nco; ++co) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const __half* wb = p.wpk + (((size_t)tap * ncc * 2 + kh) * p.Cout + co * 32 + l31) * 8;
            for (int cc = 0; cc < ncc; ++cc) {
                f16x8 a = *(const f16x8*)(wb + (size_t)cc * 2 * p.Cout * 8);
                f16x8 b = *(const f16x8*)(lds + ((cc * 2 + kh) * 32 + l31) * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            }
            if (valid) {
                __half* op = p.out + ((size_t)n * Do * Ho * Wo + ovox) * p.Cout + co * 32 + 4 * kh;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    union {
                        uint2 u;
                        __half h[4];
                    } pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pk.h[j] = __float2half_rn(acc[gq * 4 + j] + p.bias[co * 32 + 8 * gq + 4 * kh + j]);
                    *(uint2*)(op + 8 * gq) = pk.u;
                }
            }
        }
    }
}

int launch_convt_mfma(boa_ctx* ctx, const ActSrc& src, int N, const int din[3], const int s[3], int Cout,
                      const __half* wpk, const float* bias, float slope, __half* out) {
    BOA_REQUIRE(src.C % 16 == 0 && Cout % 32 == 0, "convT: channels %d -> %d unsupported", src.C, Cout);
    ConvTArgs a;
    a.src = src.data; a.ss = src.ss; a.Cin = src.C; a.Cout = Cout; a.N = N;
    a.Di = din[0]; a.Hi = din[1]; a.Wi = din[2]; a.s0 = s[0]; a.s1 = s[1]; a.s2 = s[2];
    a.wpk = wpk; a.bias = bias; a.out = out; a.slope = slope;
    size_t total = (size_t)N * din[0] * din[1] * din[2];
    int grid = (int)((total + 127) / 128);
    size_t lds = (size_t)4 * (src.C / 16) * 1024;
    static bool once = (hipFuncSetAttribute((const void*)k_convt_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    const double taps = (double)s[0] * s[1] * s[2];
    KernelTimer tm(ctx, BOA_K_CONVT, 2.0 * total * taps * src.C * Cout, 2.0 * total * (src.C + taps * Cout));
    hipLaunchKernelGGL(k_convt_mfma, dim3(grid), dim3(256), lds, ctx->stream, a);
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
};

template <int F0>
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
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pv) return;
    float y[F0];
    {
        const uint4* ap = (const uint4*)(p.act + i * F0);
#pragma unroll
        for (int v = 0; v < F0 / 8; ++v) {
            union {
                uint4 u;
                __half h[8];
            } x;
            x.u = ap[v];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = v * 8 + j;
                float f = __builtin_fmaf(__half2float(x.h[j]), lss[2 * c], lss[2 * c + 1]);
                y[c] = f > 0.f ? f : f * p.slope;
            }
        }
    }
    if (p.logits) {
        for (int c = 0; c < p.C; ++c) {
            float sum = lb[c];
#pragma unroll
            for (int k = 0; k < F0; ++k) sum = __builtin_fmaf(lw[c * F0 + k], y[k], sum);
            p.logits[(size_t)c * pv + i] = sum;
        }
        return;
    }
    const int p2 = (int)(i % p.P2);
    const int p1 = (int)((i / p.P2) % p.P1);
    const int p0 = (int)(i / ((size_t)p.P2 * p.P1));
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
    const size_t vi = ((size_t)(p.s0 + p0) * p.V1 + (p.s1 + p1)) * p.V2 + (p.s2 + p2);
    const float g = p.gauss ? us2f(p.gauss[i]) : 1.0f;
    for (int c = 0; c < p.C; ++c) {
        float sum = lb[c];
#pragma unroll
        for (int k = 0; k < F0; ++k) sum = __builtin_fmaf(lw[c * F0 + k], y[k], sum);
        if (p.gauss) sum = sum * g;
        unsigned short* ap = p.acc + (size_t)c * vv + vi;
        *ap = f2us(us2f(*ap) + sum);
    }
    p.nacc[vi] = f2us(us2f(p.nacc[vi]) + g);
}

int launch_head(boa_ctx* ctx, const __half* act, const float* ss, int F0, const int P[3], int C, const float* w,
                const float* bias, float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc,
                uint16_t* nacc, const int PV[3], const int start[3]) {
    BOA_REQUIRE(F0 == 32 || F0 == 64, "head: features[0]=%d unsupported (32 or 64)", F0);
    HeadArgs a;
    a.act = act; a.ss = ss; a.F0 = F0; a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2]; a.C = C; a.w = w; a.bias = bias;
    a.slope = slope; a.logits = logits_out; a.gauss = gauss; a.acc = acc; a.nacc = nacc;
    if (!logits_out) {
        for (int d = 0; d < 3; ++d)
            BOA_REQUIRE(start[d] >= 0 && start[d] + P[d] <= PV[d], "head: tile [%d,%d) outside accumulator dim %d (%d)",
                        start[d], start[d] + P[d], d, PV[d]);
        a.V0 = PV[0]; a.V1 = PV[1]; a.V2 = PV[2]; a.s0 = start[0]; a.s1 = start[1]; a.s2 = start[2];
    } else {
        a.V0 = a.V1 = a.V2 = a.s0 = a.s1 = a.s2 = 0;
    }
    size_t pv = (size_t)P[0] * P[1] * P[2];
    size_t lds = ((size_t)C * F0 + C + 2 * F0) * 4;
    int grid = (int)((pv + 255) / 256);
    double bytes = (double)pv * (2.0 * F0 + (logits_out ? 4.0 * C : (4.0 * (C + 1) + 2.0)));
    KernelTimer tm(ctx, BOA_K_HEAD_ACCUM, 2.0 * pv * F0 * C, bytes);
    if (F0 == 32)
        hipLaunchKernelGGL(k_head<32>, dim3(grid), dim3(256), lds, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_head<64>, dim3(grid), dim3(256), lds, ctx->stream, a);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ======================================================================================================
// layout helpers
__global__ void k_nchw_to_ndhwc_f16(const float* __restrict__ in, int C, size_t vox, __half* __restrict__ out) {
    const int n = blockIdx.y;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over vox * C, channel fastest
    if (i >= vox * C) return;
    int c = (int)(i % C);
    size_t v = i / C;
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
    float f = __half2float(in[((size_t)n * vox + v) * C + c]);
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
