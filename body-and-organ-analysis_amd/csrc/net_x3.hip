// Split-precision network mode (boa_net_create precision = 2): the mode that carries the label contract -- the PlainConvUNet with
// the arithmetic of the reference's CPU path (fp32 weights, activations and accumulation, NN/inference/predict_from_raw_data.py:648:
// autocast is CUDA-only) at matrix-core speed.  Every fp32 operand is split into two fp16 parts (x = hi + lo, 22 significant bits)
// and a product of 8 real channels is one K = 16 step of v_mfma_f32_32x32x16_f16 issued twice:
//     D += [Wh | Wh] x [Xh ; Xl]  +  [Wl | Wl] x [Xh ; Xl]      (all four cross terms, fp32 accumulation)
// tools/x3_probe.hip: rms error of a K = 864 dot product 3.2e-7 of the output rms against 5.3e-7 for an fp32 FMA chain; fp16
// subnormal inputs are kept by the matrix cores.  Activations are fp32 in OCTET planes [N][C/8][voxel][8]; InstanceNorm is deferred
// into the consumer as in the fp16 mode (y = lrelu(fma(x, scale, shift)) in fp32), its statistics come from the conv epilogue's
// fp32 partial sums reduced in fp64 (k_norm_finalize).  The 3x3x3 convs are k_conv_ws<..., X3> (conv_ws.hip); this file holds the
// rest of the stack: transposed conv, 1x1x1 head (scatter form), layout helpers.
#include <algorithm>

#include "conv.h"

namespace {

typedef float x3f4 __attribute__((ext_vector_type(4)));
typedef _Float16 x3h2 __attribute__((ext_vector_type(2)));
typedef float x3f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned x3_pk(float a, float b) {
    union {
        x3h2 v;
        unsigned u;
    } c;
    c.v = __builtin_convertvector(x3f2{a, b}, x3h2);
    return c.u;
}

// y[0..3] -> hi / lo fp16 parts (8 bytes each)
__device__ __forceinline__ void x3_split4(const float (&y)[4], uint2& hi, uint2& lo) {
    union {
        x3h2 v;
        unsigned u;
    } h01, h23;
    h01.v = __builtin_convertvector(x3f2{y[0], y[1]}, x3h2);
    h23.v = __builtin_convertvector(x3f2{y[2], y[3]}, x3h2);
    hi = make_uint2(h01.u, h23.u);
    lo = make_uint2(x3_pk(y[0] - (float)h01.v[0], y[1] - (float)h01.v[1]), x3_pk(y[2] - (float)h23.v[0], y[3] - (float)h23.v[1]));
}

struct X3ConvTArgs {
    const float* src;  // [N][Cin/8][vin][8]
    const float* ss;   // [N][Cin][2] or nullptr
    int Cin, Cout, N, Di, Hi, Wi, s0, s1, s2;
    const __half* wpk;  // [tap][Cin/8][part][Cout][8], scaled by wscale
    const float* bias;
    float* out;  // [N][Cout/8][vout][8]
    float slope, winv;
};

// ConvTranspose3d with kernel == stride: out[o] = sum_ci y[o / s][ci] W[ci][co][o % s] + b.  A block takes MT x 32 consecutive
// input voxels of one sample: all threads normalise + split their Cin channels into LDS ([chunk][part][voxel][8 halves]: the B
// fragments of every chunk), then the (tap, 32-cout chunk) pairs are dealt to the four waves; a pair is 2 MT MFMAs per chunk with
// the weight fragments straight from L2 (the layer is 1.7 % of the stack's FLOPs).
template <int MT>
__global__ __launch_bounds__(256) void k_convt_x3(X3ConvTArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int n = blockIdx.y;
    const unsigned vin = (unsigned)p.Di * p.Hi * p.Wi;
    const unsigned v0 = blockIdx.x * (32u * MT);
    const int ncc = p.Cin / 8;
    {
        const float* src = p.src + (size_t)n * p.Cin * vin;
        const int items = ncc * 32 * MT * 2;
        for (int i = tid; i < items; i += 256) {
            const int half = i & 1, v = (i >> 1) % (32 * MT), cc = (i >> 1) / (32 * MT);
            const unsigned gv = v0 + v;
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            if (gv < vin) {
                const float4 x = *(const float4*)(src + ((size_t)cc * vin + gv) * 8 + half * 4);
                y[0] = x.x; y[1] = x.y; y[2] = x.z; y[3] = x.w;
                if (p.ss) {
                    const float* ss = p.ss + ((size_t)n * p.Cin + cc * 8 + half * 4) * 2;
                    const float4 s01 = *(const float4*)ss, s23 = *(const float4*)(ss + 4);
                    const float sc[4] = {s01.x, s01.z, s23.x, s23.z}, sh[4] = {s01.y, s01.w, s23.y, s23.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f = __builtin_fmaf(y[e], sc[e], sh[e]);
                        y[e] = f > 0.f ? f : f * p.slope;
                    }
                }
            }
            uint2 hi, lo;
            x3_split4(y, hi, lo);
            *(uint2*)(smem + ((size_t)(cc * 2 + 0) * 32 * MT + v) * 16 + half * 8) = hi;
            *(uint2*)(smem + ((size_t)(cc * 2 + 1) * 32 * MT + v) * 16 + half * 8) = lo;
        }
    }
    __syncthreads();
    const int taps = p.s0 * p.s1 * p.s2, nco = p.Cout / 32;
    const int Ho = p.Hi * p.s1, Wo = p.Wi * p.s2;
    const size_t vout = (size_t)vin * taps;
    // (small inputs: the (tap, cout chunk) pairs are spread over gridDim.z blocks that stage the same voxels -- the 4^3 / 8^3 layers
    //  have 2-16 voxel groups per tile and ~100 pairs of 3 MB of weights to stream: one block per group left 240 CUs idle)
    for (int pair = wave + 4 * (int)blockIdx.z; pair < taps * nco; pair += 4 * (int)gridDim.z) {
        const int tap = pair / nco, co0 = (pair - tap * nco) * 32;
        const int tz = tap % p.s2, ty = (tap / p.s2) % p.s1, tx = tap / (p.s2 * p.s1);
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        const __half* wt = p.wpk + ((size_t)tap * ncc * 2 * p.Cout + co0 + l31) * 8;
        for (int cc = 0; cc < ncc; ++cc) {
            const f16x8 ah = *(const f16x8*)(wt + (size_t)(cc * 2 + 0) * p.Cout * 8);
            const f16x8 al = *(const f16x8*)(wt + (size_t)(cc * 2 + 1) * p.Cout * 8);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f16x8 b = *(const f16x8*)(smem + ((size_t)(cc * 2 + kh) * 32 * MT + m * 32 + l31) * 16);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[m], 0, 0, 0);
            }
        }
        float bz[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 b4 = *(const float4*)(p.bias + co0 + 8 * gq + 4 * kh);
            bz[4 * gq] = b4.x; bz[4 * gq + 1] = b4.y; bz[4 * gq + 2] = b4.z; bz[4 * gq + 3] = b4.w;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned gv = v0 + m * 32 + l31;
            if (gv >= vin) continue;
            const unsigned r = gv / (unsigned)p.Wi;
            const int iz = (int)(gv - r * (unsigned)p.Wi), ix = (int)(r / (unsigned)p.Hi), iy = (int)(r - (unsigned)ix * (unsigned)p.Hi);
            const size_t vo = ((size_t)(ix * p.s0 + tx) * Ho + (iy * p.s1 + ty)) * Wo + (iz * p.s2 + tz);
            float* op = p.out + ((size_t)n * p.Cout + co0) * vout + vo * 8 + 4 * kh;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)(op + (size_t)gq * 8 * vout) = make_float4(acc[m][4 * gq] * p.winv + bz[4 * gq], acc[m][4 * gq + 1] * p.winv + bz[4 * gq + 1],
                                                                    acc[m][4 * gq + 2] * p.winv + bz[4 * gq + 2], acc[m][4 * gq + 3] * p.winv + bz[4 * gq + 3]);
        }
    }
}

struct X3HeadArgs {
    const float* act;  // octet planes [F0/8 = 4][plane_stride voxels][8] of one tile (raw output of the last decoder conv)
    const float* ss;   // [32][2]
    size_t plane_stride, pv;
    int P1, P2, C;
    const float* w;    // [C][32]
    const float* bias;
    float slope, wscale, winv;
    float* logits;     // [C][pv] or nullptr
    const unsigned short* gauss;
    unsigned short* acc;
    unsigned short* nacc;
    int V0, V1, V2, s0, s1, s2;
};

// 1x1x1 head of the split-precision mode, scatter form (logits API, mirrored / sharded / deferred paths): one wave = 32 consecutive
// voxels of the tile.  The logits are computed EXACTLY as k_gather_head_x3 computes them (same operand split, same MFMA sequence
// from a zero accumulator, * winv, + bias), so the label path and the logits API agree bit for bit; then either the fp32 logits
// are stored, or the reference's accumulate step runs (predict_from_raw_data.py:611-614: pred * gauss in fp32, fp16 += ).
__global__ __launch_bounds__(256) void k_head_x3(X3HeadArgs p) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const size_t i = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + l31;
    const bool valid = i < p.pv;
    const size_t ii = valid ? i : 0;
    f32x16 d;
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f16x8 ah, al;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float wv = l31 < p.C ? p.w[l31 * 32 + 8 * c + k] * p.wscale : 0.f;
            const _Float16 h = (_Float16)wv;
            ah[k] = h;
            al[k] = (_Float16)(wv - (float)h);
        }
        const float4 x = *(const float4*)(p.act + ((size_t)c * p.plane_stride + ii) * 8 + 4 * kh);
        const float* ss = p.ss + (8 * c + 4 * kh) * 2;
        const float4 s01 = *(const float4*)ss, s23 = *(const float4*)(ss + 4);
        const float xs[4] = {x.x, x.y, x.z, x.w}, sc[4] = {s01.x, s01.z, s23.x, s23.z}, sh[4] = {s01.y, s01.w, s23.y, s23.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = __builtin_fmaf(xs[e], sc[e], sh[e]);
            y[e] = f > 0.f ? f : f * p.slope;
        }
        uint2 hi, lo;
        x3_split4(y, hi, lo);
        // k-half 0 ends up with the hi parts of all 8 channels, k-half 1 with the lo parts (see k_gather_head_x3)
        const auto s0 = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
        union {
            unsigned u[4];
            f16x8 f;
        } b;
        b.u[0] = s0[0]; b.u[1] = s1[0]; b.u[2] = s0[1]; b.u[3] = s1[1];
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.f, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.f, d, 0, 0, 0);
    }
    if (!valid) return;
    size_t vi = 0;
    float g = 1.f;
    if (!p.logits) {
        const unsigned u = (unsigned)i, r = u / (unsigned)p.P2;
        const int p2 = (int)(u - r * (unsigned)p.P2), p0 = (int)(r / (unsigned)p.P1), p1 = (int)(r - (unsigned)p0 * (unsigned)p.P1);
        vi = ((size_t)(p.s0 + p0) * p.V1 + (p.s1 + p1)) * p.V2 + (p.s2 + p2);
        if (p.gauss) g = us2f(p.gauss[i]);
    }
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = 8 * (k >> 2) + 4 * kh + (k & 3);
        if (c >= p.C) continue;
        const float sum = d[k] * p.winv + p.bias[c];
        if (p.logits) {
            p.logits[(size_t)c * p.pv + i] = sum;
        } else {
            const float pr = p.gauss ? sum * g : sum;
            unsigned short* ap = p.acc + (size_t)c * vv + vi;
            *ap = f2us(us2f(*ap) + pr);
        }
    }
    if (!p.logits && kh == 0) p.nacc[vi] = f2us(us2f(p.nacc[vi]) + g);
}

// octet planes [C/8][vox][8] (+ deferred norm) -> [C][vox] fp32 (debug read-back)
__global__ void k_octet_to_nchw_f32(const float* __restrict__ in, const float* __restrict__ ss, float slope, int C, size_t vox,
                                    float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over C * vox, voxel fastest
    if (i >= vox * C) return;
    const size_t v = i % vox;
    const int c = (int)(i / vox);
    float f = in[((size_t)(c >> 3) * vox + v) * 8 + (c & 7)];
    if (ss) {
        f = __builtin_fmaf(f, ss[c * 2], ss[c * 2 + 1]);
        f = f > 0.f ? f : f * slope;
    }
    out[i] = f;
}

}  // namespace

int launch_convt_x3(boa_ctx* ctx, const float* src, const float* ss, int Cin, int N, const int din[3], const int s[3], int Cout,
                    const __half* wpk, float wscale, const float* bias, float slope, float* out) {
    BOA_REQUIRE(Cin % 8 == 0 && Cout % 32 == 0, "convT_x3: channel counts %d -> %d unsupported", Cin, Cout);
    X3ConvTArgs a;
    a.src = src; a.ss = ss; a.Cin = Cin; a.Cout = Cout; a.N = N; a.Di = din[0]; a.Hi = din[1]; a.Wi = din[2];
    a.s0 = s[0]; a.s1 = s[1]; a.s2 = s[2]; a.wpk = wpk; a.bias = bias; a.out = out; a.slope = slope; a.winv = 1.0f / wscale;
    const size_t vin = (size_t)din[0] * din[1] * din[2];
    BOA_REQUIRE(vin < (1u << 31), "convT_x3: input too large");
    const int taps = s[0] * s[1] * s[2];
    // two M-tiles per block halve the weight re-reads; small inputs keep one so that more blocks exist
    const int MT = vin >= 4096 ? 2 : 1;
    const size_t lds = (size_t)Cin * MT * 128;
    BOA_REQUIRE(lds <= 160 * 1024, "convT_x3: Cin=%d does not fit LDS", Cin);
    KernelTimer tm(ctx, BOA_K_CONVT, 2.0 * N * (double)vin * taps * Cin * Cout, 4.0 * N * (double)vin * (Cin + (double)taps * Cout));
    static bool once = (hipFuncSetAttribute((const void*)k_convt_x3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                        hipFuncSetAttribute((const void*)k_convt_x3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    const unsigned gx = (unsigned)((vin + 32 * MT - 1) / (32 * MT));
    const int pairs = taps * (Cout / 32);
    const int gz = std::max(1, std::min((pairs + 3) / 4, (int)(1024 / std::max(1u, gx * (unsigned)N))));
    const dim3 grid(gx, N, gz);
    if (MT == 2)
        hipLaunchKernelGGL(k_convt_x3<2>, grid, dim3(256), lds, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_convt_x3<1>, grid, dim3(256), lds, ctx->stream, a);
    ctx->counters[BOA_CNT_X3]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_head_x3(boa_ctx* ctx, const float* act, const float* ss, int F0, const int P[3], int C, const float* w, const float* bias,
                   float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc, uint16_t* nacc, const int PV[3], const int start[3],
                   size_t plane_stride) {
    BOA_REQUIRE(F0 == 32 && C >= 1 && C <= 32, "head_x3: features[0]=%d / %d classes unsupported (32 features, <= 32 classes)", F0, C);
    X3HeadArgs a;
    a.act = act; a.ss = ss; a.plane_stride = plane_stride; a.pv = (size_t)P[0] * P[1] * P[2]; a.P1 = P[1]; a.P2 = P[2]; a.C = C;
    a.w = w; a.bias = bias; a.slope = slope; a.wscale = X3_HEAD_WSCALE; a.winv = 1.0f / X3_HEAD_WSCALE;
    a.logits = logits_out; a.gauss = gauss; a.acc = acc; a.nacc = nacc;
    BOA_REQUIRE(a.pv < (1ull << 32), "head_x3: tile too large");
    if (!logits_out) {
        for (int d = 0; d < 3; ++d)
            BOA_REQUIRE(start[d] >= 0 && start[d] + P[d] <= PV[d], "head_x3: tile [%d,%d) outside accumulator dim %d (%d)", start[d], start[d] + P[d],
                        d, PV[d]);
        a.V0 = PV[0]; a.V1 = PV[1]; a.V2 = PV[2]; a.s0 = start[0]; a.s1 = start[1]; a.s2 = start[2];
    } else {
        a.V0 = a.V1 = a.V2 = a.s0 = a.s1 = a.s2 = 0;
    }
    KernelTimer tm(ctx, BOA_K_HEAD_ACCUM, 2.0 * a.pv * F0 * C, (double)a.pv * (4.0 * F0 + (logits_out ? 4.0 * C : (4.0 * (C + 1) + 2.0))));
    hipLaunchKernelGGL(k_head_x3, dim3((unsigned)((a.pv + 127) / 128)), dim3(256), 0, ctx->stream, a);
    ctx->counters[BOA_CNT_X3]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_octet_to_nchw_f32(boa_ctx* ctx, const float* in, const float* ss, float slope, int C, size_t vox, float* out) {
    hipLaunchKernelGGL(k_octet_to_nchw_f32, dim3((unsigned)((vox * C + 255) / 256)), dim3(256), 0, ctx->stream, in, ss, slope, C, vox, out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
