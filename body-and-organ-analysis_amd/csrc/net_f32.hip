// fp32 "exact" network mode (boa_net_create precision = 1): the PlainConvUNet evaluated the way the reference's CPU path
// evaluates it -- fp32 weights, fp32 activations, fp32 accumulation (NN/inference/predict_from_raw_data.py:648: autocast is
// only entered on CUDA devices) -- so that the device result can be compared with the torch-CPU oracle without the fp16
// storage error of the production path.  A correctness mode: straightforward kernels on v_mfma_f32_32x32x2_f32 with
// operands straight from global memory; speed is irrelevant (≈20-40 ms per 128^3 tile).
//
// Layout: activations channels-last fp32 [N][X][Y][Z][C]; weights [tap][Cin][Cout] fp32; InstanceNorm is deferred into the
// consumer exactly as in the fp16 path (y = lrelu(x * scale + shift), fp32 fma), statistics by a separate fp64 reduction.
#include "conv.h"

namespace {

struct F32ConvArgs {
    const float* src0;
    const float* src1;
    const float* ss0;  // [N][C0][2] or nullptr (raw)
    const float* ss1;
    int C0, C1;
    int N, Di, Hi, Wi, Do, Ho, Wo, Cout;
    int k0, k1, k2, s0, s1, s2;
    const float* w;  // [taps][Cin][Cout]
    const float* bias;
    float* out;  // [N][Do][Ho][Wo][Cout]
    float slope;
};

// One wave: D[32 couts][32 consecutive output voxels] = sum over (tap, cin) of W[cout][k] * X[k][voxel], K = 2 per MFMA
// (lane (i = l % 32, k = l / 32) supplies A[i][k] and B[k][j = l % 32]).
__global__ __launch_bounds__(256) void k_conv_f32(F32ConvArgs p) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const size_t ovox = (size_t)p.Do * p.Ho * p.Wo, total = (size_t)p.N * ovox;
    const size_t g = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + l31;  // flattened (n, output voxel)
    const bool valid = g < total;
    const int n = valid ? (int)(g / ovox) : 0;
    const size_t vo = valid ? g % ovox : 0;
    const int oz = (int)(vo % p.Wo), oy = (int)((vo / p.Wo) % p.Ho), ox = (int)(vo / ((size_t)p.Wo * p.Ho));
    const int cout0 = blockIdx.y * 32;
    const int Cin = p.C0 + p.C1;
    const size_t ivox = (size_t)p.Di * p.Hi * p.Wi;
    const int p0 = (p.k0 - 1) / 2, p1 = (p.k1 - 1) / 2, p2 = (p.k2 - 1) / 2;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    int tap = 0;
    for (int dx = 0; dx < p.k0; ++dx)
        for (int dy = 0; dy < p.k1; ++dy)
            for (int dz = 0; dz < p.k2; ++dz, ++tap) {
                const int ix = ox * p.s0 + dx - p0, iy = oy * p.s1 + dy - p1, iz = oz * p.s2 + dz - p2;
                const bool inb = valid && ix >= 0 && ix < p.Di && iy >= 0 && iy < p.Hi && iz >= 0 && iz < p.Wi;
                const size_t vi = inb ? ((size_t)ix * p.Hi + iy) * p.Wi + iz : 0;
                const float* x0 = p.src0 + ((size_t)n * ivox + vi) * p.C0;
                const float* x1 = p.src1 ? p.src1 + ((size_t)n * ivox + vi) * p.C1 : nullptr;
                const float* wt = p.w + (size_t)tap * Cin * p.Cout + cout0 + l31;
                for (int c2 = 0; c2 < Cin; c2 += 2) {
                    const int c = c2 + kh;
                    float a = 0.f, b = 0.f;
                    if (c < Cin) {
                        a = wt[(size_t)c * p.Cout];
                        if (inb) {  // zero padding applies to the NORMALISED tensor: padding voxels contribute 0
                            if (c < p.C0) {
                                b = x0[c];
                                if (p.ss0) {
                                    const float* ss = p.ss0 + ((size_t)n * p.C0 + c) * 2;
                                    b = __builtin_fmaf(b, ss[0], ss[1]);
                                    b = b > 0.f ? b : b * p.slope;
                                }
                            } else {
                                const int c1 = c - p.C0;
                                b = x1[c1];
                                if (p.ss1) {
                                    const float* ss = p.ss1 + ((size_t)n * p.C1 + c1) * 2;
                                    b = __builtin_fmaf(b, ss[0], ss[1]);
                                    b = b > 0.f ? b : b * p.slope;
                                }
                            }
                        }
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
    if (!valid) return;
    float* op = p.out + g * p.Cout + cout0 + 4 * kh;  // D rows of this lane: 8 q + 4 kh + e
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bq = *(const float4*)(p.bias + cout0 + 8 * q + 4 * kh);
        *(float4*)(op + 8 * q) = make_float4(acc[4 * q] + bq.x, acc[4 * q + 1] + bq.y, acc[4 * q + 2] + bq.z, acc[4 * q + 3] + bq.w);
    }
}

struct F32ConvTArgs {
    const float* src;
    const float* ss;
    int Cin, Cout, N, Di, Hi, Wi, s0, s1, s2;
    const float* w;  // [tap][Cin][Cout]
    const float* bias;
    float* out;
    float slope;
};

// ConvTranspose3d, kernel == stride: out[o] = sum_ci y[o / s][ci] W[ci][co][o % s] + b: per tap a [cout] x [Cin] x [input
// voxel] product whose result lands at output voxel in * s + tap.
__global__ __launch_bounds__(256) void k_convt_f32(F32ConvTArgs p) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const size_t ivox = (size_t)p.Di * p.Hi * p.Wi, total = (size_t)p.N * ivox;
    const size_t g = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + l31;
    const bool valid = g < total;
    const int n = valid ? (int)(g / ivox) : 0;
    const size_t vi = valid ? g % ivox : 0;
    const int iz = (int)(vi % p.Wi), iy = (int)((vi / p.Wi) % p.Hi), ix = (int)(vi / ((size_t)p.Wi * p.Hi));
    const int cout0 = blockIdx.y * 32;
    const int tap = blockIdx.z;
    const int tz = tap % p.s2, ty = (tap / p.s2) % p.s1, tx = tap / (p.s2 * p.s1);
    const float* x = p.src + ((size_t)n * ivox + vi) * p.Cin;
    const float* wt = p.w + (size_t)tap * p.Cin * p.Cout + cout0 + l31;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int c2 = 0; c2 < p.Cin; c2 += 2) {
        const int c = c2 + kh;
        float a = 0.f, b = 0.f;
        if (c < p.Cin) {
            a = wt[(size_t)c * p.Cout];
            if (valid) {
                b = x[c];
                if (p.ss) {
                    const float* ss = p.ss + ((size_t)n * p.Cin + c) * 2;
                    b = __builtin_fmaf(b, ss[0], ss[1]);
                    b = b > 0.f ? b : b * p.slope;
                }
            }
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (!valid) return;
    const int Ho = p.Hi * p.s1, Wo = p.Wi * p.s2;
    const size_t ovox = (size_t)p.Di * p.s0 * Ho * Wo;
    const size_t vo = ((size_t)(ix * p.s0 + tx) * Ho + (iy * p.s1 + ty)) * Wo + (iz * p.s2 + tz);
    float* op = p.out + ((size_t)n * ovox + vo) * p.Cout + cout0 + 4 * kh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bq = *(const float4*)(p.bias + cout0 + 8 * q + 4 * kh);
        *(float4*)(op + 8 * q) = make_float4(acc[4 * q] + bq.x, acc[4 * q + 1] + bq.y, acc[4 * q + 2] + bq.z, acc[4 * q + 3] + bq.w);
    }
}

// InstanceNorm statistics of one (n, c): fp64 sums over the voxels in a fixed order -> (scale, shift)
__global__ __launch_bounds__(256) void k_stats_f32(const float* __restrict__ act, size_t vox, int C, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, float* __restrict__ ss) {
    const int c = blockIdx.x, n = blockIdx.y;
    const float* a = act + (size_t)n * vox * C + c;
    double s = 0.0, q = 0.0;
    for (size_t v = threadIdx.x; v < vox; v += 256) {
        const double x = (double)a[v * C];
        s += x;
        q += x * x;
    }
    __shared__ double red[8];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m);
        q += __shfl_xor(q, m);
    }
    if ((threadIdx.x & 63) == 0) {
        red[(threadIdx.x >> 6) * 2] = s;
        red[(threadIdx.x >> 6) * 2 + 1] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[2]) + (red[4] + red[6]);
        q = (red[1] + red[3]) + (red[5] + red[7]);
        const double mean = s / (double)vox;
        double var = q / (double)vox - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + (double)eps);
        ss[((size_t)n * C + c) * 2 + 0] = (float)((double)gamma[c] * inv);
        ss[((size_t)n * C + c) * 2 + 1] = (float)((double)beta[c] - mean * (double)gamma[c] * inv);
    }
}

// tiles out of the resident volume [Cin][V] into channels-last fp32 [N][P][Cin]; voxels outside the volume read 0 (pad_nd_image)
__global__ __launch_bounds__(256) void k_gather_tiles_f32(const float* __restrict__ vol, const int* __restrict__ origins, int V0, int V1,
                                                          int V2, int o0, int o1, int o2, int Cin, int P0, int P1, int P2, int flip,
                                                          float* __restrict__ out) {
    const int n = blockIdx.y;
    const size_t pv = (size_t)P0 * P1 * P2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pv * Cin) return;
    const int ci = (int)(i % Cin);
    const size_t v = i / Cin;
    int pz = (int)(v % P2), py = (int)((v / P2) % P1), px = (int)(v / ((size_t)P2 * P1));
    if (flip & 1) px = P0 - 1 - px;  // test-time mirroring: the network sees torch.flip(tile, axes)
    if (flip & 2) py = P1 - 1 - py;
    if (flip & 4) pz = P2 - 1 - pz;
    const int vx = origins[n * 3 + 0] + px - o0, vy = origins[n * 3 + 1] + py - o1, vz = origins[n * 3 + 2] + pz - o2;
    float x = 0.f;
    if (vx >= 0 && vx < V0 && vy >= 0 && vy < V1 && vz >= 0 && vz < V2) x = vol[(size_t)ci * V0 * V1 * V2 + ((size_t)vx * V1 + vy) * V2 + vz];
    out[(size_t)n * pv * Cin + i] = x;
}

struct F32HeadArgs {
    const float* act;
    const float* ss;
    int F0, C;
    size_t pv;
    int P1, P2;
    const float* w;
    const float* bias;
    float slope;
    float* logits;
    const unsigned short* gauss;
    unsigned short* acc;
    unsigned short* nacc;
    int V0, V1, V2, s0, s1, s2;
    size_t octet_stride;  // 0: channels-last records [voxel][F0]; else octet planes [F0/8][octet_stride voxels][8] (precision 2)
};

// 1x1x1 head in fp32 (fma chain over the features in index order) + the reference's accumulate step
// (predict_from_raw_data.py:611-614): pred * gauss in fp32, fp16 += with one RTNE rounding.
__global__ __launch_bounds__(256) void k_head_f32(F32HeadArgs p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.pv) return;
    const float* a = p.act + i * p.F0;
    float y[64];
    for (int k = 0; k < p.F0; ++k) {
        const float x = p.octet_stride ? p.act[((size_t)(k >> 3) * p.octet_stride + i) * 8 + (k & 7)] : a[k];
        float f = __builtin_fmaf(x, p.ss[2 * k], p.ss[2 * k + 1]);
        y[k] = f > 0.f ? f : f * p.slope;
    }
    size_t vi = 0;
    float g = 1.f;
    if (!p.logits) {
        const int p2 = (int)(i % p.P2), p1 = (int)((i / p.P2) % p.P1), p0 = (int)(i / ((size_t)p.P2 * p.P1));
        vi = ((size_t)(p.s0 + p0) * p.V1 + (p.s1 + p1)) * p.V2 + (p.s2 + p2);
        if (p.gauss) g = us2f(p.gauss[i]);
    }
    const size_t vv = (size_t)p.V0 * p.V1 * p.V2;
    for (int c = 0; c < p.C; ++c) {
        float sum = p.bias[c];
        for (int k = 0; k < p.F0; ++k) sum = __builtin_fmaf(p.w[c * p.F0 + k], y[k], sum);
        if (p.logits) {
            p.logits[(size_t)c * p.pv + i] = sum;
        } else {
            const float pr = p.gauss ? sum * g : sum;
            unsigned short* ap = p.acc + (size_t)c * vv + vi;
            *ap = f2us(us2f(*ap) + pr);
        }
    }
    if (!p.logits) p.nacc[vi] = f2us(us2f(p.nacc[vi]) + g);
}

__global__ void k_ndhwc32_to_nchw_f32(const float* __restrict__ in, const float* __restrict__ ss, float slope, int C, size_t vox,
                                      float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over C * vox, voxel fastest
    if (i >= vox * C) return;
    const size_t v = i % vox;
    const int c = (int)(i / vox);
    float f = in[v * C + c];
    if (ss) {
        f = __builtin_fmaf(f, ss[c * 2], ss[c * 2 + 1]);
        f = f > 0.f ? f : f * slope;
    }
    out[i] = f;
}

__global__ void k_flip_accumulate(const float* __restrict__ src, float* __restrict__ dst, int P0, int P1, int P2, int flip, int add,
                                  float scale) {
    const size_t pv = (size_t)P0 * P1 * P2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pv) return;
    const int c = blockIdx.y;
    int z = (int)(i % P2), y = (int)((i / P2) % P1), x = (int)(i / ((size_t)P2 * P1));
    if (flip & 1) x = P0 - 1 - x;
    if (flip & 2) y = P1 - 1 - y;
    if (flip & 4) z = P2 - 1 - z;
    float v = src[(size_t)c * pv + ((size_t)x * P1 + y) * P2 + z];
    if (add) v = dst[(size_t)c * pv + i] + v;   // prediction += torch.flip(network(torch.flip(x, axes)), axes)   (fp32)
    dst[(size_t)c * pv + i] = v / scale;        // prediction /= (len(axes_combinations) + 1) on the last one (scale 1 before)
}

}  // namespace

int launch_flip_accumulate(boa_ctx* ctx, const float* src, float* dst, int C, const int P[3], int flip_mask, int add, float scale) {
    const size_t pv = (size_t)P[0] * P[1] * P[2];
    KernelTimer tm(ctx, BOA_K_HEAD_ACCUM, 0, (double)pv * C * (add ? 12.0 : 8.0));
    hipLaunchKernelGGL(k_flip_accumulate, dim3((unsigned)((pv + 255) / 256), C), dim3(256), 0, ctx->stream, src, dst, P[0], P[1], P[2],
                       flip_mask, add, scale);
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_ndhwc32_to_nchw_f32(boa_ctx* ctx, const float* in, const float* ss, float slope, int C, size_t vox, float* out) {
    hipLaunchKernelGGL(k_ndhwc32_to_nchw_f32, dim3((unsigned)((vox * C + 255) / 256)), dim3(256), 0, ctx->stream, in, ss, slope, C, vox, out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_conv_f32(boa_ctx* ctx, const float* src0, const float* ss0, int C0, const float* src1, const float* ss1, int C1, int N,
                    const int din[3], const int dout[3], const int k[3], const int s[3], int Cout, const float* w,
                    const float* bias, float slope, float* out) {
    BOA_REQUIRE(Cout % 32 == 0, "conv_f32: Cout=%d must be a multiple of 32", Cout);
    F32ConvArgs a;
    a.src0 = src0; a.src1 = src1; a.ss0 = ss0; a.ss1 = ss1; a.C0 = C0; a.C1 = C1;
    a.N = N; a.Di = din[0]; a.Hi = din[1]; a.Wi = din[2]; a.Do = dout[0]; a.Ho = dout[1]; a.Wo = dout[2]; a.Cout = Cout;
    a.k0 = k[0]; a.k1 = k[1]; a.k2 = k[2]; a.s0 = s[0]; a.s1 = s[1]; a.s2 = s[2];
    a.w = w; a.bias = bias; a.out = out; a.slope = slope;
    const size_t total = (size_t)N * dout[0] * dout[1] * dout[2];
    const double taps = (double)k[0] * k[1] * k[2];
    KernelTimer tm(ctx, BOA_K_CONV_MFMA, 2.0 * total * taps * (C0 + C1) * Cout, 4.0 * ((double)N * din[0] * din[1] * din[2] * (C0 + C1) + (double)total * Cout));
    hipLaunchKernelGGL(k_conv_f32, dim3((unsigned)((total + 127) / 128), Cout / 32), dim3(256), 0, ctx->stream, a);
    ctx->counters[BOA_CNT_F32]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_convt_f32(boa_ctx* ctx, const float* src, const float* ss, int Cin, int N, const int din[3], const int s[3], int Cout,
                     const float* w, const float* bias, float slope, float* out) {
    BOA_REQUIRE(Cout % 32 == 0, "convT_f32: Cout=%d must be a multiple of 32", Cout);
    F32ConvTArgs a;
    a.src = src; a.ss = ss; a.Cin = Cin; a.Cout = Cout; a.N = N; a.Di = din[0]; a.Hi = din[1]; a.Wi = din[2];
    a.s0 = s[0]; a.s1 = s[1]; a.s2 = s[2]; a.w = w; a.bias = bias; a.out = out; a.slope = slope;
    const size_t total = (size_t)N * din[0] * din[1] * din[2];
    const int taps = s[0] * s[1] * s[2];
    KernelTimer tm(ctx, BOA_K_CONVT, 2.0 * total * taps * Cin * Cout, 4.0 * total * (Cin + (double)taps * Cout));
    hipLaunchKernelGGL(k_convt_f32, dim3((unsigned)((total + 127) / 128), Cout / 32, taps), dim3(256), 0, ctx->stream, a);
    ctx->counters[BOA_CNT_F32]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_stats_f32(boa_ctx* ctx, const float* act, int N, size_t vox, int C, const float* gamma, const float* beta, float eps,
                     float* ss_out) {
    KernelTimer tm(ctx, BOA_K_NORM_FINALIZE, 0, 4.0 * N * (double)vox * C);
    hipLaunchKernelGGL(k_stats_f32, dim3(C, N), dim3(256), 0, ctx->stream, act, vox, C, gamma, beta, eps, ss_out);
    ctx->counters[BOA_CNT_F32]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_gather_tiles_f32(boa_ctx* ctx, const float* volume, const int V[3], const int vol_off[3], const int* dev_origins, int N,
                            int Cin, const int P[3], float* out, int flip_mask) {
    const size_t tot = (size_t)P[0] * P[1] * P[2] * Cin;
    KernelTimer tm(ctx, BOA_K_CONV_FIRST, 0, 8.0 * N * (double)tot);
    hipLaunchKernelGGL(k_gather_tiles_f32, dim3((unsigned)((tot + 255) / 256), N), dim3(256), 0, ctx->stream, volume, dev_origins,
                       V[0], V[1], V[2], vol_off ? vol_off[0] : 0, vol_off ? vol_off[1] : 0, vol_off ? vol_off[2] : 0, Cin, P[0], P[1],
                       P[2], flip_mask, out);
    ctx->counters[BOA_CNT_F32]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

int launch_head_f32(boa_ctx* ctx, const float* act, const float* ss, int F0, const int P[3], int C, const float* w,
                    const float* bias, float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc, uint16_t* nacc,
                    const int PV[3], const int start[3], size_t octet_stride) {
    BOA_REQUIRE(F0 <= 64, "head_f32: features[0]=%d unsupported (<= 64)", F0);
    F32HeadArgs a;
    a.act = act; a.ss = ss; a.F0 = F0; a.C = C; a.pv = (size_t)P[0] * P[1] * P[2]; a.P1 = P[1]; a.P2 = P[2];
    a.w = w; a.bias = bias; a.slope = slope; a.logits = logits_out; a.gauss = gauss; a.acc = acc; a.nacc = nacc;
    a.octet_stride = octet_stride;
    if (!logits_out) {
        for (int d = 0; d < 3; ++d)
            BOA_REQUIRE(start[d] >= 0 && start[d] + P[d] <= PV[d], "head_f32: tile [%d,%d) outside accumulator dim %d (%d)", start[d],
                        start[d] + P[d], d, PV[d]);
        a.V0 = PV[0]; a.V1 = PV[1]; a.V2 = PV[2]; a.s0 = start[0]; a.s1 = start[1]; a.s2 = start[2];
    } else {
        a.V0 = a.V1 = a.V2 = a.s0 = a.s1 = a.s2 = 0;
    }
    KernelTimer tm(ctx, BOA_K_HEAD_ACCUM, 2.0 * a.pv * F0 * C, (double)a.pv * (4.0 * F0 + (logits_out ? 4.0 * C : (4.0 * (C + 1) + 2.0))));
    hipLaunchKernelGGL(k_head_f32, dim3((unsigned)((a.pv + 255) / 256)), dim3(256), 0, ctx->stream, a);
    ctx->counters[octet_stride ? BOA_CNT_X3 : BOA_CNT_F32]++;
    tm.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
