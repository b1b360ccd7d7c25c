// Resampling of TS/resampling.change_spacing on the device (fp64):
//   order 3: scipy.ndimage.zoom(data, zoom, order=3, mode="nearest") restated -- edge-pad by 12, separable cubic
//            B-spline prefilter (pole sqrt(3)-2, gain 6, 'reflect' boundary initialisation, which is what scipy uses for
//            mode="nearest"), then the 4x4x4 tap interpolation with coordinate in = out * (n_in-1)/(n_out-1) inside
//            the unpadded extent, terms accumulated in scipy's order (first axis outermost), `.astype(int32)`
//            truncation (TS/resampling.py:36-37,211,216-217);
//   order 0: nearest gather, index floor(in + 0.5) clamped.
// Built with -ffp-contract=off; every fp64 operation is an IEEE add/mul/div in scipy's order, the pole is the constant
// scipy's C compiler folds `sqrt(3.0) - 2.0` to (correctly rounded: -0x1.126145e9ecd56p-2, 2 ulp from the run-time
// double expression) and pow(z, n) is taken on the host (libm, correctly rounded) -- with that the result is
// bit-identical to scipy 1.15.3 on every test vector, including the int32 truncation of golden G5.
#include <math.h>

#include "common.h"

#define NPAD 12

template <typename T>
__global__ __launch_bounds__(256) void k_pad_edge_f64(const T* __restrict__ in, int X, int Y, int Z,
                                                      double* __restrict__ out) {
    const int PX = X + 2 * NPAD, PY = Y + 2 * NPAD, PZ = Z + 2 * NPAD;
    const size_t n = (size_t)PX * PY * PZ;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    idx3(i, PY, PZ, x, y, z);
    x -= NPAD; y -= NPAD; z -= NPAD;
    x = min(max(x, 0), X - 1);
    y = min(max(y, 0), Y - 1);
    z = min(max(z, 0), Z - 1);
    out[i] = (double)in[((size_t)x * Y + y) * Z + z];
}

// one thread per line along `axis` (length n, element stride `st`); lines enumerated over the other two axes
struct SplineConsts {
    double z, gain, z_n, init_scale, tail_scale;  // pole, (1-z)(1-1/z), z^n, z/(1-z_n^2), z/(z-1)
};

__global__ __launch_bounds__(64) void k_spline_filter_axis(double* __restrict__ c, int n, size_t st, int n1, size_t st1,
                                                           int n2, size_t st2, SplineConsts k) {
    const size_t line = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (line >= (size_t)n1 * n2) return;
    double* p = c + (line / n2) * st1 + (line % n2) * st2;
    const double z = k.z;
    const double gain = k.gain;
    for (int i = 0; i < n; ++i) p[i * st] *= gain;
    // causal initialisation, 'reflect' boundary
    const double z_n = k.z_n;
    double z_i = z;
    const double c0 = p[0];
    double acc = p[0] + z_n * p[(size_t)(n - 1) * st];
    for (int i = 1; i < n; ++i) {
        acc += z_i * (p[i * st] + z_n * p[(size_t)(n - 1 - i) * st]);
        z_i *= z;
    }
    acc *= k.init_scale;
    acc += c0;
    p[0] = acc;
    for (int i = 1; i < n; ++i) p[i * st] += z * p[(size_t)(i - 1) * st];
    p[(size_t)(n - 1) * st] *= k.tail_scale;
    for (int i = n - 2; i >= 0; --i) p[i * st] = z * (p[(size_t)(i + 1) * st] - p[i * st]);
}

// The same filter for lines along the CONTIGUOUS axis (element stride 1): one thread per line again, but every lane moves its
// line in 64-byte pieces (8 doubles = one full cache line per access) and runs the recursions on registers.  With one 8-byte
// access per step (above) neighbouring lanes touch addresses a whole line apart, every 64-byte line is fetched up to eight
// times and each step waits for memory: 13 ms for a 536^3 volume against ~2 ms for the other two axes.  The arithmetic is the
// same sequence of fp64 operations (the scaled value p[i] * gain is recomputed where the in-place version re-read it: the same
// double), so the results are bit-identical.
__global__ __launch_bounds__(64) void k_spline_filter_contig(double* __restrict__ c, int n, int n1, size_t st1, int n2, size_t st2,
                                                             SplineConsts k) {
    const size_t line = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (line >= (size_t)n1 * n2) return;
    double* p = c + (line / n2) * st1 + (line % n2) * st2;
    const double z = k.z, gain = k.gain, z_n = k.z_n;
    const int nc = n / 8;  // whole 8-element pieces
    // ---- causal initialisation: acc = p0 + z_n p[n-1] + sum_{i>=1} z^i (p[i] + z_n p[n-1-i]), all p scaled by gain
    const double c0 = p[0] * gain;
    double acc = c0 + z_n * (p[n - 1] * gain);
    double z_i = z;
    for (int cb = 0; cb < nc; ++cb) {
        double f[8], b[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = p[8 * cb + m];
        const int b0 = n - 8 * cb - 8;  // b[m] = p[b0 + m] <-> i = 8 cb + 7 - m
#pragma unroll
        for (int m = 0; m < 8; ++m) b[m] = p[b0 + m];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int i = 8 * cb + m;
            if (i >= 1) {
                acc += z_i * (f[m] * gain + z_n * (b[7 - m] * gain));
                z_i *= z;
            }
        }
    }
    for (int i = max(8 * nc, 1); i < n; ++i) {
        acc += z_i * (p[i] * gain + z_n * (p[n - 1 - i] * gain));
        z_i *= z;
    }
    acc *= k.init_scale;
    acc += c0;
    // ---- causal pass: p[0] = acc, p[i] = gain p[i] + z p[i-1]
    double prev = acc;
    for (int cb = 0; cb < nc; ++cb) {
        double f[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = p[8 * cb + m];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (8 * cb + m >= 1) {
                double s = f[m] * gain;
                s += z * prev;
                prev = s;
            }
            f[m] = prev;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) p[8 * cb + m] = f[m];
    }
    for (int i = 8 * nc; i < n; ++i) {
        if (i >= 1) {
            double s = p[i] * gain;
            s += z * prev;
            prev = s;
        }
        p[i] = prev;
    }
    // ---- anticausal pass: p[n-1] *= tail_scale, p[i] = z (p[i+1] - p[i])
    double nxt = p[n - 1] * k.tail_scale;
    p[n - 1] = nxt;
    int i = n - 2;
    for (; i >= 0 && ((i + 1) & 7) != 0; --i) {  // down to an 8-aligned piece boundary
        nxt = z * (nxt - p[i]);
        p[i] = nxt;
    }
    for (; i >= 7; i -= 8) {  // pieces [i - 7, i]
        double f[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = p[i - 7 + m];
#pragma unroll
        for (int m = 7; m >= 0; --m) {
            nxt = z * (nxt - f[m]);
            f[m] = nxt;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) p[i - 7 + m] = f[m];
    }
}

__device__ __forceinline__ void cubic_weights(double cc, int* start, double w[4]) {
    const double fl = floor(cc);
    const double x = cc - fl;
    const double y = x, zz = 1.0 - x;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = zz * zz * zz / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
    *start = (int)fl - 1;
}

// out_mode: 0 = int32 (C truncation of the fp64 value), 1 = float64
__global__ __launch_bounds__(256) void k_zoom_cubic(const double* __restrict__ coef, int X, int Y, int Z, int OX, int OY,
                                                    int OZ, double zx, double zy, double zz_, int out_mode, void* out) {
    const size_t n = (size_t)OX * OY * OZ;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int ox, oy, oz;
    idx3(i, OY, OZ, ox, oy, oz);
    const int PY = Y + 2 * NPAD, PZ = Z + 2 * NPAD;
    double wx[4], wy[4], wz[4];
    int sx, sy, sz;
    // no clamp to [0, n_in - 1]: the last coordinate (n_out - 1) * fl((n_in - 1) / (n_out - 1)) can land one ulp above n_in - 1
    // (e.g. 42 * (46 / 42) = 46.00000000000001); scipy evaluates the spline there inside its 12-sample edge padding, and so do we
    cubic_weights((double)ox * zx + NPAD, &sx, wx);
    cubic_weights((double)oy * zy + NPAD, &sy, wy);
    cubic_weights((double)oz * zz_ + NPAD, &sz, wz);
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double* row = coef + ((size_t)(sx + a) * PY + (sy + b)) * PZ + sz;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                double cf = row[d];
                cf *= wx[a];
                cf *= wy[b];
                cf *= wz[d];
                t += cf;
            }
        }
    if (out_mode == 0)
        ((int*)out)[i] = (int)t;
    else
        ((double*)out)[i] = t;
}

__global__ __launch_bounds__(256) void k_zoom_nearest_u8(const unsigned char* __restrict__ in, int X, int Y, int Z, int OX,
                                                         int OY, int OZ, double zx, double zy, double zz_,
                                                         unsigned char* __restrict__ out) {
    const size_t n = (size_t)OX * OY * OZ;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int ox, oy, oz;
    idx3(i, OY, OZ, ox, oy, oz);
    const int ix = min(max((int)floor((double)ox * zx + 0.5), 0), X - 1);
    const int iy = min(max((int)floor((double)oy * zy + 0.5), 0), Y - 1);
    const int iz = min(max((int)floor((double)oz * zz_ + 0.5), 0), Z - 1);
    out[i] = in[((size_t)ix * Y + iy) * Z + iz];
}

static SplineConsts spline_consts(int n) {
    SplineConsts k;
    k.z = -0x1.126145e9ecd56p-2;
    k.gain = (1.0 - 1.0 / k.z) * (1.0 - k.z);
    k.z_n = pow(k.z, (double)n);
    k.init_scale = k.z / (1.0 - k.z_n * k.z_n);
    k.tail_scale = k.z / (k.z - 1.0);
    return k;
}

static double zoom_factor(int n_in, int n_out) { return n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 1.0; }

extern "C" int boa_resample_cubic(boa_ctx* c, const void* dev_in, int in_dtype, const int in_dims[3], void* dev_out,
                                  int out_dtype, const int out_dims[3]) {
    BOA_REQUIRE(c && dev_in && dev_out && in_dims && out_dims, "boa_resample_cubic: NULL argument");
    BOA_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "boa_resample_cubic: in_dtype %d (0 int16, 1 float32, 2 float64, 3 int32)", in_dtype);
    BOA_REQUIRE(out_dtype == 0 || out_dtype == 1, "boa_resample_cubic: out_dtype %d (0 int32, 1 float64)", out_dtype);
    const int X = in_dims[0], Y = in_dims[1], Z = in_dims[2];
    for (int a = 0; a < 3; ++a) BOA_REQUIRE(in_dims[a] >= 2 && out_dims[a] >= 1, "boa_resample_cubic: bad dims");
    const int PX = X + 2 * NPAD, PY = Y + 2 * NPAD, PZ = Z + 2 * NPAD;
    const size_t pn = (size_t)PX * PY * PZ;
    double* coef = nullptr;
    BOA_TRY(boa_malloc(c, pn * sizeof(double), (void**)&coef));
    const unsigned gp = (unsigned)((pn + 255) / 256);
    KernelTimer t(c, BOA_K_RESAMPLE, 0, (double)pn * 8.0 * 8);
    switch (in_dtype) {
        case 0: hipLaunchKernelGGL(k_pad_edge_f64<short>, dim3(gp), dim3(256), 0, c->stream, (const short*)dev_in, X, Y, Z, coef); break;
        case 1: hipLaunchKernelGGL(k_pad_edge_f64<float>, dim3(gp), dim3(256), 0, c->stream, (const float*)dev_in, X, Y, Z, coef); break;
        case 2: hipLaunchKernelGGL(k_pad_edge_f64<double>, dim3(gp), dim3(256), 0, c->stream, (const double*)dev_in, X, Y, Z, coef); break;
        default: hipLaunchKernelGGL(k_pad_edge_f64<int>, dim3(gp), dim3(256), 0, c->stream, (const int*)dev_in, X, Y, Z, coef); break;
    }
    // scipy filters axis 0 first, then 1, then 2
    hipLaunchKernelGGL(k_spline_filter_axis, dim3((unsigned)(((size_t)PY * PZ + 63) / 64)), dim3(64), 0, c->stream, coef, PX,
                       (size_t)PY * PZ, PY, (size_t)PZ, PZ, (size_t)1, spline_consts(PX));
    hipLaunchKernelGGL(k_spline_filter_axis, dim3((unsigned)(((size_t)PX * PZ + 63) / 64)), dim3(64), 0, c->stream, coef, PY,
                       (size_t)PZ, PX, (size_t)PY * PZ, PZ, (size_t)1, spline_consts(PY));
    hipLaunchKernelGGL(k_spline_filter_contig, dim3((unsigned)(((size_t)PX * PY + 63) / 64)), dim3(64), 0, c->stream, coef, PZ, PX,
                       (size_t)PY * PZ, PY, (size_t)PZ, spline_consts(PZ));
    const size_t on = (size_t)out_dims[0] * out_dims[1] * out_dims[2];
    hipLaunchKernelGGL(k_zoom_cubic, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, c->stream, coef, X, Y, Z, out_dims[0],
                       out_dims[1], out_dims[2], zoom_factor(X, out_dims[0]), zoom_factor(Y, out_dims[1]),
                       zoom_factor(Z, out_dims[2]), out_dtype, dev_out);
    t.stop();
    hipError_t e = hipGetLastError();
    int rc = boa_free(c, coef);  // synchronises the stream
    BOA_HIP_TRY(e);
    return rc;
}

extern "C" int boa_resample_nearest_u8(boa_ctx* c, const uint8_t* dev_in, const int in_dims[3], uint8_t* dev_out,
                                       const int out_dims[3]) {
    BOA_REQUIRE(c && dev_in && dev_out && in_dims && out_dims, "boa_resample_nearest_u8: NULL argument");
    for (int a = 0; a < 3; ++a) BOA_REQUIRE(in_dims[a] >= 1 && out_dims[a] >= 1, "boa_resample_nearest_u8: bad dims");
    const size_t on = (size_t)out_dims[0] * out_dims[1] * out_dims[2];
    KernelTimer t(c, BOA_K_RESAMPLE, 0, (double)on * 2.0);
    hipLaunchKernelGGL(k_zoom_nearest_u8, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, c->stream, dev_in, in_dims[0],
                       in_dims[1], in_dims[2], out_dims[0], out_dims[1], out_dims[2], zoom_factor(in_dims[0], out_dims[0]),
                       zoom_factor(in_dims[1], out_dims[1]), zoom_factor(in_dims[2], out_dims[2]), dev_out);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}


// ======================================================================================================
// nnU-Net's own resampling between the image grid and the plans' spacing (NN/preprocessing/resampling/
// default_resampling.py:113-196): skimage.transform.resize(mode="edge", anti_aliasing=False, clip=True) -- published
// algorithm: scipy.ndimage.zoom(grid_mode=True, mode="nearest") + clip to the input's range -- either on the whole 3-D array
// or, for anisotropic spacings ("separate z"), slice by slice in 2-D with nearest sampling along the slice axis.
// Same fp64 operation order as the kernels above; grid_mode coordinates cc = (k + 0.5) * (n_in / n_out) - 0.5.

__device__ __forceinline__ unsigned f2ord(float f) {  // order-preserving float -> uint (atomicMin / atomicMax on floats)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// min / max per group: group = index along `gaxis` (separate z: one slice each) or the whole volume (gaxis < 0)
__global__ __launch_bounds__(256) void k_minmax_groups(const float* __restrict__ in, int X, int Y, int Z, int gaxis,
                                                       unsigned* __restrict__ mm /*[groups][2], pre-set to {~0, 0}*/) {
    const size_t n = (size_t)X * Y * Z;
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    // 8 consecutive voxels along the contiguous axis; a group change inside the run can only happen when gaxis == 2
    int cur = -1;
    float lo = 0.f, hi = 0.f;
    for (int k = 0; k < 8 && i0 + k < n; ++k) {
        const size_t i = i0 + k;
        const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((size_t)Z * Y));
        const int g = gaxis < 0 ? 0 : (gaxis == 0 ? x : (gaxis == 1 ? y : z));
        const float v = in[i];
        if (g != cur) {
            if (cur >= 0) {
                atomicMin(&mm[2 * cur], f2ord(lo));
                atomicMax(&mm[2 * cur + 1], f2ord(hi));
            }
            cur = g;
            lo = hi = v;
        } else {
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    if (cur >= 0) {
        atomicMin(&mm[2 * cur], f2ord(lo));
        atomicMax(&mm[2 * cur + 1], f2ord(hi));
    }
}

// edge padding by NPAD on the resized axes only
__global__ __launch_bounds__(256) void k_pad_edge_axes_f64(const float* __restrict__ in, int X, int Y, int Z, int px, int py, int pz,
                                                           double* __restrict__ out) {
    const int PX = X + 2 * px, PY = Y + 2 * py, PZ = Z + 2 * pz;
    const size_t n = (size_t)PX * PY * PZ;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    idx3(i, PY, PZ, x, y, z);
    x -= px; y -= py; z -= pz;
    x = min(max(x, 0), X - 1);
    y = min(max(y, 0), Y - 1);
    z = min(max(z, 0), Z - 1);
    out[i] = (double)in[((size_t)x * Y + y) * Z + z];
}

struct ResizeArgs {
    int I[3], P[3], O[3];  // input dims, padded dims, output dims
    int act[3];            // axis is resized (spline) / is the slice axis (nearest)
    double zf[3];          // n_in / n_out
    int gaxis;             // clip group axis (-1: whole volume)
};

// nearest source index along the slice axis: scipy map_coordinates(order=0, mode="nearest") at scale * (o + 0.5) - 0.5
__device__ __forceinline__ int nearest_src(int o, double scale, int n_in) {
    const double cc = scale * ((double)o + 0.5) - 0.5;
    return min(max((int)floor(cc + 0.5), 0), n_in - 1);
}

__global__ __launch_bounds__(256) void k_resize_cubic_f32(const double* __restrict__ coef, ResizeArgs a, const unsigned* __restrict__ mm,
                                                          float* __restrict__ out) {
    const size_t n = (size_t)a.O[0] * a.O[1] * a.O[2];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int o[3];
    idx3(i, a.O[1], a.O[2], o[0], o[1], o[2]);
    double w[3][4];
    int st[3], nt[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (a.act[d]) {
            cubic_weights(((double)o[d] + 0.5) * a.zf[d] - 0.5 + NPAD, &st[d], w[d]);
            nt[d] = 4;
        } else {
            st[d] = a.I[d] == a.O[d] ? o[d] : nearest_src(o[d], a.zf[d], a.I[d]);
            nt[d] = 1;
        }
    }
    double t = 0.0;
    for (int p = 0; p < nt[0]; ++p)
        for (int q = 0; q < nt[1]; ++q) {
            const double* row = coef + ((size_t)(st[0] + p) * a.P[1] + (st[1] + q)) * a.P[2] + st[2];
            for (int r = 0; r < nt[2]; ++r) {
                double cf = row[r];
                if (a.act[0]) cf *= w[0][p];
                if (a.act[1]) cf *= w[1][q];
                if (a.act[2]) cf *= w[2][r];
                t += cf;
            }
        }
    // clip to the range of the input (of the source slice in separate-z mode), then `reshaped_final[c] = ...` (float64 -> float32)
    const int g = a.gaxis < 0 ? 0 : st[a.gaxis];
    const double lo = (double)ord2f(mm[2 * g]), hi = (double)ord2f(mm[2 * g + 1]);
    t = t < lo ? lo : (t > hi ? hi : t);
    out[i] = (float)t;
}

// double -> half with ONE rounding (numpy's npy_double_to_half): to float with round-to-odd, then RTNE to half
__device__ __forceinline__ unsigned short d2h_rn(double d) {
    float f = __double2float_rz(d);
    if ((double)f != d) f = __uint_as_float(__float_as_uint(f) | 1u);
    return f2us(f);
}

struct LogitsResizeArgs {
    const unsigned short* logits;  // fp16 [C][L0][L1][L2]
    int C, L[3], off[3], I[3], O[3];  // full grid, crop origin, crop dims (= resampled shape), output dims
    int act[3];
    double zf[3];
    int merge;
    unsigned char lut[256];
};

// resampling_fn_probabilities (order 1) of every class at one output voxel + fp16 rounding + numpy argmax + lut / merge
__global__ __launch_bounds__(256) void k_resize_logits_argmax(LogitsResizeArgs a, unsigned char* __restrict__ labels) {
    const size_t n = (size_t)a.O[0] * a.O[1] * a.O[2];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int o[3];
    idx3(i, a.O[1], a.O[2], o[0], o[1], o[2]);
    double w[3][2];
    int i0[3], i1[3], nt[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (a.act[d]) {
            const double cc = ((double)o[d] + 0.5) * a.zf[d] - 0.5;  // not clamped: the tap INDICES are (mode "nearest")
            const double fl = floor(cc);
            const double x = cc - fl;
            w[d][0] = 1.0 - x;
            w[d][1] = x;
            i0[d] = min(max((int)fl, 0), a.I[d] - 1);
            i1[d] = min(max((int)fl + 1, 0), a.I[d] - 1);
            nt[d] = 2;
        } else {
            i0[d] = i1[d] = a.I[d] == a.O[d] ? o[d] : nearest_src(o[d], a.zf[d], a.I[d]);
            nt[d] = 1;
        }
    }
    const size_t lv = (size_t)a.L[0] * a.L[1] * a.L[2];
    float best = 0.f;
    int bidx = 0;
    bool bnan = false;
    for (int c = 0; c < a.C; ++c) {
        const unsigned short* base = a.logits + (size_t)c * lv;
        double t = 0.0;
        for (int p = 0; p < nt[0]; ++p)
            for (int q = 0; q < nt[1]; ++q)
                for (int r = 0; r < nt[2]; ++r) {
                    const int x = (p ? i1[0] : i0[0]) + a.off[0], y = (q ? i1[1] : i0[1]) + a.off[1], z = (r ? i1[2] : i0[2]) + a.off[2];
                    double cf = (double)us2f(base[((size_t)x * a.L[1] + y) * a.L[2] + z]);
                    if (a.act[0]) cf *= w[0][p];
                    if (a.act[1]) cf *= w[1][q];
                    if (a.act[2]) cf *= w[2][r];
                    t += cf;
                }
        const float f = us2f(d2h_rn(t));  // `reshaped_final` has the logits' dtype: float16
        const bool isn = f != f;
        if (c == 0) {
            best = f;
            bnan = isn;
        } else if (!bnan && (isn || f > best)) {
            best = f;
            bidx = c;
            bnan = isn;
        }
    }
    if (a.merge) {
        if (bidx != 0) labels[i] = a.lut[bidx];
    } else {
        labels[i] = a.lut[bidx];
    }
}

static int resize_axes(const int in_dims[3], const int out_dims[3], int slice_axis, ResizeArgs* a) {
    for (int d = 0; d < 3; ++d) {
        a->I[d] = in_dims[d];
        a->O[d] = out_dims[d];
        a->act[d] = d != slice_axis;
        a->P[d] = in_dims[d] + (a->act[d] ? 2 * NPAD : 0);
        a->zf[d] = (double)in_dims[d] / (double)out_dims[d];
    }
    a->gaxis = slice_axis;
    return BOA_OK;
}

extern "C" int boa_resize_skimage_f32(boa_ctx* c, const float* dev_in, const int in_dims[3], float* dev_out, const int out_dims[3],
                                      int order, int slice_axis) {
    BOA_REQUIRE(c && dev_in && dev_out && in_dims && out_dims, "boa_resize_skimage_f32: NULL argument");
    BOA_REQUIRE(order == 3, "boa_resize_skimage_f32: order %d (image data is resampled with order 3)", order);
    BOA_REQUIRE(slice_axis >= -1 && slice_axis <= 2, "boa_resize_skimage_f32: slice_axis %d", slice_axis);
    for (int a = 0; a < 3; ++a) BOA_REQUIRE(in_dims[a] >= 1 && out_dims[a] >= 1, "boa_resize_skimage_f32: bad dims");
    for (int a = 0; a < 3; ++a) BOA_REQUIRE(a == slice_axis || in_dims[a] >= 2, "boa_resize_skimage_f32: a resized axis needs >= 2 samples");
    ResizeArgs ra;
    resize_axes(in_dims, out_dims, slice_axis, &ra);
    const size_t pn = (size_t)ra.P[0] * ra.P[1] * ra.P[2];
    const size_t in_n = (size_t)in_dims[0] * in_dims[1] * in_dims[2];
    const int groups = slice_axis < 0 ? 1 : in_dims[slice_axis];
    double* coef = nullptr;
    unsigned* mm = nullptr;
    BOA_TRY(boa_malloc(c, pn * sizeof(double), (void**)&coef));
    int rc = boa_malloc(c, (size_t)groups * 2 * sizeof(unsigned), (void**)&mm);
    if (rc) {
        boa_free(c, coef);
        return rc;
    }
    std::vector<unsigned> init((size_t)groups * 2);
    for (int g = 0; g < groups; ++g) {
        init[2 * g] = 0xffffffffu;
        init[2 * g + 1] = 0u;
    }
    hipMemcpyAsync(mm, init.data(), init.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream);
    hipStreamSynchronize(c->stream);  // `init` is a host temporary
    c->prof_break = true;
    KernelTimer t(c, BOA_K_RESAMPLE, 0, (double)pn * 8.0 * 6 + (double)in_n * 8.0);
    hipLaunchKernelGGL(k_minmax_groups, dim3((unsigned)((in_n / 8 + 256) / 256)), dim3(256), 0, c->stream, dev_in, in_dims[0], in_dims[1],
                       in_dims[2], slice_axis, mm);
    hipLaunchKernelGGL(k_pad_edge_axes_f64, dim3((unsigned)((pn + 255) / 256)), dim3(256), 0, c->stream, dev_in, in_dims[0], in_dims[1],
                       in_dims[2], (ra.P[0] - ra.I[0]) / 2, (ra.P[1] - ra.I[1]) / 2, (ra.P[2] - ra.I[2]) / 2, coef);
    // scipy filters the axes in ascending order; the slice axis of the separate-z mode is not filtered
    const size_t s0 = (size_t)ra.P[1] * ra.P[2], s1 = (size_t)ra.P[2], s2 = 1;
    if (ra.act[0])
        hipLaunchKernelGGL(k_spline_filter_axis, dim3((unsigned)(((size_t)ra.P[1] * ra.P[2] + 63) / 64)), dim3(64), 0, c->stream, coef, ra.P[0],
                           s0, ra.P[1], s1, ra.P[2], s2, spline_consts(ra.P[0]));
    if (ra.act[1])
        hipLaunchKernelGGL(k_spline_filter_axis, dim3((unsigned)(((size_t)ra.P[0] * ra.P[2] + 63) / 64)), dim3(64), 0, c->stream, coef, ra.P[1],
                           s1, ra.P[0], s0, ra.P[2], s2, spline_consts(ra.P[1]));
    if (ra.act[2])
        hipLaunchKernelGGL(k_spline_filter_contig, dim3((unsigned)(((size_t)ra.P[0] * ra.P[1] + 63) / 64)), dim3(64), 0, c->stream, coef,
                           ra.P[2], ra.P[0], s0, ra.P[1], s1, spline_consts(ra.P[2]));
    const size_t on = (size_t)out_dims[0] * out_dims[1] * out_dims[2];
    hipLaunchKernelGGL(k_resize_cubic_f32, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, c->stream, coef, ra, mm, dev_out);
    t.stop();
    hipError_t e = hipGetLastError();
    rc = boa_free(c, coef);
    boa_free(c, mm);
    BOA_HIP_TRY(e);
    return rc;
}

extern "C" int boa_resize_logits_argmax(boa_ctx* c, const uint16_t* dev_logits, int C, const int grid_dims[3], const int* crop_off,
                                        const int* crop_dims, const int out_dims[3], int slice_axis, const uint8_t* host_lut,
                                        int merge, uint8_t* dev_labels_out) {
    BOA_REQUIRE(c && dev_logits && grid_dims && out_dims && dev_labels_out, "boa_resize_logits_argmax: NULL argument");
    BOA_REQUIRE(C >= 1 && C <= 255, "boa_resize_logits_argmax: C=%d out of range", C);
    BOA_REQUIRE(slice_axis >= -1 && slice_axis <= 2, "boa_resize_logits_argmax: slice_axis %d", slice_axis);
    LogitsResizeArgs a;
    a.logits = dev_logits;
    a.C = C;
    a.merge = merge;
    for (int d = 0; d < 3; ++d) {
        a.L[d] = grid_dims[d];
        a.off[d] = crop_off ? crop_off[d] : 0;
        a.I[d] = crop_dims ? crop_dims[d] : grid_dims[d];
        a.O[d] = out_dims[d];
        BOA_REQUIRE(a.off[d] >= 0 && a.I[d] >= 1 && a.off[d] + a.I[d] <= a.L[d] && a.O[d] >= 1, "boa_resize_logits_argmax: bad geometry on axis %d", d);
        a.act[d] = d != slice_axis;
        a.zf[d] = (double)a.I[d] / (double)a.O[d];
    }
    for (int i = 0; i < 256; ++i) a.lut[i] = host_lut ? host_lut[i] : (unsigned char)i;
    const size_t on = (size_t)out_dims[0] * out_dims[1] * out_dims[2];
    KernelTimer t(c, BOA_K_ARGMAX, 0, (double)on * (2.0 * C + 1.0));
    hipLaunchKernelGGL(k_resize_logits_argmax, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, c->stream, a, dev_labels_out);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
