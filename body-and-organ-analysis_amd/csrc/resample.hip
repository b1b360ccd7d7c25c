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
    int z = (int)(i % PZ) - NPAD, y = (int)((i / PZ) % PY) - NPAD, x = (int)(i / ((size_t)PZ * PY)) - NPAD;
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
    const int oz = (int)(i % OZ), oy = (int)((i / OZ) % OY), ox = (int)(i / ((size_t)OZ * OY));
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
    const int oz = (int)(i % OZ), oy = (int)((i / OZ) % OY), ox = (int)(i / ((size_t)OZ * OY));
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
    KernelTimer t(c, BOA_K_OTHER, 0, (double)pn * 8.0 * 8);
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
    hipLaunchKernelGGL(k_spline_filter_axis, dim3((unsigned)(((size_t)PX * PY + 63) / 64)), dim3(64), 0, c->stream, coef, PZ,
                       (size_t)1, PX, (size_t)PY * PZ, PY, (size_t)PZ, spline_consts(PZ));
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
    KernelTimer t(c, BOA_K_OTHER, 0, (double)on * 2.0);
    hipLaunchKernelGGL(k_zoom_nearest_u8, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, c->stream, dev_in, in_dims[0],
                       in_dims[1], in_dims[2], out_dims[0], out_dims[1], out_dims[2], zoom_factor(in_dims[0], out_dims[0]),
                       zoom_factor(in_dims[1], out_dims[1]), zoom_factor(in_dims[2], out_dims[2]), dev_out);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
