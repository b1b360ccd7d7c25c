// Diagnostics: what the matrix cores of THIS part sustain with no memory traffic at all (boa_mfma_peak).  bench.py reports it next to
// the conv kernel's rate: the dense fp16 peak of the data sheet (2.5 PFLOP/s at 2.4 GHz) is not attainable under power -- the clock
// the part holds depends on how many operand bits toggle.
#include "common.h"

typedef _Float16 dg_f16x8 __attribute__((ext_vector_type(8)));
typedef float dg_f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256) void k_mfma_peak(float* out, int iters) {
    constexpr int R = 4;  // independent accumulator chains per wave
    dg_f32x16 acc[R];
    dg_f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.001f + i);
        b[i] = (_Float16)(i * 0.5f);
    }
    if (RANDOM) {  // operand bits differ from lane to lane and element to element (values in [-2, 2] like normalised activations)
        unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            a[i] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 0.002f);
            h = h * 1664525u + 1013904223u;
            b[i] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 0.002f);
        }
    }
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r], 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < 16; ++i) s += acc[r][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int boa_mfma_peak(boa_ctx* c, int random_operands, int iters, double* tflops_out) {
    BOA_REQUIRE(c && tflops_out && iters > 0 && iters <= (1 << 22), "boa_mfma_peak: bad argument");
    const int blocks = c->cu_count, threads = 256;  // one 4-wave workgroup per CU: one wave per SIMD, 4 chains each
    float* d = nullptr;
    BOA_TRY(boa_malloc(c, (size_t)blocks * threads * 4, (void**)&d));
    hipEvent_t e0, e1;
    BOA_HIP_TRY(hipEventCreate(&e0));
    BOA_HIP_TRY(hipEventCreate(&e1));
    auto launch = [&](int n) {
        if (random_operands)
            hipLaunchKernelGGL(k_mfma_peak<true>, dim3(blocks), dim3(threads), 0, c->stream, d, n);
        else
            hipLaunchKernelGGL(k_mfma_peak<false>, dim3(blocks), dim3(threads), 0, c->stream, d, n);
    };
    for (int w = 0; w < 4; ++w) launch(iters);  // let the power management settle (tens of milliseconds) before timing
    hipEventRecord(e0, c->stream);
    launch(iters);
    hipEventRecord(e1, c->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    boa_free(c, d);
    BOA_HIP_TRY(e);
    *tflops_out = (double)blocks * (threads / 64) * (double)iters * 8.0 * 4.0 * 32768.0 / ((double)ms * 1e-3) / 1e12;
    return BOA_OK;
}
