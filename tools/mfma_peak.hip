// Attainable MFMA rate on this GPU: R independent v_mfma_f32_32x32x16_f16 chains per wave, W waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM: operands with random bits per lane (N(0,1)-like magnitudes) instead of near-constant ones: the matrix pipes' power
// draw -- and with it the clock the part sustains -- depends on how many operand bits toggle
template <int R, bool RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc[R];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    if (RANDOM) {
        unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            a[i] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 0.002f);
            h = h * 1664525u + 1013904223u;
            b[i] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 0.002f);
        }
    }
    for (int r = 0; r < R; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r], 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < R; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, bool RANDOM = false>
void run(int threads, int blocks) {
    float* d; hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<R, RANDOM>), dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, RANDOM>), dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * (threads / 64) * iters * 8.0 * R * 32768.0;
    printf("R=%d waves/CU=%d blocks=%d %s operands: %.1f TFLOP/s (%.2f ms)\n", R, threads / 64, blocks, RANDOM ? "random" : "constant",
           flop / ms * 1e-9, ms);
    hipFree(d);
}
int main() {
    run<1>(256, 256); run<2>(256, 256); run<4>(256, 256); run<4>(512, 256); run<4>(256, 512); run<1>(512, 256);
    run<4, true>(256, 256); run<4, true>(512, 256);
    // a longer run (the clock settles after a few hundred ms)
    for (int i = 0; i < 3; ++i) run<4, true>(256, 256);
    return 0;
}
