// Socket power and clock of the bare matrix pipes: back-to-back v_mfma_f32_32x32x16_f16 (4 chains per wave, one wave per SIMD) for a
// few seconds per operand kind while tools/power_sample.sh samples rocm-smi.  Prints the sustained rate per phase.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_power tools/mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[4];
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
    for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r], 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool RANDOM>
static void phase(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, launches = 160;   // ~4 s
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k<RANDOM>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)launches * 256 * 4 * iters * 32.0 * 2.0 * 32 * 32 * 16;
    printf("%s: %.1f TFLOP/s over %.2f s\n", name, fl / (ms * 1e-3) / 1e12, ms * 1e-3);
    fflush(stdout);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    sleep(2);
    phase<false>("constant operands", d);
    sleep(2);
    phase<true>("random operands", d);
    sleep(2);
    return 0;
}
