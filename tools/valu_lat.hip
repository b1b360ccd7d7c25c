// VALU issue/latency microbenchmark for one wave on a SIMD (gfx950): dependent vs independent chains, fp32 fma and
// packed fp16 fma, with and without a second wave on the same SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int CH, bool PK>
__global__ void k(float* out, long long* cyc, int iters) {
    float x[CH]; h2 y[CH];
    for (int i = 0; i < CH; ++i) { x[i] = threadIdx.x * 0.5f + i; y[i] = h2{(_Float16)(threadIdx.x), (_Float16)i}; }
    const float a = 1.0001f, b = 0.5f; const h2 ha = h2{(_Float16)1.001f, (_Float16)0.999f}, hb = h2{(_Float16)0.5f, (_Float16)0.25f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (PK) y[c] = __builtin_elementwise_fma(y[c], ha, hb);
                else x[c] = __builtin_fmaf(x[c], a, b);
            }
    }
    long long t1 = clock64();
    float s = 0; for (int c = 0; c < CH; ++c) s += x[c] + (float)y[c][0] + (float)y[c][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH, bool PK> void run(int threads) {
    float* d; long long* c; hipMalloc(&d, 4 * 4096); hipMalloc(&c, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<CH, PK>), dim3(1), dim3(threads), 0, 0, d, c, iters); hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%s chains=%d waves=%d: %.2f clk64-ticks per instr per wave\n", PK ? "pk_fma_f16" : "fma_f32", CH, threads / 64, (double)h / (iters * 16.0 * CH));
}
int main() {
    run<1, false>(64); run<2, false>(64); run<4, false>(64); run<8, false>(64);
    run<1, true>(64); run<4, true>(64);
    run<1, false>(256); run<1, false>(512); run<4, false>(512); run<1, false>(1024);
    return 0;
}
