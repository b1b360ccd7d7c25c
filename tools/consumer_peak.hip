// How fast can a wave run the k_conv_ws consumer loop (LDS fragment reads + MFMA) in isolation?
// W waves per CU, R M-tiles per wave, per tap: 1 A read + R B reads (ds_read_b128) + R MFMAs, prefetch distance 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int NB>   // NB: B reads per tap actually issued (R = all, 0 = none -> registers)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u;
    __syncthreads();
    const unsigned char* ap = smem + (lane) * 16;
    const unsigned char* bp[R];
    for (int r = 0; r < R; ++r) bp[r] = smem + 32768 + ((wave * R + r) & 15) * 2048 + (lane & 31) * 16 + (lane >> 5) * 24576;
    f32x16 acc[R];
    for (int r = 0; r < R; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    f16x8 a[2], b[2][R];
    for (int it = 0; it < iters; ++it) {
        a[0] = *(const f16x8*)ap;
#pragma unroll
        for (int r = 0; r < R; ++r) b[0][r] = *(const f16x8*)bp[r];
        __builtin_amdgcn_sched_group_barrier(0x100, R + 1, 0);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int cb = t & 1, nb = cb ^ 1;
            if (t + 1 < 27) {
                a[nb] = *(const f16x8*)(ap + (t + 1) * 1024);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r < NB) b[nb][r] = *(const f16x8*)(bp[r] + (t + 1) * 16);
                    else b[nb][r] = b[cb][r];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb], b[cb][r], acc[r], 0, 0, 0);
            if (t + 1 < 27) __builtin_amdgcn_sched_group_barrier(0x100, NB + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, R, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < R; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, int NB> void run(int threads) {
    float* d; hipMalloc(&d, (size_t)256 * threads * 4);
    hipFuncSetAttribute((const void*)k<R, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    hipLaunchKernelGGL((k<R, NB>), dim3(256), dim3(threads), 128 * 1024, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, NB>), dim3(256), dim3(threads), 128 * 1024, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 256.0 * (threads / 64) * iters * 27.0 * R * 32768.0;
    printf("R=%d B-reads/tap=%d waves/CU=%d: %.1f TFLOP/s (%.2f ms) err=%d\n", R, NB, threads / 64, flop / ms * 1e-9, ms, (int)hipGetLastError());
    hipFree(d);
}
int main() {
    run<4, 4>(256); run<4, 2>(256); run<4, 0>(256); run<2, 2>(256); run<2, 2>(512); run<4, 4>(512); run<1, 1>(256); run<1, 1>(512);
    return 0;
}
