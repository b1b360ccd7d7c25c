// LDS store throughput of 4 "producer" waves while 4 "consumer" waves run the k_conv_ws fragment-read + MFMA loop.
// Reports cycles per producer ds_write_b128 wave-instruction for several store counts, with / without consumer load.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NST, bool LOAD, int WIDTH>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u;
    __syncthreads();
    if (wave >= 4) {
        uint4 v = make_uint4(lane, wave, 1, 2);
        unsigned char* d = smem + 100 * 1024 + (threadIdx.x - 256) * 16;
        long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                if (WIDTH == 16) *(uint4*)(d + (j % 6) * 4096) = v;
                else { *(uint2*)(d + (j % 6) * 4096) = make_uint2(v.x, v.y); *(uint2*)(d + (j % 6) * 4096 + 8) = make_uint2(v.z, v.w); }
                v.x += 1;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        }
        long long t1 = clock64();
        if (threadIdx.x == 256 && blockIdx.x == 0) *cyc = t1 - t0;
        out[blockIdx.x * 512 + threadIdx.x] = (float)v.x;
        return;
    }
    if (!LOAD) return;
    const unsigned char* ap = smem + lane * 16;
    const unsigned char* bp[4];
    for (int r = 0; r < 4; ++r) bp[r] = smem + 32768 + ((wave * 4 + r) & 15) * 2048 + (lane & 31) * 16 + (lane >> 5) * 24576;
    f32x16 acc[4];
    for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    f16x8 a[2], b[2][4];
    for (int it = 0; it < iters * NST / 12; ++it) {
        a[0] = *(const f16x8*)ap;
#pragma unroll
        for (int r = 0; r < 4; ++r) b[0][r] = *(const f16x8*)bp[r];
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int cb = t & 1, nb = cb ^ 1;
            if (t + 1 < 27) {
                a[nb] = *(const f16x8*)(ap + (t + 1) * 1024);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[nb][r] = *(const f16x8*)(bp[r] + (t + 1) * 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb], b[cb][r], acc[r], 0, 0, 0);
            if (t + 1 < 27) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NST, bool LOAD, int WIDTH> void run() {
    float* d; long long* c; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 8);
    hipFuncSetAttribute((const void*)k<NST, LOAD, WIDTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 400;
    hipLaunchKernelGGL((k<NST, LOAD, WIDTH>), dim3(256), dim3(512), 150 * 1024, 0, d, c, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("stores/iter=%d width=%d consumer_load=%d: %.1f cycles per producer wave-store (err=%d)\n", NST, WIDTH, (int)LOAD, (double)h / (iters * (double)NST * (WIDTH == 16 ? 1 : 2)), (int)hipGetLastError());
}
int main() {
    run<12, false, 16>(); run<12, true, 16>(); run<24, true, 16>(); run<12, true, 8>(); run<12, false, 8>();
    return 0;
}
