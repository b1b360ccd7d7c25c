// Feasibility of the N-split consumer loop: 4 consumer waves of a workgroup share the SAME 4 M-tiles (x-stacked planes, stride S)
// and take DIFFERENT cout chunks; B (input planes) from LDS with reuse along x, A (weights) straight from global / L2 into
// registers (private per wave), ring of 3 (dy,dz) groups, prefetch distance 2 groups.  W "producer" waves idle at a barrier.
// hipcc --offload-arch=gfx950 -O3 tools/consumer_ns.hip -o /tmp/consumer_ns && /tmp/consumer_ns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GLOBAL __attribute__((address_space(1)))

template <int S, int RM>
__global__ __launch_bounds__(512) void k(const _Float16* __restrict__ wpk, float* out, int ncc, int Cout, int tiles, int h1, int h2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 150 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + i * 7;
    __syncthreads();
    if (wave >= 4) {
        for (int t = 0; t < tiles * ncc; ++t) __syncthreads();
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int l31 = lane & 31, kh = lane >> 5;
    constexpr int NB = S * (RM - 1) + 3;
    const int plane = (S == 1 ? 6 : 9) * h1 * h2 * 16 + 64;
    const int ly = l31 >> 3, lz = l31 & 7;
    const int boff = kh * plane + ((S * ly) * h2 + S * lz) * 16;
    const int cy = wave;  // this wave's cout chunk
    f32x16 acc[RM];
    for (int r = 0; r < RM; ++r)
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    f16x8 a[3][3];
    f16x8 b[NB];
    // weights: [(cc * 27 + tap) * 2 + kh][Cout][8]
    const unsigned voff = ((unsigned)kh * Cout + cy * 32 + l31) * 16u;
    auto abase = [&](int cc, int tap) -> const GLOBAL unsigned char* {
        const GLOBAL unsigned char* g = (const GLOBAL unsigned char*)wpk + ((size_t)(cc * 27 + tap) * 2) * Cout * 16;
        asm volatile("" : "+s"(g));
        return g;
    };
    auto fetch_a = [&](int cc, int g, int slot) {  // group g = (dy, dz): taps dx = 0..2
        const int dy = g / 3, dz = g % 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            unsigned vo = voff;
            asm volatile("" : "+v"(vo));
            a[slot][dx] = *(const GLOBAL f16x8*)(abase(cc, (dx * 3 + dy) * 3 + dz) + vo);
        }
    };
    int cc_next = 0;
    fetch_a(0, 0, 0);
    fetch_a(0, 1, 1);
    for (int t = 0; t < tiles; ++t) {
        for (int cc = 0; cc < ncc; ++cc) {
            const int ccn = cc + 1 < ncc ? cc + 1 : 0;
            int bo = boff + ((t * ncc + cc) & 1) * 2 * plane;  // alternating halo buffers
            asm volatile("" : "+v"(bo));
            const unsigned char* cur = smem + bo;
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) b[jj] = *(const f16x8*)(cur + ((jj * h1 + 0) * h2 + 0) * 16);
            __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                const int slot = g % 3;
                // prefetch the weights of group g + 2 (next chunk's first groups at the end)
                if (g + 2 < 9)
                    fetch_a(cc, g + 2, (g + 2) % 3);
                else
                    fetch_a(ccn, g + 2 - 9, (g + 2) % 3);
                __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) {
                    int cnt = 0;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int rr = jj - dx;
                        if (rr < 0 || rr % S != 0 || rr / S >= RM) continue;
                        acc[rr / S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[slot][dx], b[jj], acc[rr / S], 0, 0, 0);
                        ++cnt;
                    }
                    if (cnt == 1)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    else if (cnt == 2)
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    else if (cnt == 3)
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    if (g + 1 < 9) {
                        const int gn = g + 1, dy = gn / 3, dz = gn % 3;
                        b[jj] = *(const f16x8*)(cur + ((jj * h1 + dy) * h2 + dz) * 16);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
            __syncthreads();
        }
    }
    (void)cc_next;
    float s = 0;
    for (int r = 0; r < RM; ++r)
        for (int i = 0; i < 16; ++i) s += acc[r][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int S, int RM>
void run(int ncc, int Cout, int prod_waves) {
    const int h1 = S * 3 + 3, h2 = S * 7 + 3;
    size_t wbytes = (size_t)ncc * 27 * 2 * Cout * 16;
    _Float16* w;
    hipMalloc(&w, wbytes);
    {
        unsigned short* hw = (unsigned short*)malloc(wbytes);
        for (size_t i = 0; i < wbytes / 2; ++i) hw[i] = 0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
        hipMemcpy(w, hw, wbytes, hipMemcpyHostToDevice);
        free(hw);
    }
    float* d;
    hipMalloc(&d, 256 * 256 * 4);
    hipFuncSetAttribute((const void*)k<S, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int tiles = 64;
    const int threads = 256 + 64 * prod_waves;
    hipLaunchKernelGGL((k<S, RM>), dim3(256), dim3(threads), 150 * 1024, 0, w, d, ncc, Cout, 2, h1, h2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<S, RM>), dim3(256), dim3(threads), 150 * 1024, 0, w, d, ncc, Cout, tiles, h1, h2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = 256.0 * 4 * tiles * ncc * 27.0 * RM * 32768.0;
    printf("S=%d RM=%d ncc=%d Cout=%d (weights %.0f KB) prod_waves=%d: %.1f TFLOP/s (%.3f ms) err=%d\n", S, RM, ncc, Cout, wbytes / 1024.0,
           prod_waves, flop / ms * 1e-9, ms, (int)hipGetLastError());
    hipFree(d);
    hipFree(w);
}
int main() {
    run<1, 4>(4, 128, 4);
    run<1, 4>(8, 128, 4);
    run<1, 4>(16, 256, 4);
    run<1, 4>(20, 320, 4);
    run<2, 4>(4, 128, 4);
    run<2, 4>(8, 256, 4);
    run<2, 4>(16, 320, 4);
    run<1, 4>(8, 128, 0);
    run<2, 4>(8, 256, 0);
    return 0;
}
