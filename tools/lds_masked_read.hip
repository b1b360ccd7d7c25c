// How much LDS time does a ds_read_b128 cost when only a few lanes are active (exec-masked), and what does a wave-wide DPP shift
// (wave_shl:1) cost?  Background: deriving the dz = 1, 2 fragments of a conv row from the dz = 0 fragment by a one-lane shift needs
// the boundary lanes (31, 63) re-read from LDS.  4 waves per CU issue reads back to back; cycles per read instruction per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: all 64 lanes; 1: lanes 31 and 63; 2: 8 lanes (every 8th); 3: no read, 16 DPP wave_shl:1 per iteration; 4: 16 DPP row_shl:1;
                      // 5: 16 plain v_add (the VALU baseline)
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool act = MODE == 0 ? true : (MODE == 1 ? (lane & 31) == 31 : (lane & 7) == 7);
    unsigned addr = (unsigned)(size_t)lds + ((threadIdx.x >> 6) * 8192) + (lane & 31) * 16 + (lane >> 5) * 4096;
    u4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE >= 3) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (MODE == 3) {
                    v0[w] = __builtin_amdgcn_update_dpp(v0[w], v0[w], 0x130, 0xF, 0xF, false);
                    v1[w] = __builtin_amdgcn_update_dpp(v1[w], v1[w], 0x130, 0xF, 0xF, false);
                    v2[w] = __builtin_amdgcn_update_dpp(v2[w], v2[w], 0x130, 0xF, 0xF, false);
                    v3[w] = __builtin_amdgcn_update_dpp(v3[w], v3[w], 0x130, 0xF, 0xF, false);
                } else if (MODE == 4) {
                    v0[w] = __builtin_amdgcn_update_dpp(v0[w], v0[w], 0x101, 0xF, 0xF, false);
                    v1[w] = __builtin_amdgcn_update_dpp(v1[w], v1[w], 0x101, 0xF, 0xF, false);
                    v2[w] = __builtin_amdgcn_update_dpp(v2[w], v2[w], 0x101, 0xF, 0xF, false);
                    v3[w] = __builtin_amdgcn_update_dpp(v3[w], v3[w], 0x101, 0xF, 0xF, false);
                } else {
                    v0[w] += v1[w]; v1[w] += v2[w]; v2[w] += v3[w]; v3[w] += v0[w];
                }
            }
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        } else if (act) {
            // 16 reads in flight, one wait: throughput, not latency
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:512\n ds_read_b128 %2, %4 offset:1024\n ds_read_b128 %3, %4 offset:1536\n"
                         "ds_read_b128 %0, %4 offset:2048\n ds_read_b128 %1, %4 offset:2560\n ds_read_b128 %2, %4 offset:3072\n ds_read_b128 %3, %4 offset:3584\n"
                         "ds_read_b128 %0, %4 offset:16\n ds_read_b128 %1, %4 offset:528\n ds_read_b128 %2, %4 offset:1040\n ds_read_b128 %3, %4 offset:1552\n"
                         "ds_read_b128 %0, %4 offset:2064\n ds_read_b128 %1, %4 offset:2576\n ds_read_b128 %2, %4 offset:3088\n ds_read_b128 %3, %4 offset:3600\n s_waitcnt lgkmcnt(0)"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(addr) : "memory");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = v0[0] + v1[1] + v2[2] + v3[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    unsigned* d; long long* c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters, c);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters, c);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters, c);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters, c);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, iters, c);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, d, iters, c);
            hipDeviceSynchronize();
        }
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.1f cycles per iteration (16 reads / 16 DPP movs / 16 adds) per wave, 4 waves per CU\n", mode, (double)h / iters);
    }
    return 0;
}
