// What does an instruction of a second wave cost while the first wave of the same SIMD keeps the matrix pipe busy?
// 8 waves per CU (2 per SIMD): waves 0-3 issue back-to-back independent v_mfma_f32_32x32x16_f16 (mode bit 0) and
// optionally the consumer's LDS fragment reads (bit 1); waves 4-7 time blocks of 32 independent instructions of one
// kind with s_memtime.  Printed: cycles per instruction of the timed wave, and the MFMA rate the other waves reached.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int KIND>
__device__ __forceinline__ void block32(unsigned& a, unsigned& b, unsigned& c, unsigned& d, unsigned char* lds, const uint4* g, const uint4* gb, unsigned loff, uint4& acc) {
    if (KIND == 0) {  // packed fp16 fma, 4 independent chains
        asm volatile(REP8("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0x3c003c00u), "v"(0x00010001u));
    } else if (KIND == 1) {  // fp32 fma
        asm volatile(REP8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f), "v"(0.0f));
    } else if (KIND == 2) {  // integer add
        asm volatile(REP8("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1u));
    } else if (KIND == 3) {  // ds_write_b128
        u32x4 v = {a, b, c, d};
        unsigned addr = (unsigned)(size_t)lds;
        asm volatile(REP32("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(addr), "v"(v) : "memory");
    } else if (KIND == 4) {  // global_load_dwordx4 (L2-resident, same lines every time)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            u32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g + i * 64) : "memory");
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            acc.x ^= v.x;
        }
    } else if (KIND == 6) {  // global_load_dwordx4, scalar base + 32-bit lane offset
        const unsigned off = loff;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            u32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(off), "s"(gb), "n"(i * 64) : "memory");
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            acc.x ^= v.x;
        }
    } else if (KIND == 7) {  // global_load_lds_dwordx4: straight to LDS (M0 = LDS base), no data VGPRs
        const unsigned off = loff;
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)lds & 0xffff)));
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(off), "s"(gb), "n"(i * 64) : "memory");
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    } else if (KIND == 5) {  // scalar ALU
        unsigned s = 1;
        asm volatile(REP32("s_add_u32 %0, %0, 1\n") : "+s"(s));
        a += s;
    }
}

template <int KIND, bool AG>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, const uint4* g, int mode, int mfma_iters, int valu_iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u;
    __syncthreads();
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
        f16x8 a, b[4];
        a = *(const f16x8*)(smem + lane * 16);
        for (int r = 0; r < 4; ++r) b[r] = *(const f16x8*)(smem + 4096 + r * 1024 + lane * 16);
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (mode & 2) {
                    a = *(const f16x8*)(smem + ((t * 1024 + lane * 16) & 32767));
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[r] = *(const f16x8*)(smem + 32768 + ((t * 4 + r) * 1024 + lane * 16) % 32768);
                }
                if (AG) {  // accumulators in AccVGPRs
#pragma unroll
                    for (int r = 0; r < 4; ++r) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[r]) : "v"(a), "v"(b[r]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[r], acc[r], 0, 0, 0);
                }
            }
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
        return;
    }
    unsigned a = lane, b = lane + 1, c = lane + 2, d = lane + 3;
    uint4 acc = make_uint4(0, 0, 0, 0);
    // let the MFMA waves get going
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < valu_iters; ++it) block32<KIND>(a, b, c, d, smem + 49152 + (threadIdx.x - 256) * 16, g + lane, g, lane * 16, acc);
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a + b + c + d + acc.x);
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    float* d; unsigned long long* c; uint4* g;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 64); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    hipFuncSetAttribute((const void*)k<KIND, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute((const void*)k<KIND, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const int valu_iters = 200;
    for (int mode : {0, 1, 3, 5, 7}) {
        hipMemset(c, 0, 64);
        const int mfma_iters = 4000;
        if (mode & 4)
            hipLaunchKernelGGL((k<KIND, true>), dim3(256), dim3(512), 64 * 1024, 0, d, c, g, mode, mfma_iters, valu_iters);
        else
            hipLaunchKernelGGL((k<KIND, false>), dim3(256), dim3(512), 64 * 1024, 0, d, c, g, mode, mfma_iters, valu_iters);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        double per = (double)h[4] / (valu_iters * 32.0);
        double mf = h[0] ? (double)h[0] / (mfma_iters * 32.0) : 0.0;
        printf("%-16s %-28s %6.1f cycles / instruction   (MFMA wave: %.1f cycles per MFMA)\n", name,
               mode == 0 ? "alone" : mode == 1 ? "next to MFMA" : mode == 3 ? "next to MFMA + ds_read" : mode == 5 ? "next to MFMA (AGPR acc)" : "next to MFMA(AGPR) + ds_read", per, mf);
    }
    hipFree(d); hipFree(c); hipFree(g);
}
int main() {
    run<0>("v_pk_fma_f16"); run<1>("v_fma_f32"); run<2>("v_add_u32"); run<3>("ds_write_b128"); run<4>("global_load_x4"); run<5>("s_add_u32"); run<6>("global_load saddr"); run<7>("global_load_lds x4");
    return 0;
}
