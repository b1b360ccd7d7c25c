// Single-wave instruction issue rate on gfx950: straight-line sequences of INDEPENDENT instructions, one or two waves per SIMD,
// cycles per instruction by s_memtime.  (Round 5: the role loops of k_conv_ws run one wave per SIMD and phase; their scalar / VALU
// bookkeeping was measured at ~10 cycles per instruction -- is that the issue cadence, the encoding size (instruction fetch) or
// dependency stalls?)   hipcc --offload-arch=gfx950 -O3 -o issue_rate tools/issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP256(x) REP16(REP16(x))

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 4-byte VOP2, 8 independent chains
            REP256(asm volatile("v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %1, 1.0, %1\n v_add_f32_e32 %2, 1.0, %2\n v_add_f32_e32 %3, 1.0, %3\n"
                                "v_add_f32_e32 %4, 1.0, %4\n v_add_f32_e32 %5, 1.0, %5\n v_add_f32_e32 %6, 1.0, %6\n v_add_f32_e32 %7, 1.0, %7"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 1) {   // 8-byte VOP3P packed fp32, 4 independent chains (x2 to match instruction count)
            REP256(asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n"
                                "v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3"
                                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
        } else if (MODE == 2) {   // 8-byte VOP3 encodings of plain adds
            REP256(asm volatile("v_add_f32_e64 %0, 1.0, %0\n v_add_f32_e64 %1, 1.0, %1\n v_add_f32_e64 %2, 1.0, %2\n v_add_f32_e64 %3, 1.0, %3\n"
                                "v_add_f32_e64 %4, 1.0, %4\n v_add_f32_e64 %5, 1.0, %5\n v_add_f32_e64 %6, 1.0, %6\n v_add_f32_e64 %7, 1.0, %7"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {   // SALU, 4 independent chains
            REP256(asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n"
                                "s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1"
                                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if (MODE == 4) {   // v_readlane -> SALU use pairs (the SGPR-spill reload pattern)
            REP256(asm volatile("v_readlane_b32 %0, %4, 1\n s_add_i32 %1, %0, 1\n v_readlane_b32 %2, %4, 2\n s_add_i32 %3, %2, 1\n"
                                "v_readlane_b32 %0, %4, 3\n s_add_i32 %1, %0, 1\n v_readlane_b32 %2, %4, 4\n s_add_i32 %3, %2, 1"
                                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0) : "scc");)
        } else if (MODE == 7) {   // always-taken short forward branches (skipping one instruction), 8 per group
            REP256(asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_cbranch_scc1 2f\n s_nop 0\n2:\n s_cbranch_scc1 3f\n s_nop 0\n3:\n s_cbranch_scc1 4f\n s_nop 0\n4:\n"
                                "s_cbranch_scc1 5f\n s_nop 0\n5:\n s_cbranch_scc1 6f\n s_nop 0\n6:\n s_cbranch_scc1 7f\n s_nop 0\n7:\n s_cbranch_scc1 8f\n s_nop 0\n8:" : : : "scc");)
        } else if (MODE == 8) {   // never-taken branches
            REP256(asm volatile("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n"
                                "s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n1:" : : : "scc");)
        } else if (MODE == 5) {   // dependent 4-byte VOP2 chain
            REP256(asm volatile("v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %0, 1.0, %0\n"
                                "v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %0, 1.0, %0\n v_add_f32_e32 %0, 1.0, %0"
                                : "+v"(a0));)
        } else if (MODE == 6) {   // packed fp16 fma / mul / max as in the producers' commit (8-byte VOP3P), 4 chains
            REP256(asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n"
                                "v_pk_mul_f16 %0, %0, %0\n v_pk_mul_f16 %1, %1, %1\n v_pk_max_f16 %2, %2, %2\n v_pk_max_f16 %3, %3, %3"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + s0 + s1 + s2 + s3;
}

template <int MODE>
static void run(const char* name, int threads) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 8);
    hipMalloc(&sink, 256 * 512 * 4);
    const int iters = 4;
    printf("[%s]\n", name);
    fflush(stdout);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-58s %d waves/SIMD: %6.2f cycles per instruction\n", name, threads / 256, (double)h / (iters * 256.0 * 8.0));
    fflush(stdout);
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("v_add_f32_e32 (4 B), 8 independent chains", threads);
        run<2>("v_add_f32_e64 (8 B), 8 independent chains", threads);
        run<1>("v_pk_add_f32 (8 B), 4 independent chains", threads);
        run<6>("v_pk_fma/mul/max_f16 (8 B), 4 chains", threads);
        run<5>("v_add_f32_e32 (4 B), one dependent chain", threads);
        run<3>("s_add_i32 (4 B), 4 independent chains", threads);
        run<4>("v_readlane_b32 -> s_add_i32 pairs", threads);
        run<7>("taken forward branches (8 counted per group)", threads);
        run<8>("not-taken branches (8 counted per group)", threads);
    }
    return 0;
}
