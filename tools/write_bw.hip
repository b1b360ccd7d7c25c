// Micro-benchmark: what a pure WRITE stream, a pure READ stream and a copy reach on this MI355X (GB/s of bytes moved), to price
// the write-bound kernels (first conv: 64 B written per 4 B read; transposed conv: 8 x its input) against something measured
// rather than the 8 TB/s data-sheet figure.   hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw && ./write_bw [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 write, 1 write non-temporal, 2 read, 3 copy
__global__ __launch_bounds__(256) void k(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        if (MODE == 0) dst[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
        if (MODE == 1) __builtin_nontemporal_store(u32x4{(unsigned)i, 1u, 2u, 3u}, dst + i);
        if (MODE == 2) { const u32x4 v = src[i]; acc += v.x ^ v.w; }
        if (MODE == 3) dst[i] = src[i];
    }
    if (MODE == 2 && acc == 0x12345678u) *sink = acc;
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t n16 = mb * 1024 * 1024 / 16;
    u32x4 *a, *b; unsigned* sink;
    hipMalloc(&a, n16 * 16); hipMalloc(&b, n16 * 16); hipMalloc(&sink, 4);
    hipMemset(a, 1, n16 * 16); hipMemset(b, 2, n16 * 16);
    const char* names[4] = {"write (16 B per lane)", "write, non-temporal", "read", "copy (read + write)"};
    for (int blocks : {1024, 4096, 16384})
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto run = [&]() {
                if (mode == 0) k<0><<<blocks, 256>>>(a, b, n16, sink);
                if (mode == 1) k<1><<<blocks, 256>>>(a, b, n16, sink);
                if (mode == 2) k<2><<<blocks, 256>>>(a, b, n16, sink);
                if (mode == 3) k<3><<<blocks, 256>>>(a, b, n16, sink);
            };
            run();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)n16 * 16 * 5 * (mode == 3 ? 2.0 : 1.0);
            printf("%5zu MiB, %5d blocks, %-24s %8.1f GB/s\n", mb, blocks, names[mode], bytes / ms / 1e6);
        }
    return 0;
}
