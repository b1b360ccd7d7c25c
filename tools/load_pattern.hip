// Micro-benchmark behind the activation layout: per-CU load throughput of the k_conv_ws producers' access pattern.
//   pattern 0: lane pairs read the two 16-byte octets of a 32-byte half of a 64-byte voxel record (records 64 B apart):
//              a wave instruction covers 32 records = 2 KiB of address space and uses half of it (channels-last, C = 32,
//              one 16-channel chunk per pass)
//   pattern 1: the same bytes per lane, but contiguous (chunk-planar layout: a wave instruction reads 1 KiB of consecutive
//              bytes)
// Each wave issues `DEPTH` loads back to back, then consumes them (as the producers do); all CUs stream disjoint slices of a
// buffer of `mb` MiB repeatedly (mb > 300: HBM; mb <= 16: L2-resident).   hipcc --offload-arch=gfx950 -O3 load_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define DEPTH 10
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ buf, size_t n16, int pattern, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t waves = (size_t)gridDim.x * 4, gw = (size_t)blockIdx.x * 4 + wave;
    unsigned acc = 0;
    // a wave instruction covers `span` uint4 elements of address space
    const size_t span = pattern == 0 ? 128 : 64;
    const size_t per_wave = n16 / waves / (span * DEPTH) * (span * DEPTH);
    const uint4* base = buf + gw * per_wave;
    for (int it = 0; it < iters; ++it)
        for (size_t off = 0; off + span * DEPTH <= per_wave; off += span * DEPTH) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const size_t e = pattern == 0 ? (size_t)(lane >> 1) * 4 + (lane & 1) : (size_t)lane;
                v[d] = base[off + d * span + e];
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].w;
        }
    if (acc == 0x12345678u) *sink = acc;
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 1024;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    const size_t n16 = mb * 1024 * 1024 / 16;
    uint4* buf; unsigned* sink;
    hipMalloc(&buf, n16 * 16); hipMalloc(&sink, 4); hipMemset(buf, 1, n16 * 16);
    for (int pattern = 0; pattern < 2; ++pattern) {
        const int iters = mb >= 256 ? 4 : (int)(4096 / mb);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<<<blocks, 256>>>(buf, n16, pattern, 1, sink);
        hipEventRecord(a);
        k<<<blocks, 256>>>(buf, n16, pattern, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double used = (double)n16 * 16 * iters * (pattern == 0 ? 0.5 : 1.0);  // bytes the lanes received
        printf("buffer %zu MiB, %d blocks, pattern %d (%s): %.1f GB/s of loaded bytes, %.2f B/clk/CU at 2.0 GHz\n", mb, blocks, pattern,
               pattern ? "contiguous 1 KiB per wave load" : "32 B of every 64 B record", used / ms / 1e6, used / ms / 1e6 / 256 / 2.0);
    }
    return 0;
}
