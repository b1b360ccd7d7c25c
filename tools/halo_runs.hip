// Micro-benchmark for the stride-2 conv's input side (k_conv_ns<2,2,2>: the 32 -> 64 layer at 128^3, 2.9 TB/s of a layer whose fetched bytes
// equal its algorithmic bytes): does the achieved HBM rate of a halo gather depend on the LENGTH OF ITS CONTIGUOUS RUNS?
//   variant 0: output tile 4 x 4 x 8  -> halo 9 x 9 x 17 voxels: 81 runs of 17 x 32 B = 544 B, 4 KiB apart (what the kernel does)
//   variant 1: output tile 4 x 2 x 16 -> halo 9 x 5 x 33 voxels: 45 runs of 33 x 32 B = 1 056 B
//   variant 2: output tile 4 x 1 x 32 -> halo 9 x 3 x 65 voxels: 27 runs of 2 080 B
// Input: chunk planes [plane][128][128][128][16 halves] (32 B per voxel), `planes` of them (25 samples x 2 chunks = 3.3 GB: HBM-cold);
// 256 persistent workgroups of 256 threads walk contiguous tile runs (x fastest), lane pairs read the two 16-byte octets of a voxel,
// all loads of a tile are issued before they are consumed (the producers' pattern).   hipcc --offload-arch=gfx950 -O3 halo_runs.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define MAXJ 13
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ buf, int planes, int ty_out, int tz_out, unsigned* sink) {
    const int D = 128;
    const int hx = 9, hy = 2 * ty_out + 1, hz = 2 * tz_out + 1, HV = hx * hy * hz;
    const int nx = 64 / 4, ny = 64 / ty_out, nz = 64 / tz_out, per_plane = nx * ny * nz;
    const long total = (long)per_plane * planes;
    const long per_wg = (total + gridDim.x - 1) / gridDim.x;
    const int q = threadIdx.x;
    unsigned acc = 0;
    for (long t = (long)blockIdx.x * per_wg; t < (long)(blockIdx.x + 1) * per_wg && t < total; ++t) {
        const int pl = (int)(t / per_plane);
        int r = (int)(t % per_plane);
        const int tx = r % nx; r /= nx;
        const int tz = r % nz; const int ty = r / nz;
        const int x0 = 8 * tx - 1, y0 = 2 * ty_out * ty - 1, z0 = 2 * tz_out * tz - 1;
        const uint4* base = buf + (size_t)pl * D * D * D * 2;
        uint4 v[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int vi = (q >> 1) + 128 * j;
            if (128 * j < HV) {
                const int vv = vi < HV ? vi : 0;
                const int ix = vv / (hy * hz), rem = vv % (hy * hz), iy = rem / hz, iz = rem % hz;
                const int x = min(max(x0 + ix, 0), D - 1), y = min(max(y0 + iy, 0), D - 1), z = min(max(z0 + iz, 0), D - 1);
                v[j] = base[((size_t)(x * D + y) * D + z) * 2 + (q & 1)];
            } else {
                v[j] = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) acc += v[j].x ^ v[j].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main(int argc, char** argv) {
    const int planes = argc > 1 ? atoi(argv[1]) : 50;
    const size_t bytes = (size_t)planes * 128 * 128 * 128 * 32;
    uint4* buf; unsigned* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
    const int cfg[3][2] = {{4, 8}, {2, 16}, {1, 32}};
    for (int rep = 0; rep < 2; ++rep)
        for (int c = 0; c < 3; ++c) {
            const int ty = cfg[c][0], tz = cfg[c][1];
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            k<<<256, 256>>>(buf, planes, ty, tz, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const int HV = 9 * (2 * ty + 1) * (2 * tz + 1);
            const double tiles = (double)planes * 16 * (64 / ty) * (64 / tz);
            const double loaded = tiles * HV * 32.0, unique = (double)bytes;
            printf("tile 4x%dx%d halo 9x%dx%d (%d voxels, runs of %d B): %.3f ms, %.2f TB/s of halo bytes (%.2fx the tensor), %.2f TB/s of unique bytes\n", ty, tz,
                   2 * ty + 1, 2 * tz + 1, HV, (2 * tz + 1) * 32, ms, loaded / ms / 1e9, loaded / unique, unique / ms / 1e9);
        }
    return 0;
}
