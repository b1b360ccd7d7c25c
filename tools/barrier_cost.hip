#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(int* out, int iters, int work) {
    int acc = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        for (int w = 0; w < work; ++w) acc = acc * 3 + 1;
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    int* d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) for (int work : {0, 16}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 10, work); hipDeviceSynchronize();
        for (int iters : {0, 256, 2048}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters, work);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("threads=%d work=%d iters=%d: %.1f us\n", threads, work, iters, ms * 1000);
        }
    }
    return 0;
}
