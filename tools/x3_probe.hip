// Probe for the split-precision ("x3") network mode: (1) does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs, (2) how close is the
// K-packed hi/lo product  D += [Ah|Ah] x [Bh;Bl] + [Al|Al] x [Bh;Bl]  (8 real channels per MFMA K = 16) to an fp32 FMA chain and to the
// fp64 truth, (3) the rate of a two-MFMA-per-fragment loop.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/x3_probe tools/x3_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_denorm(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)0.f;
        b[j] = (_Float16)0.f;
    }
    // A[i][0] = 2^-20 (fp16 subnormal), B[0][n] = 2^10: D = 2^-10 if subnormal inputs are kept, 0 if flushed
    if ((lane >> 5) == 0) {
        a[0] = (_Float16)9.5367431640625e-07f;
        b[0] = (_Float16)1024.f;
    }
    f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (lane == 0) out[0] = d[0];
    // subnormal PRODUCT of normal inputs: 2^-10 * 2^-10 = 2^-20 (fine in fp32); and a tiny fp32 accumulate
    for (int j = 0; j < 8; ++j) a[j] = b[j] = (_Float16)0.f;
    if ((lane >> 5) == 0) {
        a[0] = (_Float16)0.0009765625f;
        b[0] = (_Float16)0.0009765625f;
    }
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (lane == 0) out[1] = d[0];
}

// D[32][32] = A[32][K] B[K][32], K = 8 * nchunk real channels, operands given as fp32; Sa, Sb = power-of-two pre-scales
__global__ void k_split(const float* A, const float* B, int K, float Sa, float Sb, float* D, int with_lo) {
    const int lane = threadIdx.x, l31 = lane & 31, kh = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int c0 = 0; c0 < K; c0 += 8) {
        f16x8 ah, al, b;
        for (int j = 0; j < 8; ++j) {
            const float a = A[l31 * K + c0 + j] * Sa;
            const _Float16 h = (_Float16)a;
            ah[j] = h;
            al[j] = (_Float16)(a - (float)h);
            const float x = B[(c0 + j) * 32 + l31] * Sb;
            const _Float16 xh = (_Float16)x;
            const _Float16 xl = (_Float16)(x - (float)xh);
            b[j] = kh ? xl : xh;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc, 0, 0, 0);
        if (with_lo) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc, 0, 0, 0);
    }
    const float inv = 1.f / (Sa * Sb);
    for (int gq = 0; gq < 4; ++gq)
        for (int e = 0; e < 4; ++e) D[(8 * gq + 4 * kh + e) * 32 + l31] = acc[4 * gq + e] * inv;
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, d);
    float h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("subnormal fp16 input: D = %g (kept: %g)   normal inputs, tiny product: D = %g (expect %g)\n", h[0], 9.5367431640625e-07 * 1024, h[1],
           0.0009765625 * 0.0009765625);
    const int K = 864;
    std::vector<float> A(32 * K), B(K * 32);
    unsigned s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((int)(s >> 8) % 20001 - 10000) * 1e-4f;
    };
    auto gauss = [&]() { return (rnd() + rnd() + rnd()) * 1.0f; };
    for (auto& v : A) v = 0.05f * gauss();
    for (auto& v : B) {
        float y = gauss();
        v = y > 0 ? y : 0.01f * y;  // post-LeakyReLU like
    }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dD, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> truth(1024);
    std::vector<float> chain(1024);
    double scale = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double t = 0;
            float c = 0;
            for (int k = 0; k < K; ++k) {
                t += (double)A[i * K + k] * (double)B[k * 32 + n];
                c = fmaf(A[i * K + k], B[k * 32 + n], c);
            }
            truth[i * 32 + n] = t;
            chain[i * 32 + n] = c;
            scale += t * t;
        }
    scale = sqrt(scale / 1024);
    auto report = [&](const char* what, const float* got) {
        double mx = 0, rms = 0;
        for (int i = 0; i < 1024; ++i) {
            const double e = fabs((double)got[i] - truth[i]);
            mx = fmax(mx, e);
            rms += e * e;
        }
        printf("%-44s max err %.3g  rms err %.3g  (relative to output rms %.3g: %.3g / %.3g)\n", what, mx, sqrt(rms / 1024), scale, mx / scale,
               sqrt(rms / 1024) / scale);
    };
    report("fp32 fma chain (host)", chain.data());
    std::vector<float> got(1024);
    const float scales[][2] = {{1.f, 1.f}, {32768.f, 32.f}, {32768.f, 1024.f}, {1024.f, 1.f}};
    for (auto& sc : scales)
        for (int lo = 1; lo >= 0; --lo) {
            hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, K, sc[0], sc[1], dD, lo);
            hipMemcpy(got.data(), dD, 4096, hipMemcpyDeviceToHost);
            char name[128];
            snprintf(name, sizeof name, "split MFMA Sa=%g Sb=%g%s", sc[0], sc[1], lo ? "" : " (hi weights only)");
            report(name, got.data());
        }
    return 0;
}
