// HBM-bound sliding-window arithmetic: CT normalisation, Gaussian-weighted fp16 tile accumulation,
// normalise + fold-mean + first-max argmax + part remap.  Built with -ffp-contract=off: every fp32 operation
// below is a separately rounded IEEE operation, exactly as the reference's torch-CPU / numpy path executes it.
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// CTNormalization.run  (NN/preprocessing/normalization/default_normalization_schemes.py:53-67)
template <typename T>
__global__ void k_ct_normalize(const T* __restrict__ in, float* __restrict__ out, size_t n, float mean, float sd,
                               float lo, float hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = (float)in[i];
        v = fminf(fmaxf(v, lo), hi);  // np.clip
        v = v - mean;
        v = __fdiv_rn(v, sd);
        out[i] = v;
    }
}

extern "C" int boa_ct_normalize(boa_ctx* c, const void* dev_in, int in_dtype, float* dev_out, size_t n, float mean,
                                float sd, float lo, float hi) {
    BOA_REQUIRE(c && dev_in && dev_out, "boa_ct_normalize: NULL argument");
    BOA_REQUIRE(in_dtype >= 0 && in_dtype <= 2, "boa_ct_normalize: in_dtype must be 0 (int16), 1 (float32) or 2 (int32)");
    if (n == 0) return BOA_OK;
    sd = sd > 1e-8f ? sd : 1e-8f;
    int block = 256;
    int grid = (int)((n + block - 1) / block);
    if (grid > c->cu_count * 16) grid = c->cu_count * 16;
    KernelTimer t(c, BOA_K_OTHER, 0, (double)n * (in_dtype == 0 ? 6 : 8));
    if (in_dtype == 2)
        hipLaunchKernelGGL(k_ct_normalize<int32_t>, dim3(grid), dim3(block), 0, c->stream, (const int32_t*)dev_in,
                           dev_out, n, mean, sd, lo, hi);
    else if (in_dtype == 0)
        hipLaunchKernelGGL(k_ct_normalize<int16_t>, dim3(grid), dim3(block), 0, c->stream, (const int16_t*)dev_in,
                           dev_out, n, mean, sd, lo, hi);
    else
        hipLaunchKernelGGL(k_ct_normalize<float>, dim3(grid), dim3(block), 0, c->stream, (const float*)dev_in,
                           dev_out, n, mean, sd, lo, hi);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// predicted_logits[sl] += prediction * gaussian ; n_predictions[sl[1:]] += gaussian
// (NN/inference/predict_from_raw_data.py:611-614).  One thread per patch voxel (z fastest), loop over classes.
__global__ void k_accumulate_tile(const float* __restrict__ pred, const unsigned short* __restrict__ gauss,
                                  unsigned short* __restrict__ acc, unsigned short* __restrict__ nacc, int C, int P0,
                                  int P1, int P2, int V0, int V1, int V2, int s0, int s1, int s2) {
    size_t pv = (size_t)P0 * P1 * P2;
    size_t vv = (size_t)V0 * V1 * V2;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pv) return;
    int p2 = (int)(i % P2);
    int p1 = (int)((i / P2) % P1);
    int p0 = (int)(i / ((size_t)P2 * P1));
    size_t vi = ((size_t)(s0 + p0) * V1 + (s1 + p1)) * V2 + (s2 + p2);
    float g = gauss ? us2f(gauss[i]) : 1.0f;
    for (int c = 0; c < C; ++c) {
        float p = pred[(size_t)c * pv + i];
        if (gauss) p = p * g;  // prediction *= gaussian (fp32)
        float a = us2f(acc[(size_t)c * vv + vi]);
        acc[(size_t)c * vv + vi] = f2us(a + p);  // fp16 += fp32 -> fp32 add, RTNE to fp16
    }
    nacc[vi] = f2us(us2f(nacc[vi]) + g);
}

extern "C" int boa_accumulate_tile(boa_ctx* c, const float* dev_pred, const uint16_t* dev_gauss, uint16_t* dev_acc,
                                   uint16_t* dev_n, int C, const int P[3], const int V[3], const int start[3]) {
    BOA_REQUIRE(c && dev_pred && dev_acc && dev_n && P && V && start, "boa_accumulate_tile: NULL argument");
    for (int d = 0; d < 3; ++d)
        BOA_REQUIRE(start[d] >= 0 && start[d] + P[d] <= V[d], "boa_accumulate_tile: tile [%d,%d) outside volume dim %d (%d)",
                    start[d], start[d] + P[d], d, V[d]);
    size_t pv = (size_t)P[0] * P[1] * P[2];
    int block = 256;
    int grid = (int)((pv + block - 1) / block);
    KernelTimer t(c, BOA_K_HEAD_ACCUM, 0, (double)pv * (4.0 * C + 2 + 4.0 * (C + 1)));
    hipLaunchKernelGGL(k_accumulate_tile, dim3(grid), dim3(block), 0, c->stream, dev_pred, dev_gauss, dev_acc, dev_n,
                       C, P[0], P[1], P[2], V[0], V[1], V[2], start[0], start[1], start[2]);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// normalise (+ fold sum / mean) + argmax + remap, one pass.  Each thread owns VEC consecutive voxels along the
// contiguous axis so the C plane reads are 2*VEC-byte vector loads.
struct FinalizeArgs {
    unsigned short* acc;
    const unsigned short* n;
    unsigned short* fold;
    unsigned char* labels;
    int* inf_flag;
    int C, V0, V1, V2;
    int fold_mode, n_folds_final, write_logits, do_argmax, merge;
    int crop, o0, o1, o2, c0, c1, c2;
    size_t v_begin, v_count;  // flat voxel range processed (whole planes of axis 0); the channel stride stays V0*V1*V2
    unsigned char lut[256];
};

template <int VEC>
__global__ void k_finalize_labels(FinalizeArgs a) {
    const size_t vv = (size_t)a.V0 * a.V1 * a.V2;
    const size_t nvec = a.v_count / VEC;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvec) return;
    const size_t base = a.v_begin + t * VEC;
    float nf[VEC];
    float best[VEC];
    int bidx[VEC];
    bool bnan[VEC];
    unsigned short nb[VEC];
    if (VEC == 8) {
        *(uint4*)nb = *(const uint4*)(a.n + base);
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) nb[j] = a.n[base + j];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        nf[j] = us2f(nb[j]);
        best[j] = 0.f;
        bidx[j] = 0;
        bnan[j] = false;
    }
    bool any_inf = false;
    for (int c = 0; c < a.C; ++c) {
        unsigned short v[VEC];
        unsigned short* ap = a.acc + (size_t)c * vv + base;
        if (VEC == 8) {
            *(uint4*)v = *(const uint4*)ap;
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = ap[j];
        }
        unsigned short fs[VEC];
        if (a.fold && a.fold_mode == 1) {
            unsigned short* fp = a.fold + (size_t)c * vv + base;
            if (VEC == 8) {
                *(uint4*)fs = *(const uint4*)fp;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) fs[j] = fp[j];
            }
        }
        unsigned short q[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float r = __fdiv_rn(us2f(v[j]), nf[j]);  // torch.div(half, half): fp32 divide, RTNE to half
            unsigned short h = f2us(r);
            if ((h & 0x7FFF) == 0x7C00) any_inf = true;
            q[j] = h;
        }
        if (a.write_logits) {
            if (VEC == 8) {
                *(uint4*)ap = *(const uint4*)q;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) ap[j] = q[j];
            }
        }
        if (a.fold) {
            if (a.fold_mode == 1) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) q[j] = f2us(us2f(fs[j]) + us2f(q[j]));  // prediction += fold
            }
            unsigned short o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                o[j] = q[j];
                if (a.n_folds_final > 1) {
                    q[j] = f2us(__fdiv_rn(us2f(q[j]), (float)a.n_folds_final));  // prediction /= n_folds
                    o[j] = q[j];
                }
            }
            unsigned short* fp = a.fold + (size_t)c * vv + base;
            if (VEC == 8) {
                *(uint4*)fp = *(const uint4*)o;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) fp[j] = o[j];
            }
        }
        if (a.do_argmax) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float f = us2f(q[j]);
                bool isn = f != f;
                // numpy argmax: first maximum; the first NaN wins over everything
                if (c == 0) {
                    best[j] = f;
                    bidx[j] = 0;
                    bnan[j] = isn;
                } else if (!bnan[j] && (isn || f > best[j])) {
                    best[j] = f;
                    bidx[j] = c;
                    bnan[j] = isn;
                }
            }
        }
    }
    if (any_inf) atomicOr(a.inf_flag, 1);
    if (!a.do_argmax) return;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        size_t vi = base + j;
        int z = (int)(vi % a.V2);
        int y = (int)((vi / a.V2) % a.V1);
        int x = (int)(vi / ((size_t)a.V2 * a.V1));
        size_t oi = vi;
        if (a.crop) {
            x -= a.o0;
            y -= a.o1;
            z -= a.o2;
            if (x < 0 || y < 0 || z < 0 || x >= a.c0 || y >= a.c1 || z >= a.c2) continue;
            oi = ((size_t)x * a.c1 + y) * a.c2 + z;
        }
        int idx = bidx[j];
        if (a.merge) {
            if (idx != 0) a.labels[oi] = a.lut[idx];
        } else {
            a.labels[oi] = a.lut[idx];
        }
    }
}

extern "C" int boa_finalize_labels(boa_ctx* c, uint16_t* dev_acc, const uint16_t* dev_n, int C, const int V[3],
                                   uint16_t* dev_fold_sum, int fold_mode, int n_folds_final, int write_logits,
                                   const uint8_t* host_lut, int merge, uint8_t* dev_labels_out, const int* crop_off,
                                   const int* crop_dims, int* dev_inf_flag) {
    BOA_REQUIRE(V, "boa_finalize_labels: NULL argument");
    return boa_finalize_labels_planes(c, dev_acc, dev_n, C, V, dev_fold_sum, fold_mode, n_folds_final, write_logits, host_lut,
                                      merge, dev_labels_out, crop_off, crop_dims, dev_inf_flag, 0, V[0]);
}

extern "C" int boa_finalize_labels_planes(boa_ctx* c, uint16_t* dev_acc, const uint16_t* dev_n, int C, const int V[3],
                                          uint16_t* dev_fold_sum, int fold_mode, int n_folds_final, int write_logits,
                                          const uint8_t* host_lut, int merge, uint8_t* dev_labels_out,
                                          const int* crop_off, const int* crop_dims, int* dev_inf_flag, int plane_lo,
                                          int plane_hi) {
    BOA_REQUIRE(c && dev_acc && dev_n && V && dev_inf_flag, "boa_finalize_labels: NULL argument");
    BOA_REQUIRE(C >= 1 && C <= 255, "boa_finalize_labels: C=%d out of range", C);
    BOA_REQUIRE(plane_lo >= 0 && plane_lo <= plane_hi && plane_hi <= V[0], "boa_finalize_labels: planes [%d,%d) of %d",
                plane_lo, plane_hi, V[0]);
    if (plane_lo == plane_hi) return BOA_OK;
    FinalizeArgs a;
    a.acc = dev_acc;
    a.n = dev_n;
    a.fold = dev_fold_sum;
    a.labels = dev_labels_out;
    a.inf_flag = dev_inf_flag;
    a.C = C;
    a.V0 = V[0];
    a.V1 = V[1];
    a.V2 = V[2];
    a.fold_mode = fold_mode;
    a.n_folds_final = n_folds_final;
    a.write_logits = write_logits;
    a.do_argmax = (dev_labels_out != nullptr) && (dev_fold_sum == nullptr || n_folds_final > 0);
    a.merge = merge;
    a.crop = (crop_off && crop_dims) ? 1 : 0;
    if (a.crop) {
        a.o0 = crop_off[0]; a.o1 = crop_off[1]; a.o2 = crop_off[2];
        a.c0 = crop_dims[0]; a.c1 = crop_dims[1]; a.c2 = crop_dims[2];
    } else {
        a.o0 = a.o1 = a.o2 = 0;
        a.c0 = V[0]; a.c1 = V[1]; a.c2 = V[2];
    }
    for (int i = 0; i < 256; ++i) a.lut[i] = host_lut ? host_lut[i] : (unsigned char)i;
    size_t vtot = (size_t)V[0] * V[1] * V[2];
    a.v_begin = (size_t)plane_lo * V[1] * V[2];
    a.v_count = (size_t)(plane_hi - plane_lo) * V[1] * V[2];
    size_t vv = a.v_count;
    bool vec8 = (vtot % 8 == 0) && (a.v_begin % 8 == 0) && (vv % 8 == 0) &&
                (((uintptr_t)dev_acc | (uintptr_t)dev_n | (uintptr_t)dev_fold_sum) % 16 == 0);
    int block = 256;
    double bytes = (double)vv * (2.0 * C + 2 + 1 + (write_logits ? 2.0 * C : 0) + (dev_fold_sum ? 4.0 * C : 0));
    KernelTimer t(c, BOA_K_ARGMAX, 0, bytes);
    if (vec8) {
        size_t nvec = vv / 8;
        hipLaunchKernelGGL(k_finalize_labels<8>, dim3((unsigned)((nvec + block - 1) / block)), dim3(block), 0,
                           c->stream, a);
    } else {
        hipLaunchKernelGGL(k_finalize_labels<1>, dim3((unsigned)((vv + block - 1) / block)), dim3(block), 0,
                           c->stream, a);
    }
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
