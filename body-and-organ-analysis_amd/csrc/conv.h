// Conv-stack kernels of the PlainConvUNet engine (gfx950): launch wrappers shared by net.hip and the test seams.
#pragma once
#include "common.h"

// Activation tensors are CHUNK-PLANAR fp16: [N][C/16][D][H][W][16] (D,H,W = nnU-Net array axes X,Y,Z): the 16 channels of
// one MFMA K-chunk of a voxel are 32 contiguous bytes and consecutive voxels of a row are consecutive in memory, so a
// producer's wave load of one chunk reads 1 KiB of consecutive bytes (a channels-last record layout wastes half of every
// fetched line when one chunk is staged per pass: tools/load_pattern.hip measures 3.2 against 6.3 TB/s of useful bytes).
// The ncdhw helpers below convert from / to PyTorch order.  A "source" is an activation plus the deferred InstanceNorm+LeakyReLU of
// its producer: y = lrelu(x * scale[n][c] + shift[n][c]); ss == nullptr means identity (raw tensor).
struct ActSrc {
    const __half* data = nullptr;
    const float* ss = nullptr;  // [N][C][2] (scale, shift)
    const unsigned* ss16 = nullptr;  // [N][C/2][2] words: {fp16 scales of (c, c+1), fp16 shifts of (c, c+1)}
    int C = 0;
};

struct ConvGeom {
    int N, Di, Hi, Wi, Do, Ho, Wo;
    int Cout;
    int k[3], s[3];
    int Cin;  // total input channels (both sources)
};

// tile configuration chosen on the host (see choose_conv_tile)
struct ConvTile {
    int variant;     // 0: k_conv_mfma (256 threads, 2 blocks/CU); 1: k_conv_ws (producer/consumer waves, persistent); 2: k_conv_ns
    int R;           // M-tiles (32 output voxels each) per (consumer) wave; block tile = 4R M-tiles
    int w[3];        // wave M-tile shape, product 32
    int b[3];        // M-tiles per block along each axis, product 4R
    int h[3];        // input halo extents
    int xs;          // LDS stride between the halo's x-planes in voxels (>= h[1] * h[2]; see choose_conv_tile)
    int tiles[3];    // block tiles per axis
    size_t lds_bytes;
};

bool choose_conv_tile(const ConvGeom& g, int cu_count, ConvTile* out, bool x3 = false);

// packed weight sizes / packers (host side)
size_t conv_wpk_halves(int Cin_total, int Cout, const int k[3]);
void pack_conv_weights(const float* w /*[Cout][Cin][k0][k1][k2]*/, int Cin, int Cout, const int k[3], __half* dst);
size_t convt_wpk_halves(int Cin, int Cout, const int s[3]);
// split-precision mode (precision 2): hi / lo fp16 parts of w * scale, 8 real channels per MFMA K step
float x3_weight_scale(const float* w, size_t n);
size_t conv_wpk_halves_x3(int Cin_total, int Cout, const int k[3]);
void pack_conv_weights_x3(const float* w, int Cin, int Cout, const int k[3], float scale, __half* dst);
size_t convt_wpk_halves_x3(int Cin, int Cout, const int s[3]);
void pack_convt_weights_x3(const float* w, int Cin, int Cout, const int s[3], float scale, __half* dst);
void pack_convt_weights(const float* w /*[Cin][Cout][s0][s1][s2]*/, int Cin, int Cout, const int s[3], __half* dst);

// Conv3d(k, stride, pad (k-1)/2) + bias over cat(src0, src1) -> out fp16 (pre-norm) and per-block partial
// sums of (x, x^2) per (n, cout) into partials[N][Cout][2][nblk]; returns nblk through *nblk_out.
int launch_conv_mfma(boa_ctx* ctx, const ActSrc& s0, const ActSrc& s1, const ConvGeom& g, const ConvTile& t,
                     const __half* wpk, const float* bias, float slope, __half* out, float* partials);
int conv_nblk(const ConvTile& t, int cu_count, int Cout);
int conv_ws_nslots(int tiles_per_sample, int cu_count);

// First conv: reads tiles straight out of the resident fp32 volume [Cin][V0][V1][V2] (zero outside the volume
// and outside the tile), fp32 VALU, stride 1.  w: dev fp32 [Cin][taps][Cout].
int launch_conv_first(boa_ctx* ctx, const float* volume, const int V[3], const int vol_off[3], const int* dev_origins,
                      int N, int Cin, const int P[3], const int k[3], int Cout, const float* w, const float* bias,
                      float* padded_scratch, __half* out, float* partials, int* nblk_out, int flip_mask = 0, float* out32 = nullptr);
// split-precision conv (k_conv_ws<..., X3>): fp32 octet-planar sources / output, fp32 (scale, shift) tables
int launch_conv_x3(boa_ctx* ctx, const float* src0, const float* ss0, int C0, const float* src1, const float* ss1, int C1,
                   const ConvGeom& g, const ConvTile& t, const __half* wpk, float wscale, const float* bias, float slope, float* out,
                   float* partials);
int conv_first_nblk(const int P[3], int cu_count);
void conv_first_padded_dims(const int P[3], const int k[3], int out[3]);

// InstanceNorm statistics -> (scale, shift) per (n, c):  scale = gamma * rsqrt(var + eps), shift = beta - mean * scale
// clear != 0: zero the partials after reading them (k_conv_ws expects a zeroed table: it only writes the slots of waves
// that worked on an (n, cout) -- the table is zeroed once at allocation and kept zeroed by the finalize).
int launch_norm_finalize(boa_ctx* ctx, float* partials, int nblk, int N, int C, double count,
                         const float* gamma, const float* beta, float eps, float* ss_out, unsigned* ss16_out, int clear);

// ConvTranspose3d with kernel == stride, + bias; input source with deferred norm; out fp16 raw.
int launch_convt_mfma(boa_ctx* ctx, const ActSrc& src, int N, const int din[3], const int s[3], int Cout,
                      const __half* wpk, const float* bias, float slope, __half* out);

// 1x1x1 head on the last decoder activation.  mode 0: write fp32 logits [C][P0][P1][P2];
// mode 1: pred * gauss accumulated into fp16 acc/n at `start` (NN/inference/predict_from_raw_data.py:611-614).
int launch_head(boa_ctx* ctx, const __half* act, const float* ss, int F0, const int P[3], int C, const float* w,
                const float* bias, float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc,
                uint16_t* nacc, const int PV[3], const int start[3], size_t plane_stride = 0);

// ---- shared device-side definitions -------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const __half* src0;
    const __half* src1;
    const float* ss0;
    const float* ss1;
    const unsigned* ss16_0;
    const unsigned* ss16_1;
    int C0, C1;
    int N, Di, Hi, Wi, Do, Ho, Wo, Cout;
    int k0, k1, k2, s0, s1, s2, p0, p1, p2;
    int w0, w1, w2, b0, b1, b2, h0, h1, h2, t0, t1, t2;
    int xs;  // k_conv_ws / k_conv_ns: voxels between the halo's x-planes in LDS (h1 * h2, or padded: ConvTile::xs)
    int lw1, lw2, lb1, lb2;  // log2 of the (power-of-two) wave-tile / block-tile extents
    const __half* wpk;
    const float* bias;
    __half* out;
    float* partials;
    float slope;
    int ncy;                    // k_conv_ws / k_conv_ns: cout groups per spatial tile (Cout / 32 resp. ceil(Cout / 128))
    int cy_fast;                // k_conv_ws: cout chunk is the fastest tile index (halo reuse, 2-chunk inputs)
    int nslots;                 // k_conv_ws: statistics slots per (n, cout) in `partials`
    int vw;                     // k_conv_ws: virtual workgroups per sample (<= nslots / 4)
    int vstep_n, vstep_j;       // k_conv_ws: grid size as (samples, virtual workgroups) = (G / vw, G % vw)
    const int* runs;            // k_conv_ws: [vw][8] = {count, cy, sp, ox0, oy0, oz0, -, -}: the run of virtual workgroup j
    unsigned long long* trace;  // debug (BOA_WS_TRACE): per-chunk s_memtime stamps of block 0, else nullptr
    // split-precision mode (k_conv_ws<..., X3 = true>): the packed weights carry a power-of-two scale (hi / lo fp16 parts of
    // w * wscale); the accumulators start at bias * wscale and the epilogue multiplies by winv = 1 / wscale
    float wscale, winv;
};

__device__ __forceinline__ uint4 norm_act8(uint4 raw, const float* sc, const float* sh, float slope) {
    union {
        uint4 u;
        __half h[8];
    } x;
    x.u = raw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float f = __builtin_fmaf(__half2float(x.h[j]), sc[j], sh[j]);
        f = f > 0.f ? f : f * slope;
        x.h[j] = __float2half_rn(f);
    }
    return x.u;
}

int launch_conv_ws(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, double flops, double bytes, bool x3 = false);
// fused gather head (head_gather.hip): all covering tiles of a voxel -> label, from the stash of last decoder activations
int launch_pack_head_ss(boa_ctx* ctx, const float* ss, unsigned* out, int n_tiles);
int launch_gather_head(boa_ctx* ctx, const __half* act, const unsigned* ssp, const float* w, const float* bias, const uint16_t* gauss,
                       int C, const int P[3], const int PV[3], const int ntile[3], const int* dev_tab, uint16_t* fold, int fold_mode,
                       int n_folds, const uint8_t* host_lut, int merge, uint8_t* labels, const int* crop_off, const int* crop_dims,
                       int* inf_flag, float slope, int tiles_total, bool x3 = false, const int* x_range = nullptr, uint16_t* raw_n = nullptr,
                       int raw_init = 0);
// k_conv_ns (conv_ns.hip): consumer waves split the cout axis, weights straight from L2 (stride-2 / deep 3x3x3 layers)
bool conv_ns_applicable(const ConvGeom& g);
void conv_ns_tile(const ConvGeom& g, ConvTile* t);
int conv_ns_ncy(int Cout);
int launch_conv_ns(boa_ctx* ctx, const ConvArgs& a, const ConvTile& t, double flops, double bytes, bool x3 = false);

// ---- fp32 "exact" mode (net_f32.hip): channels-last fp32 activations, weights [tap][Cin][Cout] fp32 -------------------
int launch_conv_f32(boa_ctx* ctx, const float* src0, const float* ss0, int C0, const float* src1, const float* ss1, int C1, int N,
                    const int din[3], const int dout[3], const int k[3], const int s[3], int Cout, const float* w,
                    const float* bias, float slope, float* out);
int launch_convt_f32(boa_ctx* ctx, const float* src, const float* ss, int Cin, int N, const int din[3], const int s[3], int Cout,
                     const float* w, const float* bias, float slope, float* out);
int launch_stats_f32(boa_ctx* ctx, const float* act, int N, size_t vox, int C, const float* gamma, const float* beta, float eps,
                     float* ss_out);
int launch_gather_tiles_f32(boa_ctx* ctx, const float* volume, const int V[3], const int vol_off[3], const int* dev_origins, int N,
                            int Cin, const int P[3], float* out, int flip_mask = 0);
// test-time mirroring: dst[c][p] (=|+=) src[c][flip(p)] over fp32 logits [C][P0][P1][P2]; scale applied after the add
int launch_flip_accumulate(boa_ctx* ctx, const float* src, float* dst, int C, const int P[3], int flip_mask, int add, float scale);
int launch_head_f32(boa_ctx* ctx, const float* act, const float* ss, int F0, const int P[3], int C, const float* w,
                    const float* bias, float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc, uint16_t* nacc,
                    const int PV[3], const int start[3], size_t octet_stride = 0);
// ---- split-precision mode (precision 2; net_x3.hip, k_conv_ws<..., X3>): fp32 octet planes [N][C/8][voxel][8] ---------
int launch_convt_x3(boa_ctx* ctx, const float* src, const float* ss, int Cin, int N, const int din[3], const int s[3], int Cout,
                    const __half* wpk, float wscale, const float* bias, float slope, float* out);
// head weights are O(0.1 .. 1): this power of two keeps the lo parts of their split out of the fp16 subnormal range (shared by the
// scatter and the gather head: both must split the weights identically)
#define X3_HEAD_WSCALE 1024.f
int launch_head_x3(boa_ctx* ctx, const float* act, const float* ss, int F0, const int P[3], int C, const float* w, const float* bias,
                   float slope, float* logits_out, const uint16_t* gauss, uint16_t* acc, uint16_t* nacc, const int PV[3], const int start[3],
                   size_t plane_stride);
int launch_octet_to_nchw_f32(boa_ctx* ctx, const float* in, const float* ss, float slope, int C, size_t vox, float* out);

int launch_ndhwc32_to_nchw_f32(boa_ctx* ctx, const float* in, const float* ss, float slope, int C, size_t vox, float* out);

// layout helpers (tests / debug)
int launch_nchw_to_ndhwc_f16(boa_ctx* ctx, const float* in, int N, int C, size_t vox, __half* out);
int launch_ndhwc_to_nchw_f32(boa_ctx* ctx, const __half* in, const float* ss, float slope, int N, int C, size_t vox,
                             float* out);
