/*
 * boa_hip.h -- C ABI of libboa_hip.so: the MI355X (gfx950) engine for the BOA hot path.
 *
 * The reference (UMEssen/Body-and-Organ-Analysis v1.0.1) is 100 % Python and has NO FFI / plugin interface
 * for this path: the boundary it exposes is the Python function surface of
 * body_organ_analysis/compute/ and of the vendored nnU-Net / TotalSegmentator / body_composition_analysis
 * packages.  This header is therefore the binding a maintainer would add underneath those functions (ctypes
 * stubs are shown in INTEGRATION.md).  Every entry point cites the reference code it replaces
 * (paths relative to the reference root; NN = body_organ_analysis/_external/nnunetv2,
 * TS = .../totalsegmentator, BCA = .../body_composition_analysis, BOA = body_organ_analysis).
 *
 * Conventions
 *   - plain C: pointers, sizes, ints; no torch / C++ types.  All functions return 0 on success, a negative
 *     BOA_E* code otherwise; boa_last_error() returns a thread-local message for the last failure.
 *   - "dev" pointers are HIP device pointers (from boa_malloc, or any hipMalloc / torch data_ptr() in the
 *     same process); "host" pointers are caller-owned and only borrowed for the duration of the call.
 *   - volumes use the nnU-Net array order [X][Y][Z] with Z contiguous (NN/imageio/nibabel_reader_writer.py:51-56)
 *     for the inference path and SimpleITK order [z][y][x] with x contiguous for the measurement path
 *     (BOA/compute/measurements.py:257-258); both are "3 dims, last contiguous" for the kernels.
 *   - all launches go to the context's stream; calls are asynchronous unless documented otherwise.
 *   - fp16 values are IEEE binary16 bit patterns carried as uint16_t.
 */
#ifndef BOA_HIP_H
#define BOA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BOA_OK 0
#define BOA_EINVAL (-1)   /* bad argument / unsupported geometry  (reference: ValueError / AssertionError) */
#define BOA_EHIP (-2)     /* HIP runtime error                     (reference: RuntimeError)              */
#define BOA_ENOMEM (-3)   /* device allocation failed              (reference: OOM -> CPU retry, predict_from_raw_data.py:663-672) */
#define BOA_EINF (-4)     /* inf in normalised logits              (reference: RuntimeError, predict_from_raw_data.py:622-625)     */

typedef struct boa_ctx boa_ctx;
typedef struct boa_net boa_net;

/* ------------------------------------------------------------------ context / memory / timing ------- */
const char* boa_last_error(void);
int boa_version(void);
/* One context per GPU (one process per GPU).  stream == NULL -> the context creates its own stream,
 * otherwise it borrows the given hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int boa_init(int device, void* stream, boa_ctx** out);
void boa_destroy(boa_ctx* ctx);
int boa_device_info(boa_ctx* ctx, char* name, int name_len, int* cu_count, size_t* total_mem, size_t* free_mem);
/* Device memory for the caller's buffers (what `torch.empty(..., device="cuda")` / the caching allocator behind it is to the
 * reference).  Stream-ordered caching allocator: boa_free parks the block without synchronising the device, boa_malloc hands
 * a parked block of about the requested size out again; every use of a block must therefore be enqueued on the context's
 * stream (all boa_* calls are).  At most $BOA_POOL_GB (default 48; 0 = no caching) stays parked; boa_trim releases all of it.
 * Pointers that did not come from boa_malloc may be passed to boa_free (synchronise + hipFree). */
int boa_malloc(boa_ctx* ctx, size_t bytes, void** dev_out);
int boa_free(boa_ctx* ctx, void* dev);
int boa_trim(boa_ctx* ctx);
/* Several contexts may live on one GPU (each with its own stream and pool: boa_hip/lanes.py runs `total` and the BCA nets on two),
 * each driven by one host thread at a time; the pool itself is mutex-protected.  A host thread other than the one that called
 * boa_init makes the context's GPU its current device with this call before it drives the context (torch does the same per
 * thread with torch.cuda.set_device). */
int boa_bind_thread(boa_ctx* ctx);
/* Diagnostics (not on the data path): the rate a pure v_mfma_f32_32x32x16_f16 loop sustains on this GPU -- no memory, no LDS,
 * one wave per SIMD with 4 independent accumulator chains -- with near-constant operands (random_operands = 0) or operands whose
 * bits differ per lane and element (1).  The part is power-limited under matrix load: the second figure (1.5-1.7 PFLOP/s measured,
 * against 2.3-2.5 for the first and 2.5 on the data sheet) is the ceiling a conv on real activations can be priced against. */
int boa_mfma_peak(boa_ctx* ctx, int random_operands, int iters, double* tflops_out);
int boa_memset(boa_ctx* ctx, void* dev, int value, size_t bytes);
int boa_h2d(boa_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);   /* synchronous */
int boa_d2h(boa_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);   /* synchronous */
/* Page-locked host memory for the volumes that cross PCIe (what `tensor.pin_memory()` is to the reference): boa_h2d / boa_d2h
 * from / to such a buffer run at link speed instead of through the runtime's pageable staging copies.  boa_host_free accepts
 * ctx == NULL (the buffer may outlive the context that allocated it). */
int boa_host_alloc(boa_ctx* ctx, size_t bytes, void** host_out);
int boa_host_free(boa_ctx* ctx, void* host);
int boa_sync(boa_ctx* ctx);
/* HIP-event timing on the context stream: ms between the two marks (boa_timer_stop synchronises). */
int boa_timer_start(boa_ctx* ctx, int slot);
int boa_timer_stop(boa_ctx* ctx, int slot, float* ms_out);
/* Per-kernel-class accumulated time, measured with HIP events around every launch of that class while
 * profiling is enabled (bench.py uses this for `roofline.achieved`).  Classes: see BOA_K_*. */
#define BOA_K_CONV_MFMA 0
#define BOA_K_CONV_FIRST 1
#define BOA_K_CONVT 2
#define BOA_K_NORM_FINALIZE 3
#define BOA_K_HEAD_ACCUM 4
#define BOA_K_ARGMAX 5
#define BOA_K_OTHER 6      /* memset, CTNormalization */
#define BOA_K_AGG 7        /* tissue map / slice tables / label histograms / masks / erosion (agg.hip) */
#define BOA_K_MORPH 8      /* connected components, contour fill, median (morph.hip) */
#define BOA_K_RESAMPLE 9   /* cubic / nearest resampling (resample.hip) */
#define BOA_K_COPY 10      /* boa_copy3 index remaps (reorientation, transposes, crops) */
#define BOA_K_COUNT 11
int boa_prof_enable(boa_ctx* ctx, int on);
int boa_prof_reset(boa_ctx* ctx);
int boa_prof_get(boa_ctx* ctx, int kclass, double* total_ms, long long* launches, double* flops, double* bytes);
/* Which kernel VARIANT ran (always on, independent of profiling): number of launches since the context was created or the
 * counter was last reset.  The parity tests assert with these that the kernel they check is the one the product path and
 * bench.py launch (e.g. the MFMA head, not its fp32 VALU fallback).  Returns the count, < 0 on a bad argument. */
#define BOA_CNT_HEAD_MFMA 0       /* k_head_mfma (accumulate or logits mode)                           */
#define BOA_CNT_HEAD_VALU 1       /* k_head<F0, VPT> fallback (unaligned z origin / P2 % 32 / F0 = 64) */
#define BOA_CNT_CONV_WS 2         /* k_conv_ws (wave-specialised persistent conv)                     */
#define BOA_CNT_CONV_SIMPLE 3     /* k_conv_mfma (fallback for kernel shapes k_conv_ws lacks)          */
#define BOA_CNT_FIRST_MFMA 4      /* k_conv_first_mfma                                                 */
#define BOA_CNT_FIRST_VALU 5      /* k_conv_first<K> fallback                                          */
#define BOA_CNT_F32 6             /* any kernel of the fp32 reference network mode (precision = 1)     */
#define BOA_CNT_CONV_X3 7         /* k_conv_ws<..., X3>: split-precision conv (precision = 2)          */
#define BOA_CNT_X3 8              /* the other kernels of the split-precision mode (first conv, transposed conv, head) */
#define BOA_CNT_HEAD_GATHER 9     /* k_gather_head* (fused head over all covering tiles; raw partial sums in the tile-sharded path) */
#define BOA_CNT_COUNT 10
long long boa_debug_counter(boa_ctx* ctx, int which, int reset);

/* ------------------------------------------------------------------ sliding-window arithmetic seams -- */
/* CTNormalization.run (NN/preprocessing/normalization/default_normalization_schemes.py:53-67):
 * out = ((float)clip(in, lo, hi) - mean) / max(std, 1e-8) in fp32.  in_dtype: 0 = int16, 1 = float32, 2 = int32
 * (the dtype TS/nnunet.py:472-474 resamples to; the nibabel reader casts it to float32). */
int boa_ct_normalize(boa_ctx* ctx, const void* dev_in, int in_dtype, float* dev_out, size_t n,
                     float mean, float std, float lo, float hi);

/* One iteration of the tile loop, NN/inference/predict_from_raw_data.py:611-614, given the tile's logits:
 *   pred *= gauss (fp32*fp16->fp32); acc[:, sl] += pred (fp16 RTNE); n[sl] += gauss (fp16 RTNE).
 * pred:  dev fp32 [C][P0][P1][P2]; gauss: dev fp16 [P0][P1][P2] or NULL (use_gaussian=False: weight 1);
 * acc:   dev fp16 [C][V0][V1][V2]; n: dev fp16 [V0][V1][V2]; start: tile origin in the volume. */
int boa_accumulate_tile(boa_ctx* ctx, const float* dev_pred, const uint16_t* dev_gauss, uint16_t* dev_acc,
                        uint16_t* dev_n, int C, const int P[3], const int V[3], const int start[3]);

/* torch.div(logits, n, out=logits) + inf check (predict_from_raw_data.py:620-625), fold accumulation
 * (`prediction += ...`, :494-500), `/= n_folds`, numpy argmax(0) on fp16 with first-max / NaN-is-max semantics
 * (NN/utilities/label_handling/label_handling.py:175-178) and the TotalSegmentator part->global remap
 * `seg_combined[seg == jdx] = class_map_inv[name]` (TS/nnunet.py:553-556), fused in one pass over the volume.
 *
 *   acc, n      dev fp16 accumulators of THIS fold/model (acc is overwritten with acc/n when write_logits != 0)
 *   fold_sum    dev fp16 [C][V] running fold sum, or NULL.  When non-NULL: fold_mode 0 = store (first fold),
 *               1 = add (fp16 + fp16 -> fp16), and when n_folds_final > 0 the sum is divided by n_folds_final
 *               (fp16) before the argmax; the argmax is skipped when n_folds_final == 0 (more folds to come).
 *   lut         host uint8[256] mapping local argmax index -> output label (NULL = identity)
 *   merge       0: labels_out[v] = lut[argmax];  1: labels_out[v] = lut[argmax] only where argmax != 0
 *               (later parts overwrite earlier ones, background never overwrites)
 *   crop_off/crop_dims: revert pad_nd_image (predict_from_raw_data.py:657,679): labels_out is
 *               [crop_dims] and voxel (i,j,k) comes from padded voxel (i+off0, j+off1, k+off2); NULL = no crop.
 *   inf_flag    dev int32, set to 1 if any normalised logit is +-inf (caller raises, BOA_EINF semantics).
 */
int boa_finalize_labels(boa_ctx* ctx, uint16_t* dev_acc, const uint16_t* dev_n, int C, const int V[3],
                        uint16_t* dev_fold_sum, int fold_mode, int n_folds_final, int write_logits,
                        const uint8_t* host_lut, int merge, uint8_t* dev_labels_out,
                        const int* crop_off, const int* crop_dims, int* dev_inf_flag);
/* boa_finalize_labels restricted to planes [plane_lo, plane_hi) of axis 0 of the padded grid (the planes a rank owns in
 * the tile-sharded mode); acc / n / fold_sum keep the full-grid layout. */
int boa_finalize_labels_planes(boa_ctx* ctx, uint16_t* dev_acc, const uint16_t* dev_n, int C, const int V[3],
                               uint16_t* dev_fold_sum, int fold_mode, int n_folds_final, int write_logits,
                               const uint8_t* host_lut, int merge, uint8_t* dev_labels_out, const int* crop_off,
                               const int* crop_dims, int* dev_inf_flag, int plane_lo, int plane_hi);

/* ------------------------------------------------------------------ network (PlainConvUNet) ---------- */
/* Geometry of dynamic_network_architectures PlainConvUNet as the reference instantiates it from plans.json
 * (NN/utilities/plans_handling/plans_handler.py:59-92; NN/utilities/get_network_from_plans.py:34-38):
 * encoder stages of n_conv x [Conv3d(k, stride on first conv, pad (k-1)/2, bias) -> InstanceNorm3d(eps 1e-5,
 * affine) -> LeakyReLU(0.01)], decoder stages of ConvTranspose3d(k = s = stride, bias) -> cat(up, skip) ->
 * n_conv blocks, final 1x1x1 Conv3d head (deep supervision off, predict_from_raw_data.py:110). */
#define BOA_MAX_STAGES 8
typedef struct {
    int n_stages;
    int in_channels;
    int num_classes;
    int features[BOA_MAX_STAGES];
    int kernel[BOA_MAX_STAGES][3];
    int stride[BOA_MAX_STAGES][3];
    int n_conv_enc[BOA_MAX_STAGES];
    int n_conv_dec[BOA_MAX_STAGES]; /* n_stages - 1 entries, decoder order (deepest first) */
    int patch[3];
    float norm_eps;
    float lrelu_slope;
} boa_net_desc;

/* weights: host fp32 blob, tensors concatenated in this order, PyTorch memory layout:
 *   for each encoder stage s, conv i:  W[Cout][Cin][k0][k1][k2], b[Cout], gamma[Cout], beta[Cout]
 *   for each decoder stage d (deepest first): Wt[Cin][Cout][s0][s1][s2], bt[Cout], then for conv i: W, b, gamma, beta
 *   head: W[num_classes][features[0]], b[num_classes]
 * (state-dict keys encoder.stages.S.0.convs.I.{conv,norm}.*, decoder.transpconvs.D.*, decoder.stages.D.convs.I.*,
 *  decoder.seg_layers.<last>.*).
 * precision: 0 = fp16 weights / activations on the f16 matrix cores with fp32 accumulation (production; what the reference's
 * CUDA path computes under autocast, predict_from_raw_data.py:648); 1 = fp32 "exact" mode: fp32 weights, activations and
 * accumulation (v_mfma_f32_32x32x2_f32) = what the reference's CPU path computes; a correctness mode, ~30x slower. */
int boa_net_create(boa_ctx* ctx, const boa_net_desc* desc, const float* host_weights, size_t n_floats,
                   int max_batch, int precision, boa_net** out);
void boa_net_destroy(boa_net* net);
size_t boa_net_weight_count(const boa_net_desc* desc); /* expected n_floats, 0 on invalid desc */
/* swap weights (next fold), NN/inference/predict_from_raw_data.py:486-489 `load_state_dict(params)` */
int boa_net_load_weights(boa_net* net, const float* host_weights, size_t n_floats);

/* Test-time mirroring, `_internal_maybe_mirror_and_predict` (predict_from_raw_data.py:541-557): with axes_mask != 0 (bit a =
 * array axis a of the allowed mirroring axes) every tile's logits become the fp32 mean over the plain forward and the
 * forwards of all non-empty axis combinations of the flipped tile, flipped back -- (2^axes) x the network cost.  BOA runs
 * every model with tta=False (TS/python_api.py:753); the switch exists because nnUNetPredictor(use_mirroring=True) does. */
int boa_net_set_mirroring(boa_net* net, int axes_mask);

/* network(x) for a batch of tiles gathered from a resident volume
 * (replaces `self.network(workon)` predict_from_raw_data.py:543 + producer thread :568-571):
 *   volume   dev fp32 [Cin][V0][V1][V2] (already normalised); tile voxels outside the volume read as 0
 *            (pad_nd_image) -- origins may be negative / overhang for volumes smaller than the patch.
 *   origins  host int[n_tiles][3]
 *   logits   dev fp32 [n_tiles][num_classes][P0][P1][P2]  (PyTorch NCDHW)  */
int boa_net_forward(boa_net* net, const float* dev_volume, const int V[3], const int* host_origins,
                    int n_tiles, float* dev_logits_out);

/* Whole `_internal_predict_sliding_window_return_logits` loop (predict_from_raw_data.py:560-631) for one
 * fold: for every tile in the given (canonical x->y->z) order: forward, head, *gauss, fp16 accumulate.
 * acc/n must be zeroed by the caller (boa_memset).  vol_off: position of the real volume inside the padded
 * accumulator grid (pad_nd_image `below`), NULL = {0,0,0}.  Tiles are processed in batches of <= max_batch
 * through the conv stack; accumulation is applied strictly in the given order.
 *   PV  padded accumulator dims (>= patch), V real volume dims. */
int boa_net_predict_sliding_window(boa_net* net, const float* dev_volume, const int V[3], const int PV[3],
                                   const int* vol_off, const int* host_origins, int n_tiles,
                                   const uint16_t* dev_gauss, uint16_t* dev_acc, uint16_t* dev_n);

/* Fused form of the same loop for the label-only path (csrc/head_gather.hip): the conv stack leaves the last decoder activation
 * of EVERY tile in a stash, then one pass over the volume walks, per voxel, the covering tiles in ascending tile index (the
 * reference's `+=` order, predict_from_raw_data.py:611-614) with the running fp16 sums in registers and goes on to the
 * normalisation, fold sum / mean (:483-500), argmax, label remap and crop of boa_finalize_labels -- the accumulator planes never
 * exist.  Same arithmetic and rounding sequence as boa_net_predict_sliding_window + boa_finalize_labels (labels bit-identical;
 * tests/test_gpu_gather_head.py).  One call per fold (after boa_net_load_weights), fold_index ascending:
 *   dev_fold        fp16 [C][PV] scratch, n_folds > 1 only (the fold sum lives there between the calls)
 *   dev_labels_out  written by the LAST fold's call (merge / lut / crop_off / crop_dims as in boa_finalize_labels)
 *   dev_inf_flag    set non-zero if a normalised logit is +-inf (:622-625)
 * boa_net_labels_supported: 1 when the network / tile layout qualifies (production precision, no mirroring, 32 features at
 * full resolution, <= 31 classes, patch z extent a multiple of 32, tile origins = the cartesian grid of compute_steps_for_sliding_window in canonical order).
 * Returns BOA_ENOMEM when the stash (n_tiles x patch voxels x 64 B) does not fit: the caller falls back to the loop above. */
int boa_net_labels_supported(boa_net* net, const int* host_origins, int n_tiles);
int boa_net_predict_labels_fold(boa_net* net, const float* dev_volume, const int V[3], const int PV[3], const int* vol_off,
                                const int* host_origins, int n_tiles, const uint16_t* dev_gauss, uint16_t* dev_fold,
                                int fold_index, int n_folds, const uint8_t* host_lut, int merge, uint8_t* dev_labels_out,
                                const int* crop_off, const int* crop_dims, int* dev_inf_flag);

/* ---- one volume on several GPUs (SURVEY 8e "tile partitioning"): each rank owns a block of tile rows along axis 0 ----
 * Same loop as boa_net_predict_sliding_window over THIS rank's tiles, except that for tile i the first
 * host_defer_planes[i] planes (axis 0) are not accumulated: their head input is kept in *stash_out.  The caller places
 * the lower rank's partial sums for those planes into acc / n (they were received over RCCL/xGMI) and then calls
 * boa_net_apply_deferred, which adds the stashed planes in the original tile order -- per voxel the fp16 `+=` sequence
 * is the reference's (predict_from_raw_data.py:611-614), so the result is bit-identical to the single-GPU loop.
 * The stash remembers the head weights of the fold that produced it (it may be applied after the network switched folds).
 * acc / n must be zero on entry in the planes this rank's tiles cover: when the network and tile grid qualify
 * (boa_net_labels_supported) the non-deferred planes are WRITTEN by one gather-head launch (raw partial sums, all covering
 * tiles of this rank in ascending order) and boa_net_apply_deferred is one more launch of that kernel over the deferred
 * planes, started from the planes' contents; otherwise both run the per-tile scatter head.  With host_defer_planes all
 * zero this is the single-GPU accumulate loop in its gather form (the resampled label path uses it that way). */
typedef struct boa_stash boa_stash;
int boa_net_predict_sliding_window_deferred(boa_net* net, const float* dev_volume, const int V[3], const int PV[3],
                                            const int* vol_off, const int* host_origins, int n_tiles,
                                            const uint16_t* dev_gauss, uint16_t* dev_acc, uint16_t* dev_n,
                                            const int* host_defer_planes, boa_stash** stash_out);
int boa_net_apply_deferred(boa_net* net, const boa_stash* stash, const uint16_t* dev_gauss, uint16_t* dev_acc,
                           uint16_t* dev_n, const int PV[3]);
void boa_stash_destroy(boa_stash* stash);

/* Debug / unit-test seam: run ONE conv block on device tensors in PyTorch layout.
 *   in  dev fp32 [N][Cin][D][H][W], w/b/gamma/beta host fp32; out dev fp32 [N][Cout][Do][Ho][Wo] =
 *   LeakyReLU(InstanceNorm(Conv3d(in))) when with_norm_act != 0, else the raw convolution (+bias).
 * impl: 0 = MFMA implicit-GEMM kernel, 1 = naive direct kernel (on-device cross-check). */
int boa_conv_block_test(boa_ctx* ctx, const float* dev_in, int N, int Cin, const int dims[3],
                        const float* host_w, const float* host_b, const float* host_gamma, const float* host_beta,
                        int Cout, const int kernel[3], const int stride[3], int with_norm_act, int impl,
                        float* dev_out); /* impl must be 0 */
/* Debug seam for the per-layer error report (fp16 production mode against the fp32 exact mode, tools/layer_error.py): the
 * activation a layer left behind for `tile` of the LAST batch that went through the conv stack, as fp32 [C][D][H][W] with the
 * layer's InstanceNorm + LeakyReLU applied (i.e. what the next layer consumes; transposed convs have no norm: raw output).
 * kind 0 = encoder conv (stage, conv), 1 = transposed conv (stage = decoder index, deepest first), 2 = decoder conv.
 * dev_out == NULL only returns channels / dims. */
int boa_net_debug_activation(boa_net* net, int kind, int stage, int conv, int tile, float* dev_out, int* channels_out,
                             int dims_out[3]);

/* Unit-test seam: the fused 1x1x1 head + Gaussian-weighted fp16 accumulation of ONE tile, launched exactly as the tile
 * loop of boa_net_predict_sliding_window launches it (same kernel selection: the MFMA head when F0 == 32, C <= 31,
 * P[2] % 32 == 0 and the z origin / extent are 8-voxel aligned, else the fp32 VALU head; boa_debug_counter tells which).
 *   act  dev fp16 [F0/16][P0][P1][P2][16] (the engine's chunk-planar activation layout): the last decoder conv's raw output;
 *        ss dev fp32 [F0][2] its InstanceNorm (scale, shift)
 *   w    dev fp32 [C][F0], b dev fp32 [C]
 *   logits_out != NULL: write the tile's fp32 logits [C][P0][P1][P2] (what `self.network(x)` returns, :543);
 *   logits_out == NULL: `pred *= gauss; acc[sl] += pred; n[sl] += gauss` (:611-614) on acc [C][PV] / n [PV] at `start`.
 * The two modes run the same instruction sequence up to the logit, so accumulating tiles and comparing with the oracle's
 * accumulate step applied to the logits-mode output checks the production accumulate arithmetic bit for bit. */
int boa_head_tile(boa_ctx* ctx, const uint16_t* dev_act, const float* dev_ss, int F0, const int P[3], int C,
                  const float* dev_w, const float* dev_b, float slope, float* dev_logits_out, const uint16_t* dev_gauss,
                  uint16_t* dev_acc, uint16_t* dev_n, const int PV[3], const int start[3]);
int boa_convtranspose_test(boa_ctx* ctx, const float* dev_in, int N, int Cin, const int dims[3],
                           const float* host_w, const float* host_b, int Cout, const int stride[3], float* dev_out);

/* ------------------------------------------------------------------ body-composition / HU aggregation - */
/* subclassify_tissues (BCA/tissue/subclassification.py:38-53, rules BCA/tissue/definition.py:6-30) fused with
 * the slice-wise voxel counts of Builder.prepare (BCA/report/builder.py:403-444) and the per-slice HU sums that
 * `np.mean(image[tissue_mask])` (builder.py:284-305) needs.
 *   ct int16 [Z][Y][X], regions uint8, parts uint8 (may be NULL -> no torso counts), tissues_out uint8 (may be NULL)
 *   ct_rules (may be NULL = ct): the HU the derivation rules look at -- the 3x3 in-plane median-filtered CT when
 *           median_filtering is on (subclassification.py:21-36); HU sums always come from `ct` (builder.py gets the
 *           unfiltered image)
 *   counts  dev uint32 [Z][2][8]   (index 0 unused; [0] = all voxels, [1] = body_parts == TORSO(1))
 *   hu_sums dev int64  [Z][2][8]
 * counts/hu_sums are overwritten. */
int boa_tissue_aggregate(boa_ctx* ctx, const int16_t* dev_ct, const int16_t* dev_ct_rules, const uint8_t* dev_regions,
                         const uint8_t* dev_parts,
                         uint8_t* dev_tissues_out, int Z, int Y, int X, uint32_t* dev_counts, int64_t* dev_hu_sums);

/* Per-slice presence of each label value (np.where(mask.any(axis=(1,2))) in builder.py:56-100,170-199 and
 * BCA/commands.py:24-45): present[z][label] = 1 if any voxel of slice z has that label. dev uint8 [Z][256]. */
int boa_slice_label_presence(boa_ctx* ctx, const uint8_t* dev_labels, int Z, int Y, int X, uint8_t* dev_present);

/* The reductions behind the report's coronal / sagittal tissue heat maps (BCA/report/plots/heatmaps.py:29-101), one pass
 * over the (z,y,x) uint8 volumes: for each tissue value v_t (host_values, 1..16 of them)
 *   coronal[t][z][x]  = #{y : tissues[z,y,x] == v_t}     (`tissue_mask.sum(axis=1)`)
 *   sagittal[t][z][y] = #{x : tissues[z,y,x] == v_t}     (`tissue_mask.sum(axis=2)`)
 * and the body silhouettes `((regions > 0) & (regions < 255)).any(axis)`: mask_coronal[z][x], mask_sagittal[z][y] (0/1).
 * Colour mapping / resizing / contour drawing of the report stay on the host (out of scope). */
int boa_tissue_projections(boa_ctx* ctx, const uint8_t* dev_tissues, const uint8_t* dev_regions, int Z, int Y, int X,
                           const uint8_t* host_values, int n_values, uint32_t* dev_coronal, uint32_t* dev_sagittal,
                           uint8_t* dev_mask_coronal, uint8_t* dev_mask_sagittal);

/* One pass replacing the ~125 full-volume passes of metrics_for_each_region (BOA/compute/measurements.py:74-123,
 * 203-241): histogram of HU per label.  hist dev uint32 [256][nbins], bin = clamp(hu - hu_min, 0, nbins-1);
 * mask (optional dev uint8, same shape): only voxels with mask != 0 are counted (eroded / fat-window masks).
 * hist is overwritten.  Exact order statistics (median, percentiles), mean, std, min, max follow on the host
 * from integer counts. */
int boa_label_hu_histogram(boa_ctx* ctx, const int16_t* dev_ct, const uint8_t* dev_labels, const uint8_t* dev_mask,
                           size_t n_voxels, int hu_min, int nbins, uint32_t* dev_hist);

/* mask_out = (lut[labels] != 0) && (hu_lo <= ct <= hu_hi  [in_range] | ct < hu_lo || ct > hu_hi [!in_range]);
 * get_region_minus_fat / compute_lung_measurement fat masks (BOA/compute/measurements.py:29-39,126-148).
 * lut: host uint8[256]; mode 0 = label only, 1 = inside [lo,hi], 2 = outside [lo,hi]. */
int boa_label_hu_mask(boa_ctx* ctx, const int16_t* dev_ct, const uint8_t* dev_labels, const uint8_t* host_lut,
                      int mode, int hu_lo, int hu_hi, size_t n_voxels, uint8_t* dev_mask_out);

/* erode_region (BOA/compute/measurements.py:61-71): binary erosion with a k^3 ones footprint whose anchor is
 * the centre of the end-padded (k+1)^3 footprint for even k; voxels outside the volume count as set.
 * Separable min filter, three passes. tmp: dev uint8 scratch of the same size. */
int boa_binary_erode(boa_ctx* ctx, const uint8_t* dev_mask, uint8_t* dev_out, uint8_t* dev_tmp, int Z, int Y, int X,
                     int kernel_value);

/* Connected components, 26-connectivity (skimage.measure.label default for 3-D inputs,
 * BCA/body_regions/postprocess.py:9; remove_small_objects(connectivity=3), BCA/body_parts/postprocess.py:43-48)
 * of `mask != 0` by atomic union-find on the device.
 *   roots dev int32 [n]: linear index of the component's first voxel in raster order (= the order in which
 *         skimage numbers components), -1 for background;
 *   sizes dev uint32 [n]: component size stored at the root's index, 0 elsewhere;
 *   host_n_components (optional): number of components. */
int boa_ccl26(boa_ctx* ctx, const uint8_t* dev_mask, int Z, int Y, int X, int32_t* dev_roots, uint32_t* dev_sizes,
              int* host_n_components);
/* _filter_largest_unique_segment (BCA/body_regions/postprocess.py:8-15): every component except the largest
 * (ties: the one numbered first, i.e. smallest root) gets seg = fill_value (255). No-op for <= 1 component. */
int boa_ccl_filter_largest(boa_ctx* ctx, const int32_t* dev_roots, const uint32_t* dev_sizes, size_t n,
                           uint8_t* dev_seg, int fill_value);
/* remove_small_objects(mask, max_size): clear mask where the component has <= max_size voxels. */
int boa_ccl_remove_small(boa_ctx* ctx, const int32_t* dev_roots, const uint32_t* dev_sizes, size_t n,
                         uint32_t max_size, uint8_t* dev_mask_inout);
/* ---- z-slab sharded connected components: per-component tables for the host-side merge over slab interfaces ----
 * boa_ccl_list_components: the (root index, size) pairs of boa_ccl26's result, in no particular order; *host_count is the
 *   true number (may exceed max_out: call again with larger arrays).  Synchronous.
 * boa_scatter_u32: dst[idx[i]] = val[i] (the merged sizes of the components that cross an interface go back into `sizes`,
 *   so that boa_ccl_remove_small decides on the global size).  Synchronous.
 * boa_ccl_fill_unmarked: seg = fill_value on every component whose sizes[root] != mark (_filter_largest_unique_segment when
 *   the largest component was determined globally: its local pieces carry `mark`). */
int boa_ccl_list_components(boa_ctx* ctx, const uint32_t* dev_sizes, size_t n, int max_out, int32_t* host_roots,
                            uint32_t* host_sizes, int* host_count);
int boa_scatter_u32(boa_ctx* ctx, uint32_t* dev_dst, const int32_t* host_idx, const uint32_t* host_val, int m);
int boa_ccl_fill_unmarked(boa_ctx* ctx, const int32_t* dev_roots, const uint32_t* dev_sizes, size_t n, uint32_t mark,
                          uint8_t* dev_seg, int fill_value);
/* mask_out = (labels == value) [mode 0] | (lut-free) labels > 0 [mode 1] | labels in {a,b,c} [mode 2, vals[3]] */
int boa_label_select(boa_ctx* ctx, const uint8_t* dev_labels, size_t n, int mode, const int vals[3],
                     uint8_t* dev_mask_out);

/* remove_small_labeled_objects, slice-wise contour fill (BCA/body_parts/postprocess.py:31-39: per (y,x) slice
 * cv2.findContours(RETR_EXTERNAL) + drawContours(FILLED)): out = mask != 0 OR background pixel that is not 4-connected
 * to the slice border through background.  scratch_i32: dev int32 [n]; scratch_u8: dev uint8 [n]; out must not alias. */
int boa_fill_holes_2d(boa_ctx* ctx, const uint8_t* dev_mask, int Z, int Y, int X, int32_t* dev_scratch_i32,
                      uint8_t* dev_scratch_u8, uint8_t* dev_out);
/* remove_outside_of_mask (TS/postprocessing.py:101-131, `heartchambers_highres`): scipy.ndimage.binary_dilation(mask,
 * iterations) with the default 6-neighbour cross and border_value 0; the caller then clears the labels where the result is
 * 0 (boa_mask_assign with invert = 1).  iterations >= 1; tmp: dev uint8 scratch of the same size; no aliasing. */
int boa_binary_dilate_cross(boa_ctx* ctx, const uint8_t* dev_mask, uint8_t* dev_out, uint8_t* dev_tmp, int Z, int Y, int X,
                            int iterations);
/* `out[filled] = label` (BCA/body_parts/postprocess.py:50): out[i] = value where (mask[i] != 0) != invert. */
int boa_mask_assign(boa_ctx* ctx, const uint8_t* dev_mask, size_t n, int invert, int value, uint8_t* dev_out);
/* ---- bit-mask morphology (csrc/ccl_bits.hip, round 5): the same filters on BIT-PACKED masks, batched over independent masks ----
 * A mask is [Z][Y][W] uint32 words, W = ceil(X / 32), bit i of word w <-> x = 32 w + i (bits beyond X are 0); a batch of M masks is
 * M consecutive masks (boa_bits_words words each).  Replaces, for the BCA post-processing (BCA/body_parts/postprocess.py:7-52,
 * BCA/body_regions/postprocess.py:8-40), the per-label chains of boa_label_select / boa_fill_holes_2d / boa_ccl26 /
 * boa_ccl_remove_small / boa_mask_assign: the per-label passes of remove_small_labeled_objects all read the ORIGINAL mask
 * (`label_mask = mask == label`) and are independent until `out[filled] = label`.  Components are 26-connected; a component's size and
 * first voxel (raster order: the tie-break of the largest-component filter) are those of boa_ccl26.
 *   boa_bits_words          words per mask
 *   boa_bits_select         mask m = { voxel : host_lut[label] bit m } for m < n_masks <= 8, one pass over the uint8 label volume
 *                           (`mask == label`, `seg > 0`, `seg in {a, b, c}` are all such look-ups)
 *   boa_bits_unpack         one mask -> uint8 0 / 1 volume (tests, interop with the byte-mask kernels)
 *   boa_bits_fill_holes_2d  per (mask, z slice): cv2.findContours(RETR_EXTERNAL) + drawContours(FILLED) = foreground + background not
 *                           4-connected to the slice border; in and out may not alias; boa_bits_fill_supported(Y, X) == 0: the slice
 *                           does not fit the LDS flood (use boa_fill_holes_2d on bytes)
 *   boa_bits_remove_small   remove_small_objects(max_size, connectivity = 3) in place on every mask; invert != 0: on the complements
 *                           (np.invert / remove_small_objects / np.invert: small holes are filled)
 *   boa_bits_filter_largest _filter_largest_unique_segment on ONE mask: seg = fill_value outside its largest component
 *   boa_bits_assign_labels  out[v] = host_labels[m] for the largest m whose mask holds v (`out[filled] = label`, ascending); voxels in no
 *                           mask of the batch keep their value: zero `out` first, apply batches of <= 8 labels in ascending order
 *   boa_bits_erode_u8       uint8 mask -> uint8 0 / 1 mask eroded by the box of offsets [lo, hi] per axis (outside the volume counts as
 *                           set): what boa_binary_erode runs (pack, three separable passes on bits, unpack) */
size_t boa_bits_words(int Z, int Y, int X);
int boa_bits_erode_u8(boa_ctx* ctx, const uint8_t* dev_mask, uint8_t* dev_out, int Z, int Y, int X, int lo, int hi);
int boa_bits_select(boa_ctx* ctx, const uint8_t* dev_seg, int Z, int Y, int X, const uint8_t host_lut[256], int n_masks, uint32_t* dev_bits);
int boa_bits_unpack(boa_ctx* ctx, const uint32_t* dev_bits, int Z, int Y, int X, uint8_t* dev_out);
int boa_bits_fill_supported(int Y, int X);
int boa_bits_fill_holes_2d(boa_ctx* ctx, const uint32_t* dev_in, int Z, int Y, int X, int n_masks, uint32_t* dev_out);
int boa_bits_remove_small(boa_ctx* ctx, uint32_t* dev_bits, int Z, int Y, int X, int n_masks, uint32_t max_size, int invert);
int boa_bits_filter_largest(boa_ctx* ctx, const uint32_t* dev_bits, int Z, int Y, int X, uint8_t* dev_seg, int fill_value);
int boa_bits_assign_labels(boa_ctx* ctx, const uint32_t* dev_bits, int Z, int Y, int X, int n_masks, const uint8_t* host_labels, uint8_t* dev_out);
/* the part -> combined merge `seg_combined[seg == jdx] = class_map_inv[name]` (TS/nnunet.py:553-556) on label volumes that
 * already carry global labels: out[i] = part[i] where part[i] != 0 (tile-sharded mode merges after the label exchange). */
int boa_label_overlay(boa_ctx* ctx, const uint8_t* dev_part, size_t n, uint8_t* dev_out);
/* subclassify_tissues(median_filtering=True) (BCA/tissue/subclassification.py:21-36): scipy.ndimage.median_filter
 * with size 3 on two axes and 1 on `flat_axis` (0 = z, 1 = y, 2 = x of the [Z][Y][X] array), mode="reflect". */
int boa_median3_inplane(boa_ctx* ctx, const int16_t* dev_in, int Z, int Y, int X, int flat_axis, int16_t* dev_out);

/* ------------------------------------------------------------------ index remaps around a task ------- */
/* Strided 3-D gather copy with dtype conversion: the device form of as_closest_canonical / undo_canonical
 * (TS/alignment.py:8-46: axis permutation + flips), crop_to_bbox / undo_crop (TS/cropping.py:41-49,126-132), the
 * (x,y,z) <-> (z,y,x) view change between nibabel and nnU-Net arrays (NN/imageio/nibabel_reader_writer.py:51-56),
 * crop_to_nonzero / insert_crop_into_image, and the z-splits of TS/nnunet.py:495-505,583-586.
 *   out[out_off + sum_k o_k * out_step[k]] = (out type) in[in_off + sum_k o_k * in_step[k]]   for o in [0, dims)
 * Offsets / steps are in elements (steps may be negative).  dtype codes: 0 uint8, 1 int16, 2 int32, 3 float32,
 * 4 float64; float -> integer conversion truncates (numpy astype). */
int boa_copy3(boa_ctx* ctx, const void* dev_in, int in_dtype, long long in_off, const long long in_step[3],
              const int dims[3], void* dev_out, int out_dtype, long long out_off, const long long out_step[3]);
/* Bounding box of data != 0 (crop_to_nonzero, NN/preprocessing/cropping/cropping.py:6-29; binary_fill_holes cannot
 * change it): host_bbox = {lo0, hi0, lo1, hi1, lo2, hi2} with hi exclusive; [0, dim) when all zero.
 * dtype: 0 uint8, 1 int16, 2 int32, 3 float32.  Synchronous. */
int boa_nonzero_bbox(boa_ctx* ctx, const void* dev_in, int dtype, const int dims[3], int host_bbox[6]);

/* ------------------------------------------------------------------ resampling (TS/resampling.py) --- */
/* change_spacing / resample_img order 3 (TS/resampling.py:24-56,129-222): scipy.ndimage.zoom(data, zoom, order=3,
 * mode="nearest") restated in fp64 (edge pad 12, cubic B-spline prefilter with 'reflect' initialisation, 64-tap
 * interpolation, coordinate = out * (n_in - 1) / (n_out - 1)); only the shapes determine the sampling grid.
 * in_dtype: 0 int16, 1 float32, 2 float64, 3 int32; out_dtype: 0 int32 (`.astype(np.int32)` truncation), 1 float64.
 * Every fp64 operation in scipy's order with scipy's constants: bit-identical to scipy 1.15.3 on the golden vectors
 * (tests/golden/g5_resample.npz) including the int32 truncation.  Synchronous (frees its fp64 scratch of (X+24)(Y+24)(Z+24) doubles). */
int boa_resample_cubic(boa_ctx* ctx, const void* dev_in, int in_dtype, const int in_dims[3], void* dev_out, int out_dtype,
                       const int out_dims[3]);
/* order 0 (labels back to the original grid, TS/nnunet.py:685-687): index floor(out * (n_in-1)/(n_out-1) + 0.5). */
int boa_resample_nearest_u8(boa_ctx* ctx, const uint8_t* dev_in, const int in_dims[3], uint8_t* dev_out,
                            const int out_dims[3]);


/* ------------------------------------------------------------------ nnU-Net's own resampling to / from the plans' spacing --- */
/* resampling_fn_data (NN/preprocessing/preprocessors/default_preprocessor.py:82-93 -> NN/preprocessing/resampling/
 * default_resampling.py:113-196, is_seg False, order 3, order_z 0): skimage.transform.resize(mode="edge",
 * anti_aliasing=False, clip=True) restated from its published algorithm (scipy.ndimage.zoom(grid_mode=True,
 * mode="nearest") + clip to the input's range; skimage itself is absent from the build image: unpinned) in fp64, result
 * cast to float32.  slice_axis = -1: one 3-D resize; 0..2 ("separate z", anisotropic spacing): every slice along that axis
 * is resized in 2-D (clip range per slice) and the axis itself is sampled nearest (map_coordinates order 0) when its
 * extent changes.  in / out: dev fp32 [d0][d1][d2].  Synchronous (frees its fp64 scratch). */
int boa_resize_skimage_f32(boa_ctx* ctx, const float* dev_in, const int in_dims[3], float* dev_out, const int out_dims[3],
                           int order, int slice_axis);
/* resampling_fn_probabilities + argmax (NN/inference/export_prediction.py:25-47): the (fold-mean) fp16 logits
 * [C][grid_dims], of which the box crop_off / crop_dims is the network's output for the resampled image (pad_nd_image
 * reverted), are resampled with order 1 to out_dims (slice_axis as above), rounded to fp16 (the reference's result array
 * has the logits' dtype) and reduced with numpy's argmax semantics; lut / merge as boa_finalize_labels.  The [C][out_dims]
 * tensor is never materialised.  (skimage's clip is a no-op for order 1 up to one fp64 rounding and is not applied.) */
int boa_resize_logits_argmax(boa_ctx* ctx, const uint16_t* dev_logits, int C, const int grid_dims[3], const int* crop_off,
                             const int* crop_dims, const int out_dims[3], int slice_axis, const uint8_t* host_lut, int merge,
                             uint8_t* dev_labels_out);

/* ------------------------------------------------------------------ several GPUs on one volume (RCCL) -- */
/* The tile-sharded sliding window (boa_hip/tile_shard.py): ranks own blocks of tile rows of the sliding window
 * (NN/inference/predict_from_raw_data.py:523-558 orders the tiles axis 0 outermost; :611-614 adds them one after the other into
 * fp16 buffers) and exchange the partial sums of the planes where two blocks overlap.  The reference has no counterpart (one
 * device per process); these entry points carry its accumulate loop across devices.  RCCL is resolved at run time (dlopen;
 * $BOA_RCCL_LIB overrides the search), collectives run on a communication stream of the boa_comm that is ordered against the
 * context's compute stream by events: boa_comm_* calls queue work and return, boa_comm_wait makes LATER compute work wait for
 * what has been queued (no host synchronisation anywhere).
 *   boa_comm_unique_id : ncclGetUniqueId on one rank; the 128 bytes reach the other ranks out of band (launcher / store)
 *   boa_comm_create    : ncclCommInitRank (collective over all ranks)
 *   boa_comm_shift_slab: planes [send_lo, send_hi) of the C class planes of `acc` + of `nacc` (fp16 [.][PV0][PV1][PV2]: one
 *                        contiguous run per class, sent in place) -> rank dst; planes [recv_lo, recv_hi) <- rank src, written in
 *                        place (recv_stage NULL: the exact hand-over, the receiver has not added anything there yet) or into
 *                        recv_stage [(C + 1)][planes][PV1 PV2] for boa_add_f16_planes (the pairwise fp16 sum).  dst / src < 0: none
 *   boa_comm_exchange  : generic grouped send / recv of byte ranges (pieces matched in order)
 *   boa_comm_planes_to_owner: plane ranges of the C class planes of fp16 logits [C][PV0][PV1][PV2] to / from SEVERAL peers, in place:
 *                        message i of the send list = planes [send_lo[i], send_hi[i]) of every class -> rank send_peer[i], likewise the
 *                        receive list (disjoint from what this rank keeps).  The reduce-scatter of plane-disjoint fold logits
 *                        (predict_from_raw_data.py:494-500 adds the folds; every rank only finalises its own plane share): each plane
 *                        travels once, to the rank that finalises it -- half the per-link bytes of the all-reduce.  Classes go out in
 *                        sub-groups whose size depends on $BOA_COMM_GROUP and the world size only, so all ranks walk the same groups
 *   boa_comm_all_reduce: in-place sum over all ranks; dtype 0 uint8, 1 fp16, 2 int32, 3 fp32 (label volumes with disjoint
 *                        supports: TS/nnunet.py:553-556 merges the parts afterwards; inf flags; plane-disjoint logits) */
typedef struct boa_comm boa_comm;
int boa_comm_available(void);
const char* boa_comm_library(void);
int boa_comm_unique_id(unsigned char id_out[128]);
int boa_comm_create(boa_ctx* ctx, int world, int rank, const unsigned char id[128], boa_comm** out);
void boa_comm_destroy(boa_comm* comm);
int boa_comm_wait(boa_comm* comm);
int boa_comm_exchange(boa_comm* comm, int dst, const void* const* send_ptrs, const size_t* send_bytes, int n_send, int src,
                      void* const* recv_ptrs, const size_t* recv_bytes, int n_recv);
int boa_comm_shift_slab(boa_comm* comm, int dst, int send_lo, int send_hi, int src, int recv_lo, int recv_hi, uint16_t* dev_acc,
                        uint16_t* dev_n, int C, const int PV[3], uint16_t* recv_stage);
int boa_comm_all_reduce(boa_comm* comm, void* dev, size_t count, int dtype);
int boa_comm_planes_to_owner(boa_comm* comm, uint16_t* dev_logits, int C, const int PV[3], int n_send, const int* send_peer,
                             const int* send_lo, const int* send_hi, int n_recv, const int* recv_peer, const int* recv_lo,
                             const int* recv_hi);
int boa_comm_stats(boa_comm* comm, long long* calls, long long* bytes);
/* acc[k][lo:hi] = half(float(acc[k][lo:hi]) + float(stage[k])), k = 0 .. C (C = the n plane): the owner's side of the pairwise
 * fp16 sum of a slab ("allreduce" exchange mode) */
int boa_add_f16_planes(boa_ctx* ctx, uint16_t* dev_acc, uint16_t* dev_n, const uint16_t* dev_stage, int C, const int PV[3], int lo, int hi);

#ifdef __cplusplus
}
#endif
#endif /* BOA_HIP_H */
