"""ctypes binding of libboa_hip.so (see include/boa_hip.h).  There is no CPU fallback: if the library is
missing or fails to load, importing the compute path raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# $BOA_HIP_LIB: load another build of the same library (kernel experiments: tools/build_alt.sh)
LIB_PATH = os.environ.get("BOA_HIP_LIB") or os.path.join(_HERE, "libboa_hip.so")

BOA_OK, BOA_EINVAL, BOA_EHIP, BOA_ENOMEM, BOA_EINF = 0, -1, -2, -3, -4
K_CONV_MFMA, K_CONV_FIRST, K_CONVT, K_NORM_FINALIZE, K_HEAD_ACCUM, K_ARGMAX, K_OTHER, K_AGG, K_MORPH, K_RESAMPLE, K_COPY, K_COUNT = range(12)
K_NAMES = ["conv_mfma", "conv_first", "convT_mfma", "norm_finalize", "head_accum", "finalize_argmax", "other", "aggregation",
           "morphology", "resample", "copy_remap"]
CNT_NAMES = ["head_mfma", "head_valu", "conv_ws", "conv_simple", "first_mfma", "first_valu", "f32", "conv_x3", "x3", "head_gather"]
MAX_STAGES = 8


class NetDesc(C.Structure):
    _fields_ = [
        ("n_stages", C.c_int), ("in_channels", C.c_int), ("num_classes", C.c_int),
        ("features", C.c_int * MAX_STAGES), ("kernel", (C.c_int * 3) * MAX_STAGES),
        ("stride", (C.c_int * 3) * MAX_STAGES), ("n_conv_enc", C.c_int * MAX_STAGES),
        ("n_conv_dec", C.c_int * MAX_STAGES), ("patch", C.c_int * 3), ("norm_eps", C.c_float),
        ("lrelu_slope", C.c_float),
    ]


class BoaError(RuntimeError):
    pass


_lib = None

vp, i32, u64, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
ip = C.POINTER(C.c_int)

_PROTOS = {
    "boa_last_error": (C.c_char_p, []),
    "boa_version": (i32, []),
    "boa_init": (i32, [i32, vp, C.POINTER(vp)]),
    "boa_destroy": (None, [vp]),
    "boa_device_info": (i32, [vp, C.c_char_p, i32, ip, C.POINTER(u64), C.POINTER(u64)]),
    "boa_malloc": (i32, [vp, u64, C.POINTER(vp)]),
    "boa_free": (i32, [vp, vp]),
    "boa_trim": (i32, [vp]),
    "boa_bind_thread": (i32, [vp]),
    "boa_host_alloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
    "boa_host_free": (i32, [vp, vp]),
    "boa_mfma_peak": (i32, [vp, i32, i32, C.POINTER(C.c_double)]),
    "boa_memset": (i32, [vp, vp, i32, u64]),
    "boa_h2d": (i32, [vp, vp, vp, u64]),
    "boa_d2h": (i32, [vp, vp, vp, u64]),
    "boa_sync": (i32, [vp]),
    "boa_timer_start": (i32, [vp, i32]),
    "boa_timer_stop": (i32, [vp, i32, C.POINTER(f32)]),
    "boa_prof_enable": (i32, [vp, i32]),
    "boa_prof_reset": (i32, [vp]),
    "boa_prof_get": (i32, [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                           C.POINTER(C.c_double)]),
    "boa_debug_counter": (C.c_longlong, [vp, i32, i32]),
    "boa_net_debug_activation": (i32, [vp, i32, i32, i32, i32, vp, ip, ip]),
    "boa_head_tile": (i32, [vp, vp, vp, i32, ip, i32, vp, vp, f32, vp, vp, vp, vp, ip, ip]),
    "boa_ct_normalize": (i32, [vp, vp, i32, vp, u64, f32, f32, f32, f32]),
    "boa_accumulate_tile": (i32, [vp, vp, vp, vp, vp, i32, ip, ip, ip]),
    "boa_finalize_labels": (i32, [vp, vp, vp, i32, ip, vp, i32, i32, i32, vp, i32, vp, ip, ip, vp]),
    "boa_finalize_labels_planes": (i32, [vp, vp, vp, i32, ip, vp, i32, i32, i32, vp, i32, vp, ip, ip, vp, i32, i32]),
    "boa_net_predict_sliding_window_deferred": (i32, [vp, vp, ip, ip, ip, ip, i32, vp, vp, vp, ip, C.POINTER(vp)]),
    "boa_net_apply_deferred": (i32, [vp, vp, vp, vp, vp, ip]),
    "boa_stash_destroy": (None, [vp]),
    "boa_net_create": (i32, [vp, C.POINTER(NetDesc), vp, u64, i32, i32, C.POINTER(vp)]),
    "boa_net_destroy": (None, [vp]),
    "boa_net_weight_count": (u64, [C.POINTER(NetDesc)]),
    "boa_net_load_weights": (i32, [vp, vp, u64]),
    "boa_net_set_mirroring": (i32, [vp, i32]),
    "boa_net_forward": (i32, [vp, vp, ip, ip, i32, vp]),
    "boa_net_predict_sliding_window": (i32, [vp, vp, ip, ip, ip, ip, i32, vp, vp, vp]),
    "boa_net_labels_supported": (i32, [vp, ip, i32]),
    "boa_net_predict_labels_fold": (i32, [vp, vp, ip, ip, ip, ip, i32, vp, vp, i32, i32, vp, i32, vp, ip, ip, vp]),
    "boa_conv_block_test": (i32, [vp, vp, i32, i32, ip, vp, vp, vp, vp, i32, ip, ip, i32, i32, vp]),
    "boa_convtranspose_test": (i32, [vp, vp, i32, i32, ip, vp, vp, i32, ip, vp]),
    "boa_tissue_aggregate": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "boa_slice_label_presence": (i32, [vp, vp, i32, i32, i32, vp]),
    "boa_tissue_projections": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "boa_label_hu_histogram": (i32, [vp, vp, vp, vp, u64, i32, i32, vp]),
    "boa_label_hu_mask": (i32, [vp, vp, vp, vp, i32, i32, i32, u64, vp]),
    "boa_binary_erode": (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
    "boa_ccl26": (i32, [vp, vp, i32, i32, i32, vp, vp, ip]),
    "boa_ccl_filter_largest": (i32, [vp, vp, vp, u64, vp, i32]),
    "boa_ccl_remove_small": (i32, [vp, vp, vp, u64, C.c_uint32, vp]),
    "boa_ccl_list_components": (i32, [vp, vp, u64, i32, vp, vp, ip]),
    "boa_scatter_u32": (i32, [vp, vp, vp, vp, i32]),
    "boa_ccl_fill_unmarked": (i32, [vp, vp, vp, u64, C.c_uint32, vp, i32]),
    "boa_label_select": (i32, [vp, vp, u64, i32, ip, vp]),
    "boa_fill_holes_2d": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "boa_binary_dilate_cross": (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
    "boa_mask_assign": (i32, [vp, vp, u64, i32, i32, vp]),
    "boa_label_overlay": (i32, [vp, vp, u64, vp]),
    "boa_median3_inplane": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "boa_bits_words": (u64, [i32, i32, i32]),
    "boa_bits_erode_u8": (i32, [vp, vp, vp, i32, i32, i32, i32, i32]),
    "boa_bits_select": (i32, [vp, vp, i32, i32, i32, vp, i32, vp]),
    "boa_bits_unpack": (i32, [vp, vp, i32, i32, i32, vp]),
    "boa_bits_fill_supported": (i32, [i32, i32]),
    "boa_bits_fill_holes_2d": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "boa_bits_remove_small": (i32, [vp, vp, i32, i32, i32, i32, C.c_uint32, i32]),
    "boa_bits_filter_largest": (i32, [vp, vp, i32, i32, i32, vp, i32]),
    "boa_bits_assign_labels": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "boa_copy3": (i32, [vp, vp, i32, C.c_longlong, C.POINTER(C.c_longlong), ip, vp, i32, C.c_longlong, C.POINTER(C.c_longlong)]),
    "boa_nonzero_bbox": (i32, [vp, vp, i32, ip, ip]),
    "boa_resample_cubic": (i32, [vp, vp, i32, ip, vp, i32, ip]),
    "boa_resample_nearest_u8": (i32, [vp, vp, ip, vp, ip]),
    "boa_resize_skimage_f32": (i32, [vp, vp, ip, vp, ip, i32, i32]),
    "boa_resize_logits_argmax": (i32, [vp, vp, i32, ip, ip, ip, ip, i32, vp, i32, vp]),
    "boa_comm_available": (i32, []),
    "boa_comm_library": (C.c_char_p, []),
    "boa_comm_unique_id": (i32, [vp]),
    "boa_comm_create": (i32, [vp, i32, i32, vp, C.POINTER(vp)]),
    "boa_comm_destroy": (None, [vp]),
    "boa_comm_wait": (i32, [vp]),
    "boa_comm_exchange": (i32, [vp, i32, C.POINTER(vp), C.POINTER(u64), i32, i32, C.POINTER(vp), C.POINTER(u64), i32]),
    "boa_comm_shift_slab": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, ip, vp]),
    "boa_comm_all_reduce": (i32, [vp, vp, u64, i32]),
    "boa_comm_planes_to_owner": (i32, [vp, vp, i32, ip, i32, ip, ip, ip, i32, ip, ip, ip]),
    "boa_comm_stats": (i32, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "boa_add_f16_planes": (i32, [vp, vp, vp, vp, i32, ip, i32, i32]),
}

EXPORTS = sorted(_PROTOS)


def lib():
    """Load libboa_hip.so (once).  torch is imported first so that both bind the same HIP runtime
    (same SONAME libamdhip64.so.7); without torch the system ROCm runtime is used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BoaError(
            f"{LIB_PATH} not found: build it with `make -C body-and-organ-analysis_amd` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (shares libamdhip64 with us)
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(L, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc == BOA_OK:
        return
    msg = lib().boa_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}" if what else msg
    if rc == BOA_EINVAL:
        raise ValueError(text)
    if rc == BOA_ENOMEM:
        raise MemoryError(text)
    raise BoaError(text)


def int3(v):
    return (C.c_int * 3)(*[int(x) for x in v])
