"""HipPredictor: the device counterpart of nnUNetPredictor's inference core.

Same method names and argument meaning as NN/inference/predict_from_raw_data.py for the seams the hot path
uses:
  predict_sliding_window_return_logits(input_image[C,X,Y,Z] fp32) -> fp16 [heads,X,Y,Z]     (:634-680)
  predict_logits_from_preprocessed_data(data)  (fold loop + fp16 mean)                       (:471-504)
plus the fused fast path `predict_segmentation` (normalise + fold mean + argmax + part remap on device, only the
uint8 label volume returns) replacing export_prediction.convert_predicted_logits_to_segmentation_with_correct_shape
(:14-71) for identity resampling.
Errors: RuntimeError on inf logits (:622-625), ValueError/AssertionError on bad shapes, MemoryError when the
accumulators do not fit (the reference would retry with host-side results, :663-672; with 288 GB HBM the
14.5 GB worst case fits, so no host fallback exists here).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import sliding_window as sw
from ._lib import BOA_EINVAL, check, int3
from .device import Context, DeviceBuffer
from .plans import NetGeometry


class HipPredictor:
    _PRECISIONS = {"fp16": 0, "fp32_ref": 1, "fp32": 2}

    def __init__(self, ctx: Context, geometry: NetGeometry, tile_step_size: float = 0.5, use_gaussian: bool = True,
                 use_mirroring: bool = False, max_batch: int = 4, verbose: bool = False, precision: Optional[str] = None,
                 allowed_mirroring_axes: Optional[Sequence[int]] = None):
        # BOA runs every model with tta=False / *NoMirroring trainers (TS/python_api.py:753); use_mirroring=True is honoured
        # as nnUNetPredictor does (predict_from_raw_data.py:541-557) with the checkpoint's `inference_allowed_mirroring_axes`:
        # None (the *NoMirroring trainers, and the default here) means no mirroring even with use_mirroring=True.  The mirrored
        # mean is taken in fp32 (the reference adds in the network's autocast dtype: fp16 on CUDA, fp32 on its CPU path)
        self.use_mirroring = bool(use_mirroring)
        self.allowed_mirroring_axes = None if allowed_mirroring_axes is None else tuple(int(a) for a in allowed_mirroring_axes)
        if self.use_mirroring and self.allowed_mirroring_axes is not None and \
                any(a < 0 or a > 2 for a in self.allowed_mirroring_axes):
            raise AssertionError("mirror_axes does not match the dimension of the input!")
        self.ctx = ctx
        self.lib = ctx.lib
        self.geom = geometry
        self.tile_step_size = float(tile_step_size)
        self.use_gaussian = use_gaussian
        self.verbose = verbose
        self.max_batch = int(max_batch)
        # "fp16" (default): fp16 weights / activations on the f16 matrix cores, fp32 accumulation = what the reference's CUDA
        # path computes under autocast (predict_from_raw_data.py:648); "fp32": what its CPU path computes -- fp32 weights,
        # activations and accumulation -- evaluated in split precision on the matrix cores (every fp32 operand as two fp16 parts,
        # csrc/net_x3.hip): the mode that reproduces the CPU labels; "fp32_ref": the same arithmetic on plain fp32 MFMAs straight
        # from global memory (csrc/net_f32.hip: the slow cross-check of "fp32", and its fallback for kernel shapes the split-precision
        # conv does not instantiate).  $BOA_NET_PRECISION selects it for code that does not pass the argument (the compute/ drop-in
        # surface).
        import os
        precision = precision or os.environ.get("BOA_NET_PRECISION", "fp16")
        if precision not in self._PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self._PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self.list_of_parameters: List[np.ndarray] = []
        self._desc = geometry.to_desc()
        self._net = None
        self._gauss_dev: Optional[DeviceBuffer] = None
        self._loaded_fold = None
        # label-only path: gather form of the tile loop (boa_net_predict_labels_fold) when the network / tile layout qualifies;
        # False keeps the scatter loop (one fused head launch per tile into fp16 accumulator planes)
        self.use_gather_head = True
        ctx.register(self)

    # ---- model management ---------------------------------------------------------------------------
    def set_parameters(self, weight_blobs: Sequence[np.ndarray]):
        """One fp32 blob per fold (plans.weight_blob_from_state_dict); mirrors `self.list_of_parameters`."""
        need = self.lib.boa_net_weight_count(C.byref(self._desc))
        if need == 0:
            raise ValueError("invalid network geometry")
        self.list_of_parameters = []
        for b in weight_blobs:
            b = np.ascontiguousarray(b, dtype=np.float32)
            if b.size != need:
                raise ValueError(f"weight blob has {b.size} floats, geometry needs {need}")
            self.list_of_parameters.append(b)
        self._loaded_fold = None

    def _ensure_net(self, fold: int):
        w = self.list_of_parameters[fold]
        if self._net is None:
            h = C.c_void_p()
            rc = self.lib.boa_net_create(self.ctx.h, C.byref(self._desc), w.ctypes.data_as(C.c_void_p), w.size,
                                         self.max_batch, self._PRECISIONS[self.precision], C.byref(h))
            if rc == BOA_EINVAL and self.precision == "fp32":
                # a kernel shape / channel count the split-precision kernels do not cover (BOA_EINVAL): the plain fp32 kernels take
                # any.  Out-of-memory and HIP errors are NOT retried in a slower mode with a larger footprint: check() raises them.
                import warnings
                warnings.warn("split-precision fp32 mode unavailable for this network (" + self.lib.boa_last_error().decode("utf-8", "replace") + "); using fp32_ref")
                self.precision = "fp32_ref"
                rc = self.lib.boa_net_create(self.ctx.h, C.byref(self._desc), w.ctypes.data_as(C.c_void_p), w.size,
                                             self.max_batch, 1, C.byref(h))
            check(rc, "boa_net_create")
            self._net = h
            self._loaded_fold = fold
            if self.use_mirroring and self.allowed_mirroring_axes:
                mask = 0
                for a in self.allowed_mirroring_axes:
                    mask |= 1 << a
                check(self.lib.boa_net_set_mirroring(self._net, mask), "boa_net_set_mirroring")
        elif self._loaded_fold != fold:
            check(self.lib.boa_net_load_weights(self._net, w.ctypes.data_as(C.c_void_p), w.size), "boa_net_load_weights")
            self._loaded_fold = fold

    def close(self):
        if self._net is not None:
            if self.ctx.h is not None:
                self.lib.boa_net_destroy(self._net)
            self._net = None
        self._gauss_dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _gaussian(self) -> Optional[DeviceBuffer]:
        if not self.use_gaussian:
            return None
        if self._gauss_dev is None:
            g = sw.compute_gaussian(tuple(self.geom.patch_size), sigma_scale=1.0 / 8, value_scaling_factor=10)
            self._gauss_dev = self.ctx.from_numpy(np.ascontiguousarray(g).view(np.uint16))
        return self._gauss_dev

    # ---- tile-level seam ----------------------------------------------------------------------------
    def network_forward(self, volume: np.ndarray, origins: np.ndarray) -> np.ndarray:
        """`self.network(x)` for tiles cut from `volume` [Cin,X,Y,Z] at `origins`: fp32 [n,heads,*patch]."""
        self._ensure_net(0 if self._loaded_fold is None else self._loaded_fold)
        vol = np.ascontiguousarray(volume, dtype=np.float32)
        assert vol.ndim == 4 and vol.shape[0] == self.geom.in_channels
        origins = np.ascontiguousarray(origins, dtype=np.int32).reshape(-1, 3)
        n = origins.shape[0]
        P = self.geom.patch_size
        dvol = self.ctx.from_numpy(vol)
        dout = self.ctx.alloc(n * self.geom.num_classes * int(np.prod(P)) * 4)
        check(self.lib.boa_net_forward(self._net, dvol.vp, int3(vol.shape[1:]), origins.ctypes.data_as(C.POINTER(C.c_int)),
                                       n, dout.vp), "boa_net_forward")
        out = dout.download((n, self.geom.num_classes, *P), np.float32)
        dvol.free()
        dout.free()
        return out

    # ---- sliding window on a resident volume --------------------------------------------------------
    def _run_fold(self, dvol: DeviceBuffer, V, PV, below, origins, acc: DeviceBuffer, nacc: DeviceBuffer, fold: int, gather: bool = False):
        self._ensure_net(fold)
        acc.zero()
        nacc.zero()
        g = self._gaussian()
        org = np.ascontiguousarray(origins, dtype=np.int32).reshape(-1, 3)
        if gather and self.use_gather_head and self.lib.boa_net_labels_supported(self._net, org.ctypes.data_as(C.POINTER(C.c_int)), len(org)):
            # the accumulator planes from ONE gather-head launch (raw partial sums: the tile-sharded path's kernel with nothing
            # deferred) instead of one read-modify-write per tile; the library falls back to the tile loop when the network / tile
            # grid do not qualify.  Same head arithmetic as the label-only path and the sharded path for every tile origin.
            defer = np.zeros(len(org), dtype=np.int32)
            st = C.c_void_p()
            check(self.lib.boa_net_predict_sliding_window_deferred(
                self._net, dvol.vp, int3(V), int3(PV), int3(below), org.ctypes.data_as(C.POINTER(C.c_int)), len(org),
                g.vp if g else None, acc.vp, nacc.vp, defer.ctypes.data_as(C.POINTER(C.c_int)), C.byref(st)),
                "boa_net_predict_sliding_window_deferred")
            self.lib.boa_stash_destroy(st)
            return
        check(self.lib.boa_net_predict_sliding_window(
            self._net, dvol.vp, int3(V), int3(PV), int3(below), origins.ctypes.data_as(C.POINTER(C.c_int)),
            origins.shape[0], g.vp if g else None, acc.vp, nacc.vp), "boa_net_predict_sliding_window")

    def _setup(self, input_image: np.ndarray):
        assert isinstance(input_image, np.ndarray) and input_image.ndim == 4, \
            "input_image must be a 4D np.ndarray (c, x, y, z)"
        if input_image.shape[0] != self.geom.in_channels:
            raise ValueError(f"expected {self.geom.in_channels} channels, got {input_image.shape[0]}")
        V = list(input_image.shape[1:])
        PV, below = sw.pad_amounts(V, self.geom.patch_size)
        origins = sw.get_sliding_window_origins(PV, self.geom.patch_size, self.tile_step_size)
        return V, PV, below, origins

    def predict_sliding_window_return_logits(self, input_image: np.ndarray, fold: int = 0) -> np.ndarray:
        V, PV, below, origins = self._setup(input_image)
        C_ = self.geom.num_classes
        nvox = int(np.prod(PV))
        dvol = self.ctx.from_numpy(np.ascontiguousarray(input_image, dtype=np.float32))
        acc = self.ctx.alloc(C_ * nvox * 2)
        nacc = self.ctx.alloc(nvox * 2)
        flag = self.ctx.zeros(4)
        try:
            self._run_fold(dvol, V, PV, below, origins, acc, nacc, fold)
            check(self.lib.boa_finalize_labels(self.ctx.h, acc.vp, nacc.vp, C_, int3(PV), None, 0, 0, 1, None, 0, None,
                                               None, None, flag.vp), "boa_finalize_labels")
            if int(flag.download((1,), np.int32)[0]):
                raise RuntimeError("Encountered inf in predicted array. Aborting... If this problem persists, reduce "
                                   "value_scaling_factor in compute_gaussian or increase the dtype of predicted_logits "
                                   "to fp32")
            logits = acc.download((C_, *PV), np.uint16).view(np.float16)
        finally:
            for b in (dvol, acc, nacc, flag):
                b.free()
        sl = (slice(None),) + tuple(slice(b, b + v) for b, v in zip(below, V))
        return np.ascontiguousarray(logits[sl])

    def predict_logits_from_preprocessed_data(self, data: np.ndarray) -> np.ndarray:
        """Fold loop with the reference's fp16 `+=` and `/= n_folds` (:483-500), on the host for the logits
        seam; `predict_segmentation` does the same on device."""
        pred = None
        for f in range(len(self.list_of_parameters)):
            lg = self.predict_sliding_window_return_logits(data, fold=f)
            if pred is None:
                pred = lg
            else:
                pred = (pred.astype(np.float32) + lg.astype(np.float32)).astype(np.float16)
        if len(self.list_of_parameters) > 1:
            pred = (pred.astype(np.float32) / np.float32(len(self.list_of_parameters))).astype(np.float16)
        return pred

    def predict_segmentation_device(self, dvol: DeviceBuffer, V, labels_out: DeviceBuffer, lut: Optional[np.ndarray] = None,
                                    merge: bool = False, work: Optional[dict] = None, shard=None, resample_to=None):
        """All folds -> labels on device.  dvol: fp32 [Cin,*V] resident; labels_out: uint8 [*V] resident (updated in
        place when merge=True).  `work` may carry preallocated acc / n / fold buffers to reuse across models.
        `shard` (tile_shard.TileShard): this volume is shared by the ranks of shard.comm -- every rank holds the volume,
        runs its block of tile rows, exchanges the overlap slabs and ends with the same labels_out.
        `resample_to` = (out_dims, slice_axis): the volume is at the plans' spacing and the labels are wanted on another
        grid (nnU-Net's own resampling, export_prediction.py:25-33): the fold-mean logits are resampled with order 1 to
        `out_dims` and reduced to labels there (`boa_resize_logits_argmax`); labels_out is then uint8 [*out_dims]."""
        if shard is not None and shard.comm.world > 1:
            return self._predict_segmentation_sharded(dvol, V, labels_out, lut, merge, work, shard, resample_to)
        PV, below = sw.pad_amounts(V, self.geom.patch_size)
        origins = sw.get_sliding_window_origins(PV, self.geom.patch_size, self.tile_step_size)
        C_ = self.geom.num_classes
        nvox = int(np.prod(PV))
        nf = len(self.list_of_parameters)
        own = work is None
        work = work if work is not None else {}

        def buf(name, nbytes):
            b = work.get(name)
            if b is None or b.nbytes < nbytes:
                if b is not None:
                    b.free()
                b = self.ctx.alloc(nbytes)
                work[name] = b
            return b

        ring = work.get("flag_ring")   # (device.FlagRing of the calling task: the inf flag is read once per volume, not per model)
        if ring is not None:
            flag = ring.next()
        else:
            flag = buf("flag", 4)
            flag.zero()
        lut_arr = None
        if lut is not None:
            lut_arr = np.zeros(256, dtype=np.uint8)
            lut_arr[:len(lut)] = lut
        lut_p = lut_arr.ctypes.data_as(C.c_void_p) if lut_arr is not None else None
        crop = any(b != 0 for b in below) or list(PV) != list(V)
        direct = resample_to is None
        if direct and self.use_gather_head and self._predict_labels_fused(dvol, V, PV, below, origins, labels_out, lut_p, merge, crop, flag, buf):
            if ring is None and int(flag.download((1,), np.int32)[0]):
                raise RuntimeError("Encountered inf in predicted array. Aborting...")
            if own:
                for b in work.values():
                    b.free()
            return
        acc = buf("acc", C_ * nvox * 2)
        nacc = buf("n", nvox * 2)
        fold = buf("fold", C_ * nvox * 2) if nf > 1 else None
        for f in range(nf):
            self._run_fold(dvol, V, PV, below, origins, acc, nacc, f, gather=True)
            last = f == nf - 1
            # resampled path: keep the normalised (fold-mean) logits -- in `acc` for one fold, in `fold` otherwise -- and
            # take the argmax after the resampling
            check(self.lib.boa_finalize_labels(
                self.ctx.h, acc.vp, nacc.vp, C_, int3(PV), fold.vp if fold else None, 0 if f == 0 else 1,
                nf if (last and fold) else 0, 0 if (direct or fold) else 1, lut_p,
                1 if merge else 0, labels_out.vp if (last and direct) else None, int3(below) if crop else None,
                int3(V) if crop else None, flag.vp), "boa_finalize_labels")
        if ring is None and int(flag.download((1,), np.int32)[0]):
            raise RuntimeError("Encountered inf in predicted array. Aborting...")
        if not direct:
            out_dims, slice_axis = resample_to
            check(self.lib.boa_resize_logits_argmax(self.ctx.h, (fold if fold else acc).vp, C_, int3(PV), int3(below), int3(V),
                                                    int3(out_dims), int(slice_axis), lut_p, 1 if merge else 0, labels_out.vp),
                  "boa_resize_logits_argmax")
        if own:
            for b in work.values():
                b.free()

    def _predict_labels_fused(self, dvol, V, PV, below, origins, labels_out, lut_p, merge, crop, flag, buf) -> bool:
        """The gather form of the tile loop (boa_net_predict_labels_fold): every tile's last decoder activation goes to a stash,
        one pass per fold walks the volume with the fp16 running sums in registers and ends in the labels -- no accumulator
        planes.  Same arithmetic and rounding order as the scatter loop below (labels bit-identical).  False = not applicable
        (exact mode, mirroring, > 32 classes, features[0] != 32, irregular tile list, stash does not fit): the caller runs
        the scatter loop."""
        nf = len(self.list_of_parameters)
        self._ensure_net(0)
        op = origins.ctypes.data_as(C.POINTER(C.c_int))
        if not self.lib.boa_net_labels_supported(self._net, op, origins.shape[0]):
            return False
        C_, nvox = self.geom.num_classes, int(np.prod(PV))
        fold = buf("fold", C_ * nvox * 2) if nf > 1 else None
        g = self._gaussian()
        for f in range(nf):
            self._ensure_net(f)
            rc = self.lib.boa_net_predict_labels_fold(
                self._net, dvol.vp, int3(V), int3(PV), int3(below), op, origins.shape[0], g.vp if g else None,
                fold.vp if fold else None, f, nf, lut_p, 1 if merge else 0, labels_out.vp, int3(below) if crop else None,
                int3(V) if crop else None, flag.vp)
            if rc == -3 and f == 0:      # BOA_ENOMEM: the stash does not fit -> scatter loop
                return False
            check(rc, "boa_net_predict_labels_fold")
        return True

    def _predict_segmentation_sharded(self, dvol, V, labels_out, lut, merge, work, shard, resample_to=None):
        job = self.begin_segmentation_sharded(dvol, V, labels_out, lut, merge, work, shard, resample_to)
        job.finish()

    def begin_segmentation_sharded(self, dvol, V, labels_out, lut, merge, work, shard, resample_to=None, after_first_start=None):
        """Shared-volume prediction in two halves (tile_shard.start_fold_sharded / finish_fold_sharded): this call runs the tiles
        of every fold and queues the slab exchanges; the returned job's `finish()` applies what arrived, reduces the owned planes
        to labels and merges the ranks' label planes.  A multi-model task calls `begin` of model k + 1 before `finish` of model k
        (`after_first_start`: called once this model's first fold is queued -- the previous model's finish), so that with the RCCL
        transport the exchange of model k runs under the tiles of model k + 1.  The accumulators alternate between two slots."""
        job = _ShardedJob(self, dvol, V, labels_out, lut, merge, work, shard, resample_to)
        job.begin(after_first_start)
        return job

    def predict_segmentation(self, input_image: np.ndarray, lut: Optional[np.ndarray] = None) -> np.ndarray:
        V = list(input_image.shape[1:])
        dvol = self.ctx.from_numpy(np.ascontiguousarray(input_image, dtype=np.float32))
        lab = self.ctx.zeros(int(np.prod(V)))
        try:
            self.predict_segmentation_device(dvol, V, lab, lut=lut, merge=False)
            return lab.download(tuple(V), np.uint8)
        finally:
            dvol.free()
            lab.free()


class _ShardedJob:
    """One model (all folds) of a volume that several ranks share; see HipPredictor.begin_segmentation_sharded."""

    def __init__(self, pred: HipPredictor, dvol, V, labels_out, lut, merge, work, shard, resample_to):
        from . import tile_shard as ts
        self.ts, self.p, self.ctx, self.lib = ts, pred, pred.ctx, pred.lib
        self.dvol, self.V, self.labels_out, self.merge, self.shard, self.resample_to = dvol, list(V), labels_out, merge, shard, resample_to
        self.own = work is None
        self.work = work if work is not None else {}
        self.PV, self.below = sw.pad_amounts(self.V, pred.geom.patch_size)
        self.origins = sw.get_sliding_window_origins(self.PV, pred.geom.patch_size, pred.tile_step_size)
        self.plan = ts.plan_rows(self.origins, pred.geom.patch_size[0], self.PV[0], shard.comm.world,
                                 assignment=getattr(shard, "assignment", None))
        self.C_, self.nvox, self.nv = pred.geom.num_classes, int(np.prod(self.PV)), int(np.prod(self.V))
        self.nf = len(pred.list_of_parameters)
        self.lut_arr = None
        if lut is not None:
            self.lut_arr = np.zeros(256, dtype=np.uint8)
            self.lut_arr[:len(lut)] = lut
        self.crop = any(b != 0 for b in self.below) or list(self.PV) != list(self.V)
        self.direct = resample_to is None
        self.pending = None
        self.lo = self.hi = 0
        self.done = False
        # (fold, tile row) units -- a model ensemble on a grid with fewer tile rows than ranks (the BCA nets: 5 folds, 2 rows at 5 mm
        # slices): rows alone keep `plan.active` ranks busy.  The folds are independent until `prediction += fold` (fp16, in fold
        # order, predict_from_raw_data.py:494-500), so the unit list (fold-major) is cut into one contiguous run per rank
        # (tile_shard.plan_units), every fold runs the exact row protocol among ITS ranks, the normalised fp16 logits of a fold --
        # plane-disjoint over its ranks -- are summed over all ranks (x + 0 is exact), and every rank then adds the folds IN ORDER on
        # its share of the planes: the same fp16 operations in the same order as on one GPU -> bit-identical labels.
        import os as _os
        self.fold_plans = None
        if (self.nf > 1 and self.direct and shard.mode == "exact" and getattr(shard, "assignment", None) is None
                and shard.comm.world > self.plan.active and not _os.environ.get("BOA_NO_FOLD_UNITS")):
            units = ts.plan_units([len(self.plan.rows)] * self.nf, shard.comm.world)
            self.fold_plans = [ts.plan_rows(self.origins, pred.geom.patch_size[0], self.PV[0], shard.comm.world, assignment=units[f])
                               for f in range(self.nf)]

    def _buf(self, name, nbytes):
        b = self.work.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self.ctx.alloc(nbytes)
            self.work[name] = b
        return b

    def _lut_p(self):
        return self.lut_arr.ctypes.data_as(C.c_void_p) if self.lut_arr is not None else None

    def _begin_fold_units(self, after_first_start):
        ts, comm, rank = self.ts, self.shard.comm, self.shard.comm.rank
        self.part = self._buf("part", self.nv)
        self.flogits = []
        member = 0
        for f, plan_f in enumerate(self.fold_plans):
            F = self._buf(f"flog{f}", self.C_ * self.nvox * 2)       # fold f's normalised logits, complete on every rank after the sum
            lo = hi = 0
            if plan_f.index(rank) is not None:
                self.p._ensure_net(f)
                nacc = self._buf(f"fn{member}", self.nvox * 2)
                member += 1
                eng = ts.HipShardEngine(self.p, comm, self.dvol, self.V, self.PV, self.below, self.origins, F, nacc)
                try:
                    lo, hi = ts.finish_fold_sharded(ts.start_fold_sharded(eng, plan_f, comm, self.shard.mode))
                finally:
                    eng.close()
                # acc / n -> normalised fp16 logits in place on the planes this rank owns (inf check included)
                check(self.lib.boa_finalize_labels_planes(self.ctx.h, F.vp, nacc.vp, self.C_, int3(self.PV), None, 0, 0, 1, None, 0, None,
                                                          None, None, self.flag.vp, lo, hi), "boa_finalize_labels_planes")
            self.flogits.append((F, lo, hi))
            if f == 0 and after_first_start is not None:
                after_first_start()
        # every rank, every fold, in fold order.  The folds are only ever read on this rank's finalize share (_finish_fold_units), so the
        # plane-disjoint normalised logits go to the rank that finalises them (reduce-scatter as a plane exchange to the owner: half the
        # per-link bytes of the all-reduce; $BOA_FOLD_ALLREDUCE=1 restores the sum that leaves every rank with complete logits)
        import os as _os
        if _os.environ.get("BOA_FOLD_ALLREDUCE"):
            for F, lo, hi in self.flogits:
                ts.all_reduce_logit_planes(self.ctx, comm, F, self.C_, self.PV, lo, hi)
        else:
            shares = ts.plane_shares(self.PV[0], comm.world)
            for (F, _, _), plan_f in zip(self.flogits, self.fold_plans):
                owned = [plan_f.owned_planes(q) for q in range(comm.world)]
                ts.reduce_scatter_logit_planes(self.ctx, comm, F, self.C_, self.PV, owned, shares)

    def _finish_fold_units(self):
        comm = self.shard.comm
        P0, P1 = (self.PV[0] * comm.rank) // comm.world, (self.PV[0] * (comm.rank + 1)) // comm.world
        ones = self.work.get("ones")
        if ones is None or ones.nbytes < self.nvox * 2:
            if ones is not None:
                ones.free()
            ones = self.ctx.from_numpy(np.full(self.nvox, 0x3C00, np.uint16))       # fp16 1.0: x / 1 == x, the fold logits are normalised already
            self.work["ones"] = ones
        fold = self._buf("fold", self.C_ * self.nvox * 2)
        for f, (F, _, _) in enumerate(self.flogits):
            last = f == self.nf - 1
            check(self.lib.boa_finalize_labels_planes(
                self.ctx.h, F.vp, ones.vp, self.C_, int3(self.PV), fold.vp, 0 if f == 0 else 1, self.nf if last else 0, 0, self._lut_p(), 0,
                self.part.vp if last else None, int3(self.below) if self.crop else None, int3(self.V) if self.crop else None, self.flag.vp,
                P0, P1), "boa_finalize_labels_planes")

    def begin(self, after_first_start=None):
        job_slot = self.work.get("_job", 0) & 1
        self.work["_job"] = self.work.get("_job", 0) + 1
        self.flag = self._buf(f"flag{job_slot}", 4)
        self.flag.zero()
        if self.fold_plans is not None:
            return self._begin_fold_units(after_first_start)
        self.fold = self._buf("fold", self.C_ * self.nvox * 2) if self.nf > 1 else None
        self.part = self._buf("part", self.nv)
        for f in range(self.nf):
            self.p._ensure_net(f)
            slot = self.work.get("_seq", 0) & 1            # two accumulator sets: the one whose exchange is in flight and the one being filled
            self.work["_seq"] = self.work.get("_seq", 0) + 1
            acc, nacc = self._buf(f"acc{slot}", self.C_ * self.nvox * 2), self._buf(f"n{slot}", self.nvox * 2)
            eng = self.ts.HipShardEngine(self.p, self.shard.comm, self.dvol, self.V, self.PV, self.below, self.origins, acc, nacc)
            state = self.ts.start_fold_sharded(eng, self.plan, self.shard.comm, self.shard.mode)
            if f == 0 and after_first_start is not None:
                after_first_start()
            if self.pending is not None:
                self._finish_fold(*self.pending)
            self.pending = (f, eng, state)

    def _finish_fold(self, f, eng, state):
        try:
            self.lo, self.hi = self.ts.finish_fold_sharded(state)
        finally:
            eng.close()
        last = f == self.nf - 1
        fold, direct = self.fold, self.direct
        # (resampled path: keep the normalised fold-mean logits of the owned planes -- in the accumulators for one fold, in `fold` otherwise)
        check(self.lib.boa_finalize_labels_planes(
            self.ctx.h, eng.acc.vp, eng.nacc.vp, self.C_, int3(self.PV), fold.vp if fold else None, 0 if f == 0 else 1,
            self.nf if (last and fold) else 0, 0 if (direct or fold) else 1, self._lut_p(), 0, self.part.vp if (last and direct) else None,
            int3(self.below) if self.crop else None, int3(self.V) if self.crop else None, self.flag.vp, self.lo, self.hi),
            "boa_finalize_labels_planes")
        self.logits = fold if fold else eng.acc

    def finish(self):
        if self.done:
            return
        self.done = True
        ts, comm, nv = self.ts, self.shard.comm, self.nv
        check(self.lib.boa_memset(self.ctx.h, self.part.vp, 0, nv), "boa_memset")
        if self.fold_plans is not None:
            self._finish_fold_units()
        if self.pending is not None:
            self._finish_fold(*self.pending)
            self.pending = None
        if self.direct:
            ts.all_reduce_labels(self.ctx, comm, self.part, nv)
        else:
            # nnU-Net resamples the logits (order 1) before the argmax: every rank needs all planes -> sum the plane-disjoint logits
            # over the ranks, then the same fused resize + argmax as on one GPU (identical labels on every rank)
            out_dims, slice_axis = self.resample_to
            ts.all_reduce_logit_planes(self.ctx, comm, self.logits, self.C_, self.PV, self.lo, self.hi)
        if ts.all_reduce_flag(self.ctx, comm, self.flag):
            raise RuntimeError("Encountered inf in predicted array. Aborting...")
        if not self.direct:
            check(self.lib.boa_resize_logits_argmax(self.ctx.h, self.logits.vp, self.C_, int3(self.PV), int3(self.below), int3(self.V), int3(out_dims),
                                                    int(slice_axis), self._lut_p(), 1 if self.merge else 0, self.labels_out.vp),
                  "boa_resize_logits_argmax")
        elif self.merge:
            check(self.lib.boa_label_overlay(self.ctx.h, self.part.vp, nv, self.labels_out.vp), "boa_label_overlay")
        else:
            one, st = (C.c_int * 3)(1, 1, nv), (C.c_longlong * 3)(0, 0, 1)
            check(self.lib.boa_copy3(self.ctx.h, self.part.vp, 0, 0, st, one, self.labels_out.vp, 0, 0, st), "boa_copy3")
        if self.own:
            for b in self.work.values():
                if hasattr(b, "free"):
                    b.free()

