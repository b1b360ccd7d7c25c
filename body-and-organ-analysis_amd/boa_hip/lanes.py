"""One CT through `total` -> total measurements -> bca on resident data, on one or on two streams of the same GPU.

The two halves of `--models total+bca` only meet at the vertebra table: the BCA nets (body_parts, body_regions; 5 folds each
at 5 mm slices) read the CT alone, `total`'s five part models read the CT alone, and `create_vertebrae_info`
(BCA/cli.py:110-150 via BOA/compute/inference.py:118-141) is the first consumer of both.  The reference runs them one after
the other (compute_all_models :102-141); so does `lanes=1` here.  With two lanes the `total` half (+ its measurements) is
driven by a worker thread on the task's Context and the BCA half by the calling thread on a SECOND Context (own stream, own
buffer pool, own predictors): the kernels of one stream fill the launch tails and the thin deep layers of the other
(tools/two_streams.py measured the effect on two volumes).  Every kernel of either half is launched exactly as in the
one-stream run -- same tiles, same batches, same buffers' contents -- so labels and tables are bit-identical
(tests/test_gpu_lanes.py).

Hand-over between the streams is by value: the CT is copied into the second Context's pool before the BCA nets start and
the finished `total` labels after lane A synchronised; no kernel of one stream ever reads a buffer the other stream is still
writing.
"""
from __future__ import annotations

import threading
from typing import Optional

import numpy as np

from . import measurements as M
from .devarray import DevArray
from .pipeline import BcaPipelineHip
from .task import SegmentationTask


class TotalBcaRunner:
    """`total_task` and `pipe` on the same Context: one stream (the reference's order).  `pipe` built on another Context of
    the same GPU: two lanes.  `pipe=None`: `total` + measurements only."""

    def __init__(self, total_task: SegmentationTask, pipe: Optional[BcaPipelineHip], label_map, cnr_adjustment: bool = True):
        self.total_task, self.pipe = total_task, pipe
        self.ctx = total_task.ctx
        self.label_map, self.cnr_adjustment = label_map, cnr_adjustment
        self.two_lanes = pipe is not None and pipe.ctx is not self.ctx
        if self.two_lanes and pipe.ctx.device != self.ctx.device:
            raise ValueError("TotalBcaRunner: both Contexts must be on the same GPU")

    # ---- the `total` half -------------------------------------------------------------------------------------------
    def _total(self, d_ct: DevArray, affine):
        return self.total_task.predict_image(d_ct, affine, return_device=True)

    def _total_measurements(self, d_ct: DevArray, d_total: DevArray, spacing, defer: bool):
        """compute_measurements' view: SimpleITK (z,y,x) arrays of the file (BOA/compute/measurements.py:257-258).  The device
        passes run now; defer=True returns `finish()` for the per-label order statistics (numpy on the downloaded histogram)."""
        c_zyx = d_ct.transpose((2, 1, 0)).contiguous(np.int16, force_copy=True)
        s_zyx = d_total.transpose((2, 1, 0)).contiguous(force_copy=True)
        try:
            fin, d_mask = M.total_measurements(self.ctx, None, None, self.label_map, spacing, cnr_adjustment=self.cnr_adjustment,
                                               d_ct=c_zyx.buf, d_lab=s_zyx.buf, shape=c_zyx.shape, mask_on_device=True,
                                               defer_host=defer)
            d_mask.free()
        finally:
            c_zyx.free()
            s_zyx.free()
        return fin

    # ---- one volume --------------------------------------------------------------------------------------------------
    def run_resident(self, d_ct: DevArray, affine, spacing=(1.5, 1.5, 1.5), on_stage=None) -> dict:
        """`d_ct`: the file's int16 array (x,y,z) resident on the `total` Context (not freed).  -> {"total": DevArray,
        "total_measurements": dict, and with a pipe "body_parts" / "body_regions" / "tissues": DevArrays (file axis order,
        the caller frees all four), "bca_measurements", "vertebrae", "examined_body_part"}.  `on_stage(name)`: called at the
        stage boundaries of the one-stream run (bench.py's stage clock)."""
        affine = np.asarray(affine, dtype=np.float64)
        if not self.two_lanes:
            return self._run_one_stream(d_ct, affine, spacing, on_stage or (lambda name: None))
        return self._run_two_lanes(d_ct, affine, spacing)

    def _run_one_stream(self, d_ct, affine, spacing, on_stage) -> dict:
        d_total = self._total(d_ct, affine)
        out = {"total": d_total}
        try:
            on_stage("total")
            fin = self._total_measurements(d_ct, d_total, spacing, defer=self.pipe is not None)
            on_stage("total measurements")
            if self.pipe is None:
                out["total_measurements"] = fin
                return out
            # the order statistics on a worker thread under the BCA nets' kernels, joined before the volume is handed back
            box = {}

            def stats():
                try:
                    box["meas"] = fin()
                except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread after the join)
                    box["error"] = e

            th = threading.Thread(target=stats, name="boa-total-measurements")
            th.start()
            try:
                out.update(self.pipe.run_resident(d_ct, affine, d_total))
            finally:
                th.join()
            if "error" in box:
                raise box["error"]
            out["total_measurements"] = box["meas"]
            on_stage("bca")
            return out
        except BaseException:
            for k in ("total", "body_parts", "body_regions", "tissues"):
                if isinstance(out.get(k), DevArray):
                    out[k].free()
            raise

    def _run_two_lanes(self, d_ct, affine, spacing) -> dict:
        ctx_a, ctx_b = self.ctx, self.pipe.ctx
        ctx_a.sync()                                  # the CT is complete before the other stream copies it
        ct_b = d_ct.to_context(ctx_b)
        box: dict = {}
        total_ready = threading.Event()

        def lane_a():
            try:
                ctx_a.bind_thread()
                box["total"] = self._total(d_ct, affine)
                ctx_a.sync()                          # labels complete before lane B copies them
                total_ready.set()
                box["meas"] = self._total_measurements(d_ct, box["total"], spacing, defer=False)
            except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread)
                box["error"] = e
            finally:
                total_ready.set()

        copies = []

        def total_for_lane_b():
            total_ready.wait()
            if "error" in box:
                raise RuntimeError("lane A (`total`) failed") from box["error"]
            copies.append(box["total"].to_context(ctx_b))
            return copies[-1]

        th = threading.Thread(target=lane_a, name="boa-lane-total")
        th.start()
        res = None
        try:
            res = self.pipe.run_resident(ct_b, affine, total_for_lane_b)
        finally:
            th.join()
            ctx_b.sync()
            ct_b.free()
            for c in copies:
                c.free()
            if "error" in box or res is None:
                if isinstance(box.get("total"), DevArray):
                    box["total"].free()
                if res is not None:
                    for k in ("body_parts", "body_regions", "tissues"):
                        res[k].free()
        if "error" in box:
            raise box["error"]
        out = {"total": box["total"], "total_measurements": box["meas"]}
        out.update(res)
        return out
