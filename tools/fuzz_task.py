#!/usr/bin/env python3
"""Randomised differential test of the task driver (SegmentationTask.predict_image) against the oracle pipeline: random file
orientations (all 48 axis permutations x flips), voxel spacings (with / without resampling to the model spacing), volume
extents around the patch size, zero slabs (crop_to_nonzero), single / multi-model tasks, forced z-split.  Every remap around
the networks is integer-exact, so the label agreement bar only leaves room for fp16 near-tie flips (>= 97 %)."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip import orientation as o  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.task import SegmentationTask  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402
import test_gpu_tasks as T  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
perms = list(itertools.permutations(range(3)))
bad = 0
for i in range(n_cases):
    resample = float(rng.choice([1.5, 3.0]))
    same = rng.random() < 0.4
    sp = (resample,) * 3 if same else tuple(float(v) for v in np.round(rng.uniform(0.7, 3.2, size=3), 2))
    shape = tuple(int(v) for v in rng.integers(20, 70, size=3))
    ct_ras = T._ct(shape, 50 + i)
    if rng.random() < 0.5:
        ct_ras[: int(rng.integers(1, 6))] = 0
        ct_ras[:, :, -int(rng.integers(1, 5)):] = 0
    n_models = int(rng.choice([1, 2]))
    models, omodels, luts = [], [], {}
    for k in range(n_models):
        nc = int(rng.integers(3, 8))
        m, om = T._model(700 + k, nc, 700 + 10 * i + k, (resample,) * 3)
        models.append(m)
        lut = np.concatenate([[0], np.arange(1, nc) + 20 * k]).astype(np.uint8)
        luts[700 + k] = lut
        omodels.append(om + ({int(j): f"c{int(v)}" for j, v in enumerate(lut) if j},))
    class_inv = {f"c{int(v)}": int(v) for lut in luts.values() for v in lut[1:]}
    multimodel = n_models > 1
    force_split = bool(rng.random() < 0.3)
    if force_split:   # the reference splits only long volumes (z > 200); its margins need z >= ~65 after resampling
        shape = (shape[0], shape[1], int(np.ceil(rng.integers(70, 100) * resample / sp[2])))
        ct_ras = T._ct(shape, 150 + i)
    # random orientation of the file
    perm = perms[int(rng.integers(0, 6))]
    flips = rng.integers(0, 2, size=3)
    aff_ras = np.diag([sp[0], sp[1], sp[2], 1.0])
    aff_ras[:3, 3] = rng.uniform(-100, 100, size=3)
    ornt_target = np.array([[perm[a], -1 if flips[a] else 1] for a in range(3)], dtype=float)   # RAS axis a -> file axis
    ct_file = np.ascontiguousarray(o.apply_orientation(ct_ras, ornt_target))
    aff = aff_ras @ o.inv_ornt_aff(ornt_target, ct_ras.shape)
    try:
        np.testing.assert_array_equal(o.apply_orientation(ct_file, o.io_orientation(aff)), ct_ras)
        if multimodel:
            want_ras = opipe.predict_image(ct_ras, sp, omodels, class_inv, "total", resample, force_split=force_split)
        else:
            want_ras = opipe.predict_image(ct_ras, sp, [omodels[0][:4] + (None,)], None, "other", resample, multimodel=False,
                                           force_split=force_split)
        t = SegmentationTask(ctx, "total" if multimodel else "other", models, resample=resample, multimodel=multimodel, max_batch=4,
                             part_luts=luts if multimodel else None)
        got = t.predict_image(ct_file, aff, force_split=force_split)
        t.close()
        want = o.apply_orientation(want_ras, ornt_target)
        agree = float((got == want).mean()) if got.shape == want.shape else 0.0
        ok = agree >= 0.97
        print(f"{'ok ' if ok else 'BAD'} case {i}: shape={shape} sp={sp} resample={resample} axcodes={o.aff2axcodes(aff)} models={n_models} "
              f"split={force_split}: agreement {agree:.4f} labels {len(np.unique(got))}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        print(f"BAD case {i}: shape={shape} sp={sp} perm={perm} flips={flips}: {type(e).__name__}: {e}", flush=True)
        bad += 1
print(f"{n_cases} cases, {bad} failures")
ctx.close()
sys.exit(1 if bad else 0)
