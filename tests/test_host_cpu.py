"""CPU: host logic of the product package + the C ABI library loads and exports what include/boa_hip.h declares."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_library_exports_every_declared_symbol():
    from boa_hip import _lib
    hdr = open(os.path.join(ROOT, "include", "boa_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(boa_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    assert os.path.exists(_lib.LIB_PATH), "build libboa_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in boa_hip.h but not exported"
    # and every symbol is bound with a prototype on the Python side
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.lib().boa_version() >= 100


def test_tile_starts_match_reference():
    from boa_hip import sliding_window as sw
    for c in _json("g1_steps.json"):
        assert sw.compute_steps_for_sliding_window(c["size"], c["patch"], c["step"]) == c["steps"]
    o = sw.get_sliding_window_origins((512, 512, 512), (128, 128, 128), 0.8)
    assert o.shape == (125, 3) and o[1].tolist() == [0, 0, 96] and o[-1].tolist() == [384, 384, 384]


def test_gaussian_matches_reference_bits():
    from boa_hip import sliding_window as sw
    z = np.load(os.path.join(GOLDEN, "g2_gaussian.npz"))
    for k in z.files:
        ts = tuple(int(v) for v in k[2:].split("x"))
        np.testing.assert_array_equal(sw.compute_gaussian(ts, 1. / 8, 10).view(np.uint16), z[k])
    for k, m in _json("g2_gaussian_big.json").items():
        ts = tuple(int(v) for v in k.split("x"))
        bits = np.ascontiguousarray(sw.compute_gaussian(ts, 1. / 8, 10).view(np.uint16))
        assert hashlib.sha256(bits.tobytes()).hexdigest() == m["sha256"]


def test_pad_amounts():
    from boa_hip import sliding_window as sw
    assert sw.pad_amounts((12, 40, 20), (16, 16, 16)) == ([16, 40, 20], [2, 0, 0])
    assert sw.pad_amounts((13, 40, 9), (16, 16, 16)) == ([16, 40, 16], [1, 0, 3])


def test_plans_roundtrip_and_weight_blob():
    from boa_hip import _lib, plans
    pj, dj = plans.synthetic_plans()
    cfg = plans.model_config_from_plans(pj, dj)
    g = cfg.geometry
    assert g.features == [32, 64, 128, 256, 320, 320] and g.num_classes == 25 and g.patch_size == [128, 128, 128]
    # 478.8 GMAC, 2.074 GB
    assert abs(g.flops_per_tile() / 2 / 1e9 - 478.8) < 0.5
    assert abs(g.activation_bytes_per_tile() / 1e6 - 2073.7) < 2.0
    sd = plans.synthetic_state_dict(g, 0)
    blob = plans.weight_blob_from_state_dict(g, sd)
    d = g.to_desc()
    assert _lib.lib().boa_net_weight_count(ctypes.byref(d)) == blob.size
    assert abs(blob.size - 31.19e6) < 0.3e6  # 31.19 M parameters
    # old-format plans are reconstructed like plans_handler.py:36-97
    old = {"configurations": {"3d_fullres": {
        "UNet_class_name": "PlainConvUNet", "UNet_base_num_features": 32, "unet_max_num_features": 320,
        "n_conv_per_stage_encoder": [2] * 6, "n_conv_per_stage_decoder": [2] * 5, "num_pool_per_axis": [5, 5, 5],
        "pool_op_kernel_sizes": [[1, 1, 1]] + [[2, 2, 2]] * 5, "conv_kernel_sizes": [[3, 3, 3]] * 6,
        "patch_size": [128, 128, 128], "spacing": [1.5, 1.5, 1.5], "normalization_schemes": ["CTNormalization"]}},
        "foreground_intensity_properties_per_channel": {"0": {"mean": 1.0, "std": 2.0, "percentile_00_5": 0, "percentile_99_5": 3}}}
    g2 = plans.model_config_from_plans(old, dj).geometry
    assert g2.features == g.features and g2.strides == g.strides and g2.kernels == g.kernels
    import pytest
    bad = dict(sd)
    bad.pop("decoder.transpconvs.0.bias")
    with pytest.raises(KeyError):
        plans.weight_blob_from_state_dict(g, bad)


def test_oracle_network_accepts_upstream_key_names():
    import torch
    from boa_hip import plans
    from oracle.network import build_from_arch
    pj, dj = plans.synthetic_plans(patch=(16, 16, 16), features=(32, 64), num_classes=3)
    g = plans.model_config_from_plans(pj, dj).geometry
    sd = plans.synthetic_state_dict(g, 1)
    net = build_from_arch(pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"], 1, 3)
    res = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    y = net(torch.zeros(1, 1, 16, 16, 16))
    assert tuple(y.shape) == (1, 3, 16, 16, 16)


def test_every_product_module_imports():
    import importlib
    for m in ("boa_hip", "boa_hip._lib", "boa_hip.device", "boa_hip.sliding_window", "boa_hip.plans", "boa_hip.predictor",
              "boa_hip.synthetic", "boa_hip.label_maps", "boa_hip.bca", "boa_hip.measurements", "boa_hip.compute.config",
              "boa_hip.compute.constants", "boa_hip.compute.util", "boa_hip.totalseg", "boa_hip.task", "boa_hip.pipeline",
              "boa_hip.resample", "boa_hip.orientation", "boa_hip.distributed", "boa_hip.devarray", "boa_hip.nifti",
              "boa_hip.model_store", "boa_hip.compute.inference", "boa_hip.compute.measurements", "boa_hip.report",
              "boa_hip.agg_shard", "boa_hip.nnunet_resample", "boa_hip.tile_shard"):
        importlib.import_module(m)


def test_stats_from_hist_matches_numpy():
    """Order statistics / mean from an integer histogram are bit-identical to numpy on the raw values."""
    from boa_hip import measurements as M
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 101, 1000, 4097):
        x = rng.integers(-1100, 1300, size=n).astype(np.int16)
        h = np.bincount(x.astype(np.int64) - M.HU_MIN, minlength=M.NBINS)
        st = M.stats_from_hist(h)
        assert st["n"] == n
        assert st["mean"] == float(np.mean(x)) and st["min"] == float(np.min(x)) and st["max"] == float(np.max(x))
        assert st["median"] == float(np.median(x))
        assert st["p25"] == float(np.percentile(x, 25)) and st["p75"] == float(np.percentile(x, 75))
        assert np.isclose(st["std"], float(np.std(x)), rtol=1e-12, atol=1e-12)
    assert M.stats_from_hist(np.zeros(M.NBINS, np.int64)) is None


def test_orientation_restatement():
    """nibabel.orientations restated (boa_hip/orientation.py): hand-checked affines, world-coordinate consistency of the
    reoriented affine, undo_canonical round trip for every axis permutation x flip."""
    import itertools
    from boa_hip import orientation as o
    assert o.aff2axcodes(np.diag([1.5, 1.5, 1.5, 1])) == ("R", "A", "S")
    assert o.aff2axcodes(np.diag([-0.8, -0.8, 3.0, 1])) == ("L", "P", "S")
    a = np.arange(2 * 3 * 4).reshape(2, 3, 4)
    for perm in itertools.permutations(range(3)):
        for flips in itertools.product([1, -1], repeat=3):
            aff = np.zeros((4, 4))
            for in_ax, (out_ax, f) in enumerate(zip(perm, flips)):
                aff[out_ax, in_ax] = f * (0.7 + 0.4 * in_ax)
            aff[:3, 3] = [11.0, -7.0, 3.0]
            aff[3, 3] = 1
            # slightly oblique: the closest axis still wins
            aff[:3, :3] += 0.01
            c, caff, ornt = o.as_closest_canonical(a, aff)
            assert o.aff2axcodes(caff) == ("R", "A", "S")
            assert [int(v) for v in ornt[:, 0]] == list(perm)
            for idx in [(0, 0, 0), (1, 2, 3), (1, 0, 2)]:
                pos = np.argwhere(c == a[idx])[0]
                assert np.allclose(caff @ np.append(pos, 1.0), aff @ np.append(idx, 1.0))
            assert np.array_equal(o.undo_canonical(c, aff), a)
            l, laff = o.with_axcodes(a, aff, "LPS")
            assert o.aff2axcodes(laff) == ("L", "P", "S")
            pos = np.argwhere(l == a[1, 2, 3])[0]
            assert np.allclose(laff @ np.append(pos, 1.0), aff @ np.array([1, 2, 3, 1.0]))


def test_task_bookkeeping():
    from boa_hip.task import get_bbox_from_mask, nonzero_bbox, split_bounds
    parts, comb = split_bounds(1024)       # SURVEY 8b / TS/nnunet.py:496-505,583-586
    assert parts == [(0, 361), (322, 702), (663, 1024)]
    full = np.zeros(1024, int)
    for (lo, hi), (dst, src) in zip(parts, comb):
        full[dst] = np.arange(lo, hi)[src]
    assert np.array_equal(full, np.arange(1024))      # every slice comes from the part where it is >= 19 slices from a cut
    for nz in (201, 230, 599, 768):
        parts, comb = split_bounds(nz)
        full = np.full(nz, -1)
        for (lo, hi), (dst, src) in zip(parts, comb):
            full[dst] = np.arange(lo, hi)[src]
        assert np.array_equal(full, np.arange(nz))
    m = np.zeros((10, 12, 14))
    m[3:6, 4:5, 7:12] = 1
    assert get_bbox_from_mask(m, 0, [2, 2, 3]) == [[1, 8], [2, 7], [4, 14]]
    assert get_bbox_from_mask(np.zeros((3, 4, 5)), 0, 1) == [[0, 3], [0, 4], [0, 5]]
    assert nonzero_bbox(m) == [[3, 6], [4, 5], [7, 12]]


def test_nifti_roundtrip_and_label_extension(tmp_path):
    from boa_hip import nifti
    rng = np.random.default_rng(0)
    ct = rng.integers(-1024, 3000, size=(7, 9, 5)).astype(np.int16)
    aff = np.array([[-0.8, 0, 0, 100], [0, -0.8, 0.01, 50], [0, 0, 3.0, -20], [0, 0, 0, 1.0]])
    for name in ("a.nii", "a.nii.gz"):
        nifti.save(tmp_path / name, ct, aff)
        x, a, h = nifti.load(tmp_path / name)
        assert np.array_equal(x, ct) and np.allclose(a, aff, atol=1e-6)
        assert h.get_data_shape() == (7, 9, 5) and h.get_data_dtype() == np.int16
        assert np.allclose(h.get_zooms(), (0.8, 0.8, 3.0), atol=1e-4)
    lab = (ct > 0).astype(np.uint8)
    nifti.save(tmp_path / "l.nii.gz", lab, a, like=h, extensions=[(0, nifti.label_xml({1: "spleen", 2: "kidney_right"}))])
    y, a2, h2 = nifti.load(tmp_path / "l.nii.gz")
    assert np.array_equal(y, lab) and np.allclose(a2, aff, atol=1e-6) and h2.get_data_dtype() == np.uint8
    assert nifti.parse_label_xml(h2.extensions[0][1]) == {1: "spleen", 2: "kidney_right"}
    assert int(h2.vox_offset) % 16 == 0


def test_nifti_reader_on_foreign_files(tmp_path):
    """nifti.load on files this repo did NOT write: the reference's own test volumes (NN/tests/example_data, committed as data
    fixtures tests/golden/ref_*.nii.gz) and byte-patched variants (qform only with qfac -1, no form code, scl_slope / scl_inter,
    big-endian) against golden G15 -- an independent field-by-field parse by tests/golden/make_golden.py.  What the reference gets
    from nibabel for them: NN/imageio/nibabel_reader_writer.py:38-99 (shape, affine, zooms, dtype, get_fdata)."""
    import gzip
    import hashlib
    import json
    import sys
    from boa_hip import nifti
    sys.path.insert(0, GOLDEN)
    exp_all = json.load(open(os.path.join(GOLDEN, "g15_nifti.json")))

    def check(path, exp):
        data, aff, h = nifti.load(path)
        assert list(data.shape) == exp["shape"] and list(h.get_data_shape()) == exp["shape"]
        assert h.datatype == exp["datatype"] and data.dtype.itemsize * 8 == exp["bitpix"]
        assert data.dtype.isnative
        np.testing.assert_array_equal(np.asarray(h.get_zooms(), dtype=np.float64), np.asarray(exp["zooms"]))
        assert (h.qform_code, h.sform_code) == (exp["qform_code"], exp["sform_code"])
        np.testing.assert_allclose(aff, np.asarray(exp["affine"], dtype=np.float64), rtol=0, atol=1e-12)
        assert hashlib.sha256(np.ascontiguousarray(data).tobytes()).hexdigest() == exp["voxels_sha256"]
        assert int(data.astype(np.int64).sum()) == exp["voxel_sum"]
        assert float(data.min()) == exp["voxel_min"] and float(data.max()) == exp["voxel_max"]
        f = nifti.fdata(data, h)
        assert f.dtype == np.float64 and float(f.sum()) == exp["fdata_sum"]
        for k, v in exp["fdata_at"].items():
            assert float(f[tuple(int(i) for i in k.split(","))]) == v
        assert nifti.is_scaled(h) == (exp["scl_slope"] not in (0.0, 1.0) or exp["scl_inter"] != 0.0)

    for name, cases in exp_all.items():
        check(os.path.join(GOLDEN, "ref_" + name), cases["orig"])
    # the variants: same byte patches as the generator (offsets of the NIfTI-1 header table)
    import struct
    name = "example_ct_sm.nii.gz"
    raw = gzip.open(os.path.join(GOLDEN, "ref_" + name), "rb").read()
    patched = {}
    b = bytearray(raw)
    b[252:256] = struct.pack("<hh", 1, 0)
    b[76:80] = struct.pack("<f", -1.0)
    b[256:280] = struct.pack("<6f", 0.0, 0.0, 0.70710678, 10.5, -20.25, 7.0)
    patched["qform_only"] = bytes(b)
    b = bytearray(raw)
    b[252:256] = struct.pack("<hh", 0, 0)
    patched["no_form"] = bytes(b)
    b = bytearray(raw)
    b[112:120] = struct.pack("<ff", 2.0, -1024.0)
    patched["scaled"] = bytes(b)
    hdr_fmt = "i10s18sihcB8h3f4h8f3fhBB4f2i80s24s2h3f3f12f16s4s"
    v = struct.unpack("<" + hdr_fmt, raw[:348])
    n, off = v[8] * v[9] * v[10], int(v[30])
    patched["big_endian"] = struct.pack(">" + hdr_fmt, *v) + raw[348:off] + np.frombuffer(raw, "<i2", n, off).astype(">i2").tobytes()
    for k, blob in patched.items():
        pth = tmp_path / (k + (".nii.gz" if k != "scaled" else ".nii"))
        with (gzip.open(pth, "wb") if str(pth).endswith(".gz") else open(pth, "wb")) as fh:
            fh.write(blob)
        check(pth, exp_all[name][k])
    # and the drop-in reader keeps the CT what the reference's pipeline sees: int16 HU with the identity scaling
    data, aff, h = nifti.load(os.path.join(GOLDEN, "ref_" + name))
    assert data.dtype == np.int16 and not nifti.is_scaled(h)


def test_model_store_folder_contract(tmp_path):
    from boa_hip import model_store, plans
    pj, dj = plans.synthetic_plans(patch=(16, 16, 16), features=(8, 16), num_classes=3)
    geom = plans.model_config_from_plans(pj, dj).geometry
    sds = [plans.synthetic_state_dict(geom, seed=s) for s in range(5)]
    model_store.write_model_folder(str(tmp_path), 542, "BCA_inference", "nnUNetTrainerNoMirroring", pj, dj, sds)
    with pytest.raises(RuntimeError):
        model_store.find_dataset_dir(543, str(tmp_path))
    (tid, cfg, blobs), = model_store.load_task_models("body_regions", root=str(tmp_path))
    assert tid == 542 and len(blobs) == 5 and cfg.geometry.num_classes == 3
    np.testing.assert_array_equal(blobs[3], plans.weight_blob_from_state_dict(geom, sds[3]))
    assert len(model_store.load_task_models("body_regions", fast_bca=True, root=str(tmp_path))[0][2]) == 1


def test_measurement_label_maps_match_reference_tables():
    """boa_hip/data/measurement_label_maps.json (product data) == G11 (derived by the reference's own code)."""
    from boa_hip import label_maps
    with open(os.path.join(GOLDEN, "g11_measurement_label_maps.json")) as f:
        g = json.load(f)
    for model, pairs in g["label_maps"].items():
        lm = label_maps.measurement_label_map(model)
        assert list(lm.items()) == [(k, v) for k, v in pairs], model          # content and order
    assert len(label_maps.measurement_label_map("total")) == 295              # total + total_v1 + total_mr quirk
    for model, cm in g["class_maps"].items():
        assert label_maps.class_map(model) == {int(k): v for k, v in cm.items()}
    assert label_maps.class_map("total") == label_maps.CLASS_MAP_TOTAL
    assert label_maps.output_name("lung_vessels") == "lung_vessels_airways" and label_maps.output_name("total") == "total"
    assert label_maps.cnr_adjusted_regions() == {k: set(v) for k, v in g["cnr_adjusted_regions"].items()}


def test_cnr_with_constant_autochthon_follows_numpy_semantics():
    """BOA/compute/measurements.py:117-119 divides numpy scalars: std 0 of the autochthon gives inf / nan, no exception."""
    from boa_hip import measurements as M
    st = {"n": 10, "mean": 50.0, "std": 2.0, "min": 40.0, "median": 50.0, "max": 60.0, "p25": 45.0, "p75": 55.0}
    assert M._metrics(st, 0.003375, 30.0, 4.0)["cnr"] == 5.0
    assert np.isposinf(M._metrics(st, 0.003375, 30.0, 0.0)["cnr"])
    assert np.isnan(M._metrics(st, 0.003375, 50.0, 0.0)["cnr"])
    assert M._metrics(st, 0.003375, None, None)["cnr"] is None


def test_descriptive_statistics_equal_pandas_describe():
    """bca._descriptive evaluates `DataFrame.describe()` (builder.py:263-307) with the numpy reductions pandas dispatches to;
    it must give the very same doubles (the reference calls pandas)."""
    import pandas as pd
    from boa_hip import bca
    rng = np.random.default_rng(0)
    for _ in range(12):
        Z = int(rng.integers(2, 300))
        counts = rng.integers(0, 50000, size=(Z, 8)).astype(np.uint32)
        counts[:, 0] = 0
        sums = rng.integers(-10 ** 6, 10 ** 6, size=(Z, 8)).astype(np.int64)
        df = bca._slicewise(counts, float(rng.uniform(0.3, 5)) ** 3 / 1000)
        lo = int(rng.integers(0, Z - 1))
        hi = int(rng.integers(lo + 1, Z + 1))
        m = bca._descriptive(df[bca.COLS].to_numpy(dtype=np.float64), bca.COLS, counts, sums, lo, hi)
        sw = df[(df.slice_idx >= lo) & (df.slice_idx < hi)].drop("slice_idx", axis=1)
        ref = sw.describe()
        ref.drop("count", inplace=True)
        ref.index = ["Mean", "StdDev", "Minimum", "25%", "Median", "75%", "Maximum"]
        ref.loc["Total"] = sw.sum()
        # the reference's own tail (builder.py:284-307 + run_pipeline's rename / to_dict): MeanHU row, NaN -> None, dict
        ref = ref.astype(object)
        c = counts[lo:hi].astype(np.int64).sum(axis=0)
        sm = sums[lo:hi].sum(axis=0)
        for nme, v in bca.TISSUES:
            ref.loc["MeanHU", bca._tname(nme)] = (float(sm[v]) / float(c[v])) if c[v] else None
        adip = [5, 3, 4, 6, 7]
        ref.loc["MeanHU", "TAT"] = (float(sm[adip].sum()) / float(c[adip].sum())) if c[adip].sum() else None
        ref = ref.where(ref.notna(), None)
        want = ref.rename(index=bca._ROW, columns={k: k.lower() for k in ref.columns}).to_dict()
        assert list(m) == list(want)
        for col in want:
            assert list(m[col]) == list(want[col]), (list(m[col]), list(want[col]))
            for k, y in want[col].items():
                x = m[col][k]
                assert (x is None and y is None) or (type(x) is float and x == float(y)), (col, k, x, y)


def test_nifti_parallel_gzip_members(tmp_path):
    """`.nii.gz` written as independently deflated gzip members (nr_thr_saving cores): standard readers see one stream, the
    bytes do not depend on the thread count, load() round-trips."""
    import gzip
    import subprocess
    from boa_hip import nifti
    rng = np.random.default_rng(5)
    a = (rng.integers(0, 118, (160, 150, 230)) * (rng.random((160, 150, 230)) < 0.25)).astype(np.uint8)     # 5.5 MB: 2 members
    aff = np.array([[-1.5, 0, 0, 10], [0, -1.5, 0, 20], [0, 0, 3.0, -5], [0, 0, 0, 1]], dtype=np.float64)
    p1, p4 = tmp_path / "t1.nii.gz", tmp_path / "t4.nii.gz"
    nifti.save(p1, a, aff, threads=1, extensions=[(0, nifti.label_xml({1: "one", 2: "two"}))])
    nifti.save(p4, a, aff, threads=4, extensions=[(0, nifti.label_xml({1: "one", 2: "two"}))])
    assert p1.read_bytes() == p4.read_bytes()
    raw = gzip.open(p4, "rb").read()                                   # Python's reader: all members, one stream
    assert len(raw) == int(np.frombuffer(raw[108:112], "<f4")[0]) + a.size
    out = subprocess.run(["gunzip", "-c", str(p4)], capture_output=True)
    if out.returncode == 0:
        assert out.stdout == raw
    b, baff, hdr = nifti.load(p4)
    assert np.array_equal(a, b) and np.allclose(aff, baff)
    assert nifti.parse_label_xml(hdr.extensions[0][1]) == {1: "one", 2: "two"}


def test_nifti_indexed_gzip_members_parallel_read(tmp_path):
    """nifti.save writes gzip members that carry their own length in a FEXTRA subfield; nifti.load inflates them in parallel.
    The file stays an ordinary gzip stream (python's gzip module and `gzip -t` read it), a foreign single-stream file takes the
    sequential path, and a damaged member is detected by its CRC."""
    import gzip
    import subprocess
    from boa_hip import nifti
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 118, size=(96, 100, 110)).astype(np.uint8)            # ~1 MB raw: several members at block 256 KiB
    path = tmp_path / "v.nii.gz"
    with open(path, "wb") as f:
        payload = vol.tobytes(order="F")
        nifti.write_gzip_members(f, payload, 1, threads=4, block=256 << 10)
    raw = open(path, "rb").read()
    tab = nifti._member_table(raw)
    assert tab is not None and len(tab) == -(-len(payload) // (256 << 10)) and sum(t[2] for t in tab) == len(payload)
    assert gzip.decompress(raw) == payload                                       # one standard stream
    assert subprocess.run(["gzip", "-t", str(path)]).returncode == 0
    for th in (1, 2, 8):
        assert bytes(nifti.read_bytes(path, threads=th)) == payload
    # through save / load, with a header + extension in front of the data
    aff = np.diag([1.5, 1.5, 3.0, 1.0])
    nifti.save(tmp_path / "w.nii.gz", vol, aff, extensions=[(0, nifti.label_xml({1: "a"}))], threads=3)
    for th in (1, 4):
        got, a2, h = nifti.load(tmp_path / "w.nii.gz", threads=th)
        np.testing.assert_array_equal(got, vol)
        np.testing.assert_allclose(a2, aff)
        assert nifti.parse_label_xml(h.extensions[0][1]) == {1: "a"}
    # a foreign writer's single stream
    with open(tmp_path / "f.nii.gz", "wb") as f:
        f.write(gzip.compress(gzip.decompress(open(tmp_path / "w.nii.gz", "rb").read()), 6))
    assert nifti._member_table(open(tmp_path / "f.nii.gz", "rb").read()) is None
    np.testing.assert_array_equal(nifti.load(tmp_path / "f.nii.gz")[0], vol)
    many = nifti.load_many([tmp_path / "w.nii.gz", tmp_path / "f.nii.gz", tmp_path / "w.nii.gz"], threads=4)
    assert all(np.array_equal(m[0], vol) for m in many)
    # corruption inside the second member's deflate data
    bad = bytearray(raw)
    bad[tab[1][0] + 40] ^= 0x5A
    open(tmp_path / "bad.gz", "wb").write(bytes(bad))
    with pytest.raises(Exception):
        nifti.read_bytes(tmp_path / "bad.gz", threads=4)


def _upstream_checkpoint_keys(sd, n_stages):
    """The key set a real dynamic_network_architectures PlainConvUNet checkpoint carries on top of the canonical names: the
    nn.Sequential aliases of every conv block, the decoder's reference to the encoder, the deep-supervision heads."""
    out = dict(sd)
    for k, v in sd.items():
        if ".convs." in k and (".conv." in k or ".norm." in k):
            out[k.replace(".conv.", ".all_modules.0.").replace(".norm.", ".all_modules.1.")] = v
    for k, v in list(out.items()):
        if k.startswith("encoder."):
            out["decoder." + k] = v
    last = n_stages - 2
    w, b = sd[f"decoder.seg_layers.{last}.weight"], sd[f"decoder.seg_layers.{last}.bias"]
    for lvl in range(last):                                   # coarser deep-supervision heads (other input widths upstream; unused)
        out[f"decoder.seg_layers.{lvl}.weight"] = np.zeros((w.shape[0], 7, 1, 1, 1), np.float32)
        out[f"decoder.seg_layers.{lvl}.bias"] = np.zeros_like(b)
    return {("_orig_mod." + k): v for k, v in out.items()}    # torch.compile prefix (predict_from_raw_data.py:95-98 strips it)


def test_loader_accepts_upstream_alias_keys_and_rejects_unknown_ones():
    from boa_hip import plans
    pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64, 128), num_classes=5)
    geom = plans.model_config_from_plans(pj, dj).geometry
    sd = plans.synthetic_state_dict(geom, 3)
    want = plans.weight_blob_from_state_dict(geom, sd)
    full = _upstream_checkpoint_keys(sd, geom.n_stages)
    assert len(full) > 2 * len(sd)
    np.testing.assert_array_equal(plans.weight_blob_from_state_dict(geom, full), want)
    bad = dict(full)
    bad["_orig_mod.encoder.stages.0.0.convs.0.all_modules.0.weight"] = bad["_orig_mod.encoder.stages.0.0.convs.0.all_modules.0.weight"] + 1
    with pytest.raises(ValueError, match="alias"):
        plans.weight_blob_from_state_dict(geom, bad)
    for extra in ("encoder.stages.0.0.convs.0.norm.running_mean", "decoder.stages.0.convs.5.conv.weight", "some.other.module.weight"):
        bad = dict(sd)
        bad[extra] = np.zeros(3, np.float32)
        with pytest.raises(ValueError, match="unexpected checkpoint key"):
            plans.weight_blob_from_state_dict(geom, bad)
    missing = {k: v for k, v in sd.items() if k != "decoder.transpconvs.0.bias"}
    with pytest.raises(KeyError):
        plans.weight_blob_from_state_dict(geom, missing)


def test_legacy_plans_format_gives_the_same_geometry():
    """plans.json of the older nnU-Net format (UNet_class_name / conv_kernel_sizes / pool_op_kernel_sizes / modality /
    foreground_intensity_properties_by_modality: plans_handler.py:36-97 reconstructs the architecture from them) -> the same
    ModelConfig as the new `architecture.arch_kwargs` form."""
    from boa_hip import plans
    pj, dj = plans.synthetic_plans(patch=(32, 48, 40), features=(32, 64, 128, 256), num_classes=7, spacing=(3.0, 1.5, 1.5))
    new = plans.model_config_from_plans(pj, dj)
    lj, ldj = plans.legacy_plans_from(pj, dj)
    assert "architecture" not in lj["configurations"]["3d_fullres"] and "channel_names" not in ldj
    old = plans.model_config_from_plans(lj, ldj)
    assert old.geometry == new.geometry and old.spacing == new.spacing and old.intensity_properties == new.intensity_properties
