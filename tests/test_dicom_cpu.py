"""DICOM series -> image.nii.gz + ct_info (BOA/compute/io.py:254-383).  No reference fixture and no pydicom /
SimpleITK here: the series are written by tests/dicom_writer.py (an independent encoder of PS3.5) -- parity unpinned."""
import datetime
import os

import numpy as np
import pytest

from boa_hip import dicom, nifti
from boa_hip.compute.io import get_image_info

from dicom_writer import write_series, write_slice


def _volume(n=12, rows=9, cols=7, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 4000, size=(n, rows, cols)).astype(np.uint16)


@pytest.mark.parametrize("explicit", [True, False])
def test_series_to_nifti_roundtrip(tmp_path, explicit):
    vol = _volume()
    # written in reverse file order: the IPP sort must restore the stack
    write_series(tmp_path / "in", vol, order=list(range(len(vol)))[::-1], explicit=explicit, bits_stored=12)
    path, info = get_image_info(tmp_path / "in", tmp_path / "out")
    assert path == tmp_path / "out" / "image.nii.gz"
    data, affine, hdr = nifti.load(path)
    assert data.dtype == np.int16                      # 12 bits unsigned, intercept -1024 -> [-1024, 3071]
    assert data.shape == (7, 9, 12)                    # (columns, rows, slices) = ITK's x, y, z
    np.testing.assert_array_equal(data, vol.astype(np.int64).transpose(2, 1, 0) - 1024)
    # LPS origin (-100, -120, 50), spacing x = column spacing 0.7, y = row spacing 0.8, z = 1.5 -> RAS affine
    want = np.array([[-0.7, 0, 0, 100.0], [0, -0.8, 0, 120.0], [0, 0, 1.5, 50.0], [0, 0, 0, 1]])
    np.testing.assert_allclose(affine, want, atol=1e-6)
    assert (hdr.qform_code, hdr.sform_code) == (1, 1)
    np.testing.assert_allclose(nifti._qform(hdr), want, atol=1e-5)     # the q-form encodes the same matrix
    np.testing.assert_allclose(hdr.get_zooms(), (0.7, 0.8, 1.5), rtol=1e-6)
    d = {e["name"]: e["value"] for e in info}
    assert [e["name"] for e in info] == ["StudyInstanceUID", "SeriesInstanceUID", "Date", "AgeYears", "Gender", "AccessionNumber",
                                         "SeriesNumber", "SeriesDescription", "Modality", "CTDIvol", "ExposureTime", "XRayTubeCurrent",
                                         "Exposure", "KVP", "SpiralPitchFactor", "ConvolutionKernel", "SliceThickness", "PixelSpacingX",
                                         "PixelSpacingY", "ScanLength"]
    assert d["Date"] == "17.03.2024" and d["AgeYears"] == 63 and d["Gender"] == "F" and d["Modality"] == "CT"
    assert d["ConvolutionKernel"] == "B31f" and d["KVP"] == 120.0 and d["XRayTubeCurrent"] == 220 and d["CTDIvol"] == 7.25
    assert d["PixelSpacingX"] == 0.8 and d["PixelSpacingY"] == 0.7 and d["ScanLength"] is None and d["SeriesNumber"] == 4


def test_age_before_birthday():
    info = dicom.ct_info_from_dataset({"SeriesDate": "20240317", "PatientBirthDate": "19600318"})
    assert {e["name"]: e["value"] for e in info}["AgeYears"] == 63
    info = dicom.ct_info_from_dataset({"SeriesDate": "20240317", "PatientBirthDate": "1960031"})   # malformed -> None, as _safe_da
    assert {e["name"]: e["value"] for e in info}["AgeYears"] is None
    assert datetime.date(2024, 3, 17)


def test_tilted_within_tolerance_keeps_direction(tmp_path):
    vol = _volume(seed=1)
    t = np.deg2rad(10.0)
    iop = (1, 0, 0, 0, np.cos(t), np.sin(t))            # gantry tilt about x: normal = (0, -sin t, cos t)
    write_series(tmp_path / "in", vol, iop=iop)
    path, _ = get_image_info(tmp_path / "in", tmp_path / "out")
    data, affine, _ = nifti.load(path)
    normal = np.array([0, -np.sin(t), np.cos(t)])
    col = np.array([0, np.cos(t), np.sin(t)])
    lps = np.stack([np.array([1.0, 0, 0]) * 0.7, col * 0.8, normal * 1.5], axis=1)
    np.testing.assert_allclose(affine[:3, :3], np.diag([-1, -1, 1]) @ lps, atol=1e-6)
    # shuffled file order, same series -> same volume
    write_series(tmp_path / "in2", vol, iop=iop, order=[5, 0, 11, 3, 8, 1, 10, 2, 9, 4, 7, 6])
    p2, _ = get_image_info(tmp_path / "in2", tmp_path / "out2")
    np.testing.assert_array_equal(nifti.load(p2)[0], data)


def test_refusals(tmp_path):
    vol = _volume()
    write_series(tmp_path / "few", vol[:5])
    with pytest.raises(ValueError, match="less than 10 instances: 5"):
        get_image_info(tmp_path / "few", tmp_path / "o")
    write_series(tmp_path / "mr", vol, modality="MR")
    with pytest.raises(ValueError, match="The modality is not CT: MR"):
        get_image_info(tmp_path / "mr", tmp_path / "o")
    write_series(tmp_path / "cor", vol, iop=(1, 0, 0, 0, 0, -1))
    with pytest.raises(ValueError, match="Image plane is coronal, not axial"):
        get_image_info(tmp_path / "cor", tmp_path / "o")
    t = np.deg2rad(40.0)
    write_series(tmp_path / "obl", vol, iop=(1, 0, 0, 0, np.cos(t), np.sin(t)))
    with pytest.raises(ValueError, match="tilted beyond tolerance"):
        get_image_info(tmp_path / "obl", tmp_path / "o")
    write_series(tmp_path / "loc", vol, image_type=("DERIVED", "SECONDARY", "REFORMATTED"))
    with pytest.raises(ValueError, match="disqualifying marker"):
        get_image_info(tmp_path / "loc", tmp_path / "o")
    write_series(tmp_path / "gap", vol, skip=(6,))
    with pytest.raises(ValueError, match="non-uniform slice distances"):
        get_image_info(tmp_path / "gap", tmp_path / "o")
    write_series(tmp_path / "jpeg", vol, transfer_syntax="1.2.840.10008.1.2.4.70")
    with pytest.raises(NotImplementedError, match="transfer syntax"):
        get_image_info(tmp_path / "jpeg", tmp_path / "o")
    os.makedirs(tmp_path / "empty")
    (tmp_path / "empty" / "notes.txt").write_text("not dicom")
    with pytest.raises(ValueError, match="no DICOM series"):
        get_image_info(tmp_path / "empty", tmp_path / "o")


def test_first_series_only_and_dtype_choice(tmp_path):
    vol = _volume(seed=2)
    write_series(tmp_path / "in", vol, series_uid="1.2.3.20", name="B%04d.dcm", bits_stored=16)
    other = _volume(seed=3)
    write_series(tmp_path / "in", other, series_uid="1.2.3.100", name="A%04d.dcm", bits_stored=16)   # sorts behind "1.2.3.100" < "1.2.3.20"
    files = dicom.series_file_names(tmp_path / "in")
    assert len(files) == 12 and all(os.path.basename(f).startswith("A") for f in files)             # first series UID in sorted order
    data, geom, _ = dicom.load_series(tmp_path / "in")
    assert data.dtype == np.int32                       # 16 bits unsigned with intercept -1024 does not fit int16 (GDCM's rule)
    np.testing.assert_array_equal(data, other.astype(np.int64).transpose(2, 1, 0) - 1024)
    # non-integer slope -> float64
    write_series(tmp_path / "f", vol, slope=0.5, intercept=-10.25, bits_stored=12)
    data, _, _ = dicom.load_series(tmp_path / "f")
    assert data.dtype == np.float64
    np.testing.assert_allclose(data, vol.transpose(2, 1, 0) * 0.5 - 10.25)
    # signed stored values, slope 1 / intercept 0
    sv = (vol.astype(np.int32) - 2000).astype(np.int16)
    write_series(tmp_path / "s", sv, signed=True, intercept=0, bits_stored=16)
    data, _, _ = dicom.load_series(tmp_path / "s")
    assert data.dtype == np.int16
    np.testing.assert_array_equal(data, sv.transpose(2, 1, 0))


def test_reader_skips_unknown_and_nested_elements(tmp_path):
    px = np.arange(12, dtype=np.uint16).reshape(3, 4)
    for explicit in (True, False):
        p = tmp_path / f"x{int(explicit)}.dcm"
        write_slice(p, px, ipp=(1.5, 2.5, 3.5), explicit=explicit)
        ds = dicom.read_file(p)
        assert ds["Rows"] == 3 and ds["Columns"] == 4 and ds["ImagePositionPatient"] == [1.5, 2.5, 3.5]
        assert ds["ImageType"] == ["ORIGINAL", "PRIMARY", "AXIAL"] and ds["PixelData"] == px.tobytes()
        assert "PixelData" not in dicom.read_file(p, stop_before_pixels=True)
    with open(tmp_path / "bad.dcm", "wb") as f:
        f.write(b"\0" * 200)
    with pytest.raises(dicom.DicomError):
        dicom.read_file(tmp_path / "bad.dcm")


def test_more_refusals_and_geometry_checks(tmp_path):
    vol = _volume()
    # explicit VR big endian
    write_series(tmp_path / "be", vol, transfer_syntax="1.2.840.10008.1.2.2")
    with pytest.raises(NotImplementedError, match="transfer syntax"):
        dicom.load_series(tmp_path / "be")
    # multi-frame object
    write_series(tmp_path / "mf", vol, extra=[((0x0028, 0x0008), "IS", 4)])
    with pytest.raises(NotImplementedError, match="multi-frame"):
        dicom.load_series(tmp_path / "mf")
    # a slice without ImagePositionPatient cannot be ordered
    write_series(tmp_path / "noipp", vol)
    write_slice(tmp_path / "noipp" / "extra.dcm", vol[0], ipp=None, instance=99)
    with pytest.raises(ValueError, match="ImagePositionPatient"):
        dicom.series_file_names(tmp_path / "noipp")
    # a duplicated instance (same position twice)
    write_series(tmp_path / "dup", vol)
    write_slice(tmp_path / "dup" / "again.dcm", vol[3], ipp=(-100.0, -120.0, 50.0 + 3 * 1.5), instance=77)
    with pytest.raises(ValueError, match="share a position"):
        dicom.load_series(tmp_path / "dup")
    # a sheared stack: slice origins drift in-plane
    import os
    os.makedirs(tmp_path / "shear")
    for z in range(len(vol)):
        write_slice(tmp_path / "shear" / f"s{z:03d}.dcm", vol[z], ipp=(-100.0 + 0.4 * z, -120.0, 50.0 + 1.5 * z), instance=z + 1)
    with pytest.raises(ValueError, match="sheared"):
        dicom.load_series(tmp_path / "shear")
    # slices of different size
    write_series(tmp_path / "size", vol)
    write_slice(tmp_path / "size" / "big.dcm", np.zeros((10, 7), np.uint16), ipp=(-100.0, -120.0, 50.0 + 12 * 1.5), instance=13)
    with pytest.raises(ValueError, match="slice size differs"):
        dicom.load_series(tmp_path / "size")
    # a single slice: z spacing falls back to SliceThickness
    write_series(tmp_path / "one", vol[:1])
    data, geom, files = dicom.load_series(tmp_path / "one")
    assert data.shape == (7, 9, 1) and geom["spacing"][2] == 1.5 and len(files) == 1
