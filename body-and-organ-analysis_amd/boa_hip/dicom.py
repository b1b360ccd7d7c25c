"""Minimal DICOM CT series reader: folder of single-frame CT slices -> (volume, LPS geometry) -> `image.nii.gz`
(the reference: BOA/compute/io.py:254-259 `_load_series_from_disk` = SimpleITK `ImageSeriesReader` over
`GetGDCMSeriesFileNames`, :326-338 `get_image_info`).  pydicom, SimpleITK and GDCM are absent from this image and the reference
holds no DICOM fixture: this module follows the published standard (PS3.5 encoding, PS3.3 C.7.6.2 image plane module) and the
documented behaviour of ITK's series reader / NIfTI writer; it is exercised against files written by the test suite's own writer
only -- PARITY UNPINNED vs GDCM / SimpleITK.

Scope (everything else raises, nothing is guessed):
  * transfer syntaxes: implicit VR little endian (1.2.840.10008.1.2) and explicit VR little endian (1.2.840.10008.1.2.1), i.e.
    uncompressed; encapsulated / deflated / big-endian files raise NotImplementedError;
  * single-frame, MONOCHROME2, SamplesPerPixel 1, BitsAllocated 16 (8 and 32 are read too);
  * one series per call: like `GetGDCMSeriesFileNames(dir)` without a series id, the FIRST series (smallest SeriesInstanceUID
    in sorted order) of the folder is taken, other series' files are ignored;
  * slices are ordered by the projection of ImagePositionPatient on the slice normal (the IPP sort GDCM applies), so a folder
    written in reverse or shuffled order yields the same volume;
  * geometry as ITK's ImageSeriesReader builds it: origin, in-plane spacing and direction cosines from the first slice of the
    sorted series, slice spacing = |IPP(last) - IPP(first)| / (n - 1), third direction = row x column;
  * departure from SimpleITK, on purpose: a series with a missing slice / non-uniform slice distances (> 1 % of the mean, or
    10 um) and slices whose orientation, size or spacing differ RAISE ValueError -- ITK only warns ("Non uniform sampling or
    missing slices detected") and writes a volume with a wrong z-spacing.
"""
from __future__ import annotations

import os
import pathlib
import struct
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

IMPLICIT_LE = "1.2.840.10008.1.2"
EXPLICIT_LE = "1.2.840.10008.1.2.1"
_LONG_VR = {b"OB", b"OW", b"OF", b"OD", b"OL", b"OV", b"SQ", b"UT", b"UN", b"UC", b"UR", b"SV", b"UV"}

# (group, element) -> (keyword, VR): the attributes this path reads (image plane, pixel description, `ct_info`) -- the implicit-VR
# dictionary; any other tag of an implicit-VR file is skipped by its length (undefined length = a sequence)
TAGS: Dict[Tuple[int, int], Tuple[str, str]] = {
    (0x0002, 0x0010): ("TransferSyntaxUID", "UI"),
    (0x0008, 0x0008): ("ImageType", "CS"),
    (0x0008, 0x0016): ("SOPClassUID", "UI"),
    (0x0008, 0x0018): ("SOPInstanceUID", "UI"),
    (0x0008, 0x0021): ("SeriesDate", "DA"),
    (0x0008, 0x0050): ("AccessionNumber", "SH"),
    (0x0008, 0x0060): ("Modality", "CS"),
    (0x0008, 0x103E): ("SeriesDescription", "LO"),
    (0x0010, 0x0030): ("PatientBirthDate", "DA"),
    (0x0010, 0x0040): ("PatientSex", "CS"),
    (0x0018, 0x0050): ("SliceThickness", "DS"),
    (0x0018, 0x0060): ("KVP", "DS"),
    (0x0018, 0x1150): ("ExposureTime", "IS"),
    (0x0018, 0x1151): ("XRayTubeCurrent", "IS"),
    (0x0018, 0x1152): ("Exposure", "IS"),
    (0x0018, 0x1210): ("ConvolutionKernel", "SH"),
    (0x0018, 0x1302): ("ScanLength", "IS"),
    (0x0018, 0x9311): ("SpiralPitchFactor", "FD"),
    (0x0018, 0x9345): ("CTDIvol", "FD"),
    (0x0020, 0x000D): ("StudyInstanceUID", "UI"),
    (0x0020, 0x000E): ("SeriesInstanceUID", "UI"),
    (0x0020, 0x0011): ("SeriesNumber", "IS"),
    (0x0020, 0x0013): ("InstanceNumber", "IS"),
    (0x0020, 0x0032): ("ImagePositionPatient", "DS"),
    (0x0020, 0x0037): ("ImageOrientationPatient", "DS"),
    (0x0028, 0x0002): ("SamplesPerPixel", "US"),
    (0x0028, 0x0004): ("PhotometricInterpretation", "CS"),
    (0x0028, 0x0008): ("NumberOfFrames", "IS"),
    (0x0028, 0x0010): ("Rows", "US"),
    (0x0028, 0x0011): ("Columns", "US"),
    (0x0028, 0x0030): ("PixelSpacing", "DS"),
    (0x0028, 0x0100): ("BitsAllocated", "US"),
    (0x0028, 0x0101): ("BitsStored", "US"),
    (0x0028, 0x0102): ("HighBit", "US"),
    (0x0028, 0x0103): ("PixelRepresentation", "US"),
    (0x0028, 0x1052): ("RescaleIntercept", "DS"),
    (0x0028, 0x1053): ("RescaleSlope", "DS"),
    (0x7FE0, 0x0010): ("PixelData", "OW"),
}
_STRUCT = {"US": "<H", "SS": "<h", "UL": "<I", "SL": "<i", "FL": "<f", "FD": "<d"}


class DicomError(ValueError):
    """The file is not a DICOM object this reader understands (raised instead of guessing)."""


def _convert(vr: str, raw: bytes) -> Any:
    """Value as pydicom's `Dataset.get` would hand it out, in plain Python types: numbers for DS / IS / binary VRs, stripped
    strings otherwise; a multi-valued element is a list."""
    if vr in _STRUCT:
        size = struct.calcsize(_STRUCT[vr])
        vals = [struct.unpack_from(_STRUCT[vr], raw, i)[0] for i in range(0, len(raw) - size + 1, size)]
        return vals[0] if len(vals) == 1 else (vals or None)
    if vr in ("OB", "OW", "OF", "OD", "UN", "OL", "OV"):
        return raw
    text = raw.decode("latin-1").rstrip("\0 ")
    if vr in ("UT", "ST", "LT", "UR"):            # single-valued text: a backslash is data
        return text
    parts = [p.strip(" ") for p in text.split("\\")]
    if vr == "DS":
        parts = [float(p) for p in parts if p != ""]
    elif vr == "IS":
        parts = [int(p) for p in parts if p != ""]
    if not parts:
        return None
    return parts[0] if len(parts) == 1 else parts


class _Cursor:
    def __init__(self, buf: bytes, pos: int = 0):
        self.buf, self.pos = buf, pos

    def take(self, n: int) -> bytes:
        if self.pos + n > len(self.buf):
            raise DicomError("truncated DICOM element")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b


def _skip_sequence(cur: _Cursor, explicit: bool) -> None:
    """Skip the items of an undefined-length sequence up to its sequence delimiter (FFFE,E0DD)."""
    while True:
        g, e, length = struct.unpack("<HHI", cur.take(8))
        if (g, e) == (0xFFFE, 0xE0DD):
            return
        if (g, e) != (0xFFFE, 0xE000):
            raise DicomError(f"unexpected tag ({g:04X},{e:04X}) inside a sequence")
        if length != 0xFFFFFFFF:
            cur.take(length)
            continue
        while True:                                # undefined-length item: elements up to the item delimiter (FFFE,E00D)
            g2, e2 = struct.unpack("<HH", cur.buf[cur.pos:cur.pos + 4])
            if (g2, e2) == (0xFFFE, 0xE00D):
                cur.take(8)
                break
            _read_element(cur, explicit, want=None)


def _read_element(cur: _Cursor, explicit: bool, want) -> Optional[Tuple[Tuple[int, int], str, bytes]]:
    g, e = struct.unpack("<HH", cur.take(4))
    if explicit and g != 0xFFFE:
        vr_b = cur.take(2)
        if vr_b in _LONG_VR:
            cur.take(2)
            length = struct.unpack("<I", cur.take(4))[0]
        else:
            if not (vr_b.isalpha() and vr_b.isupper()):
                raise DicomError(f"({g:04X},{e:04X}): {vr_b!r} is not a value representation (file is not explicit VR?)")
            length = struct.unpack("<H", cur.take(2))[0]
        vr = vr_b.decode("ascii")
    else:
        length = struct.unpack("<I", cur.take(4))[0]
        vr = TAGS.get((g, e), ("", "UN"))[1]
    if length == 0xFFFFFFFF:
        if (g, e) == (0x7FE0, 0x0010):
            raise NotImplementedError("encapsulated (compressed) PixelData: only uncompressed transfer syntaxes are read")
        # (PS3.5 6.2.2: an UN element of undefined length holds a sequence encoded in IMPLICIT VR, whatever the file's syntax)
        _skip_sequence(cur, explicit and vr != "UN")
        return (g, e), "SQ", b""
    if want is not None and (g, e) not in want:
        cur.take(length) if (g, e) != (0x7FE0, 0x0010) else cur.take(min(length, len(cur.buf) - cur.pos))
        return (g, e), vr, b""
    return (g, e), vr, cur.take(length)


def read_file(path, stop_before_pixels: bool = False) -> Dict[str, Any]:
    """One DICOM Part-10 file -> {keyword: value} for the attributes in TAGS (absent attributes are absent keys; `PixelData` = raw
    bytes unless stop_before_pixels).  Also `"_explicit"` and `"_path"`."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 132 or buf[128:132] != b"DICM":
        raise DicomError(f"{path}: no DICM preamble (not a DICOM Part-10 file)")
    cur = _Cursor(buf, 132)
    out: Dict[str, Any] = {"_path": str(path)}
    # file meta information: always explicit VR little endian, group 0002
    tsuid = None
    while cur.pos < len(buf):
        g = struct.unpack("<H", buf[cur.pos:cur.pos + 2])[0]
        if g != 0x0002:
            break
        tag, vr, raw = _read_element(cur, True, want=None)
        if tag == (0x0002, 0x0010):
            tsuid = _convert("UI", raw)
    if tsuid is None:
        raise DicomError(f"{path}: no TransferSyntaxUID in the file meta information")
    if tsuid not in (IMPLICIT_LE, EXPLICIT_LE):
        raise NotImplementedError(f"{path}: transfer syntax {tsuid} (compressed, deflated or big endian) is not supported; "
                                  "only implicit / explicit VR little endian (uncompressed) are read")
    explicit = tsuid == EXPLICIT_LE
    out["TransferSyntaxUID"] = tsuid
    out["_explicit"] = explicit
    want = set(TAGS)
    while cur.pos + 8 <= len(buf):
        g, e = struct.unpack("<HH", buf[cur.pos:cur.pos + 4])
        if stop_before_pixels and (g, e) >= (0x7FE0, 0x0010):
            break
        tag, vr, raw = _read_element(cur, explicit, want)
        if tag in TAGS and vr != "SQ":
            kw, dvr = TAGS[tag]
            out[kw] = _convert(dvr, raw)           # (the dictionary VR: an explicit file's own VR agrees for standard attributes)
    return out


def classify_orientation(iop):
    """BOA/compute/io.py:269-283: plane of the slice normal's dominant component."""
    if iop is None or len(iop) != 6:
        return None, None
    row = np.asarray(iop[:3], dtype=float)
    col = np.asarray(iop[3:], dtype=float)
    normal = np.cross(row, col)
    ax, ay, az = abs(normal[0]), abs(normal[1]), abs(normal[2])
    if az >= ax and az >= ay:
        return "axial", normal
    if ay >= ax and ay >= az:
        return "coronal", normal
    return "sagittal", normal


def validate_dicom(dcm: Dict[str, Any], num_dicoms: int, *, minimum_images: int = 10, axial_normal_z_min: float = 0.85) -> Optional[str]:
    """BOA/compute/io.py:286-323, same messages: None when the series is an axial CT acquisition with enough slices."""
    if num_dicoms < minimum_images:
        return f"The series has less than {minimum_images} instances: {num_dicoms}."
    modality = dcm.get("Modality")
    if modality is not None and modality != "CT":
        return f"The modality is not CT: {modality}"
    iop = dcm.get("ImageOrientationPatient")
    if iop is not None:
        plane, normal = classify_orientation(iop)
        if plane is not None and normal is not None and plane != "axial":
            return f"Image plane is {plane}, not axial. IOP={list(iop)}, slice normal={normal.tolist()}"
        if normal is not None and abs(normal[2]) < axial_normal_z_min:
            return f"Axial but tilted beyond tolerance: |normal_z|={abs(normal[2]):.3f} < {axial_normal_z_min}. IOP={list(iop)}"
    it = dcm.get("ImageType") or ()
    image_type = set([it] if isinstance(it, str) else it)
    hits = {"LOCALIZER", "REFORMATTED", "DERIVED", "PROJECTION IMAGE"} & image_type
    if hits:
        return f"ImageType contains disqualifying marker(s) {hits}: {list(image_type)}"
    return None


def series_file_names(folder) -> List[str]:
    """`GetGDCMSeriesFileNames(folder)`: the files of the folder's first series, sorted along the slice normal."""
    folder = pathlib.Path(folder)
    if not folder.is_dir():
        raise FileNotFoundError(f"{folder} is not a directory")
    heads = []
    for name in sorted(os.listdir(folder)):
        p = folder / name
        if not p.is_file():
            continue
        try:
            h = read_file(p, stop_before_pixels=True)
        except DicomError:
            continue                                # (not a DICOM file: GDCM's directory scan skips it too)
        if "SeriesInstanceUID" in h:
            heads.append(h)
    if not heads:
        raise ValueError(f"no DICOM series found in {folder}")
    first_uid = sorted({h["SeriesInstanceUID"] for h in heads})[0]
    heads = [h for h in heads if h["SeriesInstanceUID"] == first_uid]
    for h in heads:
        if h.get("ImagePositionPatient") is None or len(h["ImagePositionPatient"]) != 3:
            raise ValueError(f"{h['_path']}: no ImagePositionPatient (the series cannot be ordered)")
        if h.get("ImageOrientationPatient") is None or len(h["ImageOrientationPatient"]) != 6:
            raise ValueError(f"{h['_path']}: no ImageOrientationPatient")
    iop = np.asarray(heads[0]["ImageOrientationPatient"], dtype=np.float64)
    normal = np.cross(iop[:3], iop[3:])
    heads.sort(key=lambda h: (float(np.dot(normal, np.asarray(h["ImagePositionPatient"], dtype=np.float64))), h["_path"]))
    return [h["_path"] for h in heads]


def _output_dtype(ds: Dict[str, Any]) -> np.dtype:
    """The pixel type after the modality rescale, chosen as GDCM's Rescaler does (ComputeInterceptSlopePixelType): integer
    slope and intercept -> the smallest integer type that holds slope * [stored range] + intercept; otherwise float64."""
    slope, inter = float(ds.get("RescaleSlope", 1.0) or 1.0), float(ds.get("RescaleIntercept", 0.0) or 0.0)
    bits = int(ds.get("BitsStored", ds["BitsAllocated"]))
    signed = int(ds.get("PixelRepresentation", 0)) == 1
    lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
    if slope != int(slope) or inter != int(inter):
        return np.dtype(np.float64)
    a, b = sorted((slope * lo + inter, slope * hi + inter))
    for dt in ((np.uint8, np.uint16, np.uint32) if a >= 0 else (np.int8, np.int16, np.int32)):
        info = np.iinfo(dt)
        if info.min <= a and b <= info.max:
            return np.dtype(dt)
    return np.dtype(np.float64)


def _slice_pixels(ds: Dict[str, Any]) -> np.ndarray:
    rows, cols, alloc = int(ds["Rows"]), int(ds["Columns"]), int(ds["BitsAllocated"])
    if int(ds.get("SamplesPerPixel", 1)) != 1 or ds.get("PhotometricInterpretation", "MONOCHROME2") != "MONOCHROME2":
        raise NotImplementedError(f"{ds['_path']}: only MONOCHROME2 single-sample images are read")
    if int(ds.get("NumberOfFrames", 1) or 1) != 1:
        raise NotImplementedError(f"{ds['_path']}: multi-frame objects are not read")
    if alloc not in (8, 16, 32):
        raise NotImplementedError(f"{ds['_path']}: BitsAllocated {alloc}")
    signed = int(ds.get("PixelRepresentation", 0)) == 1
    dt = np.dtype({8: "i1" if signed else "u1", 16: "<i2" if signed else "<u2", 32: "<i4" if signed else "<u4"}[alloc])
    raw = ds.get("PixelData")
    if raw is None or len(raw) < rows * cols * dt.itemsize:
        raise DicomError(f"{ds['_path']}: PixelData holds {0 if raw is None else len(raw)} bytes, {rows * cols * dt.itemsize} expected")
    px = np.frombuffer(raw, dtype=dt, count=rows * cols).reshape(rows, cols)
    stored = int(ds.get("BitsStored", alloc))
    if stored < alloc:                              # bits above BitsStored are not part of the value (HighBit = BitsStored - 1 assumed)
        if int(ds.get("HighBit", stored - 1)) != stored - 1:
            raise NotImplementedError(f"{ds['_path']}: HighBit != BitsStored - 1")
        wide = px.astype(np.int64)
        wide &= (1 << stored) - 1
        if signed:
            wide = np.where(wide >= (1 << (stored - 1)), wide - (1 << stored), wide)
        px = wide
    return px


def load_series(folder) -> Tuple[np.ndarray, Dict[str, Any], List[str]]:
    """Folder -> (volume [x, y, z] in the file axis order of the NIfTI that `sitk.WriteImage` would write, geometry, files).
    geometry: LPS `origin` (3), `spacing` (3), `direction` (3 x 3, columns = row-direction / column-direction / slice normal),
    and the RAS `affine` (4 x 4) of the NIfTI file."""
    files = series_file_names(folder)
    sl = [read_file(p) for p in files]
    first = sl[0]
    rows, cols = int(first["Rows"]), int(first["Columns"])
    iop = np.asarray(first["ImageOrientationPatient"], dtype=np.float64)
    ps = first.get("PixelSpacing")
    if ps is None or not isinstance(ps, list) or len(ps) != 2:
        raise ValueError(f"{first['_path']}: no PixelSpacing")
    for d in sl[1:]:
        if (int(d["Rows"]), int(d["Columns"])) != (rows, cols):
            raise ValueError(f"{d['_path']}: slice size differs within the series")
        if not np.allclose(np.asarray(d["ImageOrientationPatient"], dtype=np.float64), iop, atol=1e-4):
            raise ValueError(f"{d['_path']}: ImageOrientationPatient differs within the series")
        if not np.allclose(np.asarray(d.get("PixelSpacing"), dtype=np.float64), np.asarray(ps, dtype=np.float64), atol=1e-6):
            raise ValueError(f"{d['_path']}: PixelSpacing differs within the series")
    row_dir, col_dir = iop[:3], iop[3:]
    normal = np.cross(row_dir, col_dir)
    pos = np.asarray([d["ImagePositionPatient"] for d in sl], dtype=np.float64)
    n = len(sl)
    if n > 1:
        proj = pos @ normal
        steps = np.diff(proj)
        if (np.abs(steps) < 1e-6).any():
            raise ValueError("two slices of the series share a position (duplicated instances)")
        mean = float(np.linalg.norm(pos[-1] - pos[0])) / (n - 1)
        tol = max(0.01 * mean, 0.01)
        # in-plane drift of the slice origins (a sheared stack): ITK would silently ignore it
        drift = (pos - pos[0]) - np.outer(proj - proj[0], normal)
        if np.abs(drift).max() > tol:
            raise ValueError("slice origins are not aligned along the slice normal (sheared series)")
        if (np.abs(steps - mean) > tol).any():
            k = int(np.argmax(np.abs(steps - mean)))
            raise ValueError(f"non-uniform slice distances (missing slice?): {steps[k]:.4f} mm between {files[k]} and {files[k + 1]}, "
                             f"{mean:.4f} mm on average")
        dz = mean
    else:
        dz = float(first.get("SliceThickness") or 1.0)
    odt = _output_dtype(first)
    vol = np.empty((n, rows, cols), dtype=odt)
    for i, d in enumerate(sl):
        px = _slice_pixels(d)
        slope, inter = float(d.get("RescaleSlope", 1.0) or 1.0), float(d.get("RescaleIntercept", 0.0) or 0.0)
        if odt.kind == "f":
            vol[i] = px.astype(np.float64) * slope + inter
        else:
            val = px.astype(np.int64) * int(slope) + int(inter)
            info = np.iinfo(odt)
            if val.min() < info.min or val.max() > info.max:
                raise ValueError(f"{d['_path']}: rescaled values leave the range of {odt} chosen from the first slice")
            vol[i] = val
    # ITK index order (x = column index, y = row index, z = slice): NIfTI file order x fastest
    data = np.ascontiguousarray(vol.transpose(2, 1, 0))
    spacing = np.array([float(ps[1]), float(ps[0]), dz])       # PixelSpacing = (row spacing = dy, column spacing = dx)
    direction = np.stack([row_dir, col_dir, normal], axis=1)
    origin = pos[0]
    lps = np.eye(4)
    lps[:3, :3] = direction * spacing[None, :]
    lps[:3, 3] = origin
    affine = np.diag([-1.0, -1.0, 1.0, 1.0]) @ lps             # LPS -> RAS, as ITK's NiftiImageIO writes the s/q-form
    geom = {"origin": origin, "spacing": spacing, "direction": direction, "affine": affine}
    return data, geom, files


def _parse_da(value) -> Optional[Tuple[int, int, int]]:
    """`_safe_da` (BOA/compute/io.py:48-54): a DA value -> (year, month, day), None when empty or malformed."""
    if not value:
        return None
    s = str(value).strip().replace(".", "")      # (the retired ACR-NEMA form yyyy.mm.dd is accepted by pydicom's DA)
    if len(s) != 8 or not s.isdigit():
        return None
    y, m, d = int(s[:4]), int(s[4:6]), int(s[6:8])
    import datetime
    try:
        datetime.date(y, m, d)
    except ValueError:
        return None
    return y, m, d


def ct_info_from_dataset(dcm: Dict[str, Any]) -> List[Dict[str, Any]]:
    """The (name, value) list of BOA/compute/io.py:340-383, in the same display order."""
    sd, bd = _parse_da(dcm.get("SeriesDate")), _parse_da(dcm.get("PatientBirthDate"))
    age = None
    if sd and bd:
        age = sd[0] - bd[0] - ((sd[1], sd[2]) < (bd[1], bd[2]))      # `_compute_age`, :262-267
    kernel = dcm.get("ConvolutionKernel")
    if isinstance(kernel, list):                                       # `_first_if_multi`
        kernel = kernel[0] if kernel else None
    ordered = [
        ("StudyInstanceUID", dcm.get("StudyInstanceUID")),
        ("SeriesInstanceUID", dcm.get("SeriesInstanceUID")),
        ("Date", "%02d.%02d.%04d" % (sd[2], sd[1], sd[0]) if sd else None),
        ("AgeYears", age),
        ("Gender", dcm.get("PatientSex")),
        ("AccessionNumber", dcm.get("AccessionNumber")),
        ("SeriesNumber", dcm.get("SeriesNumber")),
        ("SeriesDescription", dcm.get("SeriesDescription")),
        ("Modality", dcm.get("Modality")),
        ("CTDIvol", dcm.get("CTDIvol")),
        ("ExposureTime", dcm.get("ExposureTime")),
        ("XRayTubeCurrent", dcm.get("XRayTubeCurrent")),
        ("Exposure", dcm.get("Exposure")),
        ("KVP", dcm.get("KVP")),
        ("SpiralPitchFactor", dcm.get("SpiralPitchFactor")),
        ("ConvolutionKernel", kernel),
        ("SliceThickness", dcm.get("SliceThickness")),
    ]
    ps = dcm.get("PixelSpacing")
    if isinstance(ps, list) and len(ps) >= 2:
        ordered += [("PixelSpacingX", ps[0]), ("PixelSpacingY", ps[1])]
    else:
        ordered.append(("PixelSpacing", ps))
    ordered.append(("ScanLength", dcm.get("ScanLength")))
    return [{"name": k, "value": v} for k, v in ordered]
