"""Test-only DICOM Part-10 writer (PS3.5 / PS3.10): single-frame CT slices in explicit or implicit VR little endian, written from the
published encoding rules independently of boa_hip/dicom.py's reader tables (own tag list, own VR rules).  No fixture of the reference
exists for DICOM input and pydicom / GDCM are absent, so the reader is exercised against THESE files only: parity unpinned."""
import os
import struct

import numpy as np

EXPLICIT = "1.2.840.10008.1.2.1"
IMPLICIT = "1.2.840.10008.1.2"
CT_STORAGE = "1.2.840.10008.5.1.4.1.1.2"
_LONG = {"OB", "OW", "SQ", "UN", "UT"}


def _pad(b: bytes, vr: str) -> bytes:
    if len(b) % 2:
        b += b"\0" if vr in ("UI", "OB", "OW", "UN") else b" "
    return b


def _value(vr: str, v) -> bytes:
    if isinstance(v, bytes):
        return _pad(v, vr)
    if vr == "US":
        return b"".join(struct.pack("<H", int(x)) for x in (v if isinstance(v, (list, tuple)) else [v]))
    if vr == "FD":
        return b"".join(struct.pack("<d", float(x)) for x in (v if isinstance(v, (list, tuple)) else [v]))
    if isinstance(v, (list, tuple)):
        v = "\\".join(_fmt(vr, x) for x in v)
    else:
        v = _fmt(vr, v)
    return _pad(v.encode("latin-1"), vr)


def _fmt(vr, x):
    if vr == "DS":
        s = repr(float(x))
        return s if len(s) <= 16 else "%.10g" % float(x)
    if vr == "IS":
        return str(int(x))
    return str(x)


def element(tag, vr, value, explicit=True) -> bytes:
    g, e = tag
    body = _value(vr, value)
    if not explicit:
        return struct.pack("<HHI", g, e, len(body)) + body
    if vr in _LONG:
        return struct.pack("<HH2sHI", g, e, vr.encode(), 0, len(body)) + body
    return struct.pack("<HH2sH", g, e, vr.encode(), len(body)) + body


def sequence_undefined(tag, items, explicit=True) -> bytes:
    """A sequence of undefined length with undefined-length items (the form scanners write for e.g. ProcedureCodeSequence)."""
    g, e = tag
    head = struct.pack("<HH2sHI", g, e, b"SQ", 0, 0xFFFFFFFF) if explicit else struct.pack("<HHI", g, e, 0xFFFFFFFF)
    body = b""
    for it in items:
        body += struct.pack("<HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + it + struct.pack("<HHI", 0xFFFE, 0xE00D, 0)
    return head + body + struct.pack("<HHI", 0xFFFE, 0xE0DD, 0)


def write_slice(path, pixels, *, ipp, iop=(1, 0, 0, 0, 1, 0), spacing=(0.8, 0.7), slope=1, intercept=-1024, explicit=True,
                series_uid="1.2.826.0.1.3680043.8.498.1", study_uid="1.2.826.0.1.3680043.8.498.0", instance=1, modality="CT",
                image_type=("ORIGINAL", "PRIMARY", "AXIAL"), bits_stored=16, signed=False, extra=None, transfer_syntax=None,
                with_sequence=True, private_implicit_unknown=True):
    """pixels: [rows, cols] stored values (before the rescale).  spacing = PixelSpacing (row spacing, column spacing)."""
    px = np.asarray(pixels)
    rows, cols = px.shape
    ts = transfer_syntax or (EXPLICIT if explicit else IMPLICIT)
    sop_uid = f"{series_uid}.{instance}"
    meta_body = (element((2, 1), "OB", b"\0\1") + element((2, 2), "UI", CT_STORAGE) + element((2, 3), "UI", sop_uid) +
                 element((2, 0x10), "UI", ts) + element((2, 0x12), "UI", "1.2.826.0.1.3680043.8.498.99"))
    meta = struct.pack("<HH2sHI", 2, 0, b"UL", 4, len(meta_body)) + meta_body   # (0002,0000) group length
    E = lambda tag, vr, v: element(tag, vr, v, explicit)   # noqa: E731
    ds = []
    ds.append(((0x0008, 0x0008), E((0x0008, 0x0008), "CS", list(image_type))))
    ds.append(((0x0008, 0x0016), E((0x0008, 0x0016), "UI", CT_STORAGE)))
    ds.append(((0x0008, 0x0018), E((0x0008, 0x0018), "UI", sop_uid)))
    ds.append(((0x0008, 0x0021), E((0x0008, 0x0021), "DA", "20240317")))
    ds.append(((0x0008, 0x0050), E((0x0008, 0x0050), "SH", "ACC0042")))
    ds.append(((0x0008, 0x0060), E((0x0008, 0x0060), "CS", modality)))
    ds.append(((0x0008, 0x103E), E((0x0008, 0x103E), "LO", "Abdomen 1.5 B31f")))
    if with_sequence:   # (0008,1032) ProcedureCodeSequence: one item with a nested element; a reader must skip it by structure
        item = element((0x0008, 0x0100), "SH", "CTABD", explicit) + element((0x0008, 0x0104), "LO", "CT Abdomen", explicit)
        ds.append(((0x0008, 0x1032), sequence_undefined((0x0008, 0x1032), [item], explicit)))
    ds.append(((0x0010, 0x0030), E((0x0010, 0x0030), "DA", "19600502")))
    ds.append(((0x0010, 0x0040), E((0x0010, 0x0040), "CS", "F")))
    ds.append(((0x0018, 0x0050), E((0x0018, 0x0050), "DS", 1.5)))
    ds.append(((0x0018, 0x0060), E((0x0018, 0x0060), "DS", 120)))
    ds.append(((0x0018, 0x1150), E((0x0018, 0x1150), "IS", 500)))
    ds.append(((0x0018, 0x1151), E((0x0018, 0x1151), "IS", 220)))
    ds.append(((0x0018, 0x1152), E((0x0018, 0x1152), "IS", 110)))
    ds.append(((0x0018, 0x1210), E((0x0018, 0x1210), "SH", ["B31f", "3"])))
    ds.append(((0x0018, 0x9311), E((0x0018, 0x9311), "FD", 0.6)))
    ds.append(((0x0018, 0x9345), E((0x0018, 0x9345), "FD", 7.25)))
    if private_implicit_unknown:   # a private element the reader's dictionary cannot know (must be skipped by its length)
        ds.append(((0x0019, 0x0010), E((0x0019, 0x0010), "LO", "TEST PRIVATE")))
        ds.append(((0x0019, 0x1001), E((0x0019, 0x1001), "UN", b"\x01\x02\x03\x04\x05\x06")))
    ds.append(((0x0020, 0x000D), E((0x0020, 0x000D), "UI", study_uid)))
    ds.append(((0x0020, 0x000E), E((0x0020, 0x000E), "UI", series_uid)))
    ds.append(((0x0020, 0x0011), E((0x0020, 0x0011), "IS", 4)))
    ds.append(((0x0020, 0x0013), E((0x0020, 0x0013), "IS", instance)))
    if ipp is not None:
        ds.append(((0x0020, 0x0032), E((0x0020, 0x0032), "DS", list(ipp))))
    if iop is not None:
        ds.append(((0x0020, 0x0037), E((0x0020, 0x0037), "DS", list(iop))))
    ds.append(((0x0028, 0x0002), E((0x0028, 0x0002), "US", 1)))
    ds.append(((0x0028, 0x0004), E((0x0028, 0x0004), "CS", "MONOCHROME2")))
    ds.append(((0x0028, 0x0010), E((0x0028, 0x0010), "US", rows)))
    ds.append(((0x0028, 0x0011), E((0x0028, 0x0011), "US", cols)))
    ds.append(((0x0028, 0x0030), E((0x0028, 0x0030), "DS", list(spacing))))
    ds.append(((0x0028, 0x0100), E((0x0028, 0x0100), "US", 16)))
    ds.append(((0x0028, 0x0101), E((0x0028, 0x0101), "US", bits_stored)))
    ds.append(((0x0028, 0x0102), E((0x0028, 0x0102), "US", bits_stored - 1)))
    ds.append(((0x0028, 0x0103), E((0x0028, 0x0103), "US", 1 if signed else 0)))
    ds.append(((0x0028, 0x1052), E((0x0028, 0x1052), "DS", intercept)))
    ds.append(((0x0028, 0x1053), E((0x0028, 0x1053), "DS", slope)))
    for tag, vr, v in (extra or []):
        ds.append((tag, E(tag, vr, v)))
    ds.append(((0x7FE0, 0x0010), E((0x7FE0, 0x0010), "OW", px.astype("<i2" if signed else "<u2").tobytes())))
    ds.sort(key=lambda t: t[0])
    with open(path, "wb") as f:
        f.write(b"\0" * 128 + b"DICM" + meta + b"".join(b for _, b in ds))


def write_series(folder, volume_zyx_stored, *, origin=(-100.0, -120.0, 50.0), iop=(1, 0, 0, 0, 1, 0), spacing=(0.8, 0.7), dz=1.5,
                 order=None, skip=(), explicit=True, name="IM%04d.dcm", **kw):
    """volume[z] = stored pixels of the slice at origin + z * dz * normal; `order`: file-name order of the z indices (shuffled /
    reversed folders); `skip`: z indices left out (a gapped series)."""
    os.makedirs(folder, exist_ok=True)
    iop_a = np.asarray(iop, dtype=float)
    normal = np.cross(iop_a[:3], iop_a[3:])
    zs = list(range(len(volume_zyx_stored))) if order is None else list(order)
    paths = []
    for k, z in enumerate(zs):
        if z in skip:
            continue
        ipp = np.asarray(origin, dtype=float) + z * dz * normal
        p = os.path.join(folder, name % k)
        write_slice(p, volume_zyx_stored[z], ipp=ipp, iop=iop, spacing=spacing, instance=z + 1, explicit=explicit, **kw)
        paths.append(p)
    return paths
