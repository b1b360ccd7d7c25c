"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer for the folder contract of the path (SURVEY 8b
"File contract"; 8f rank 3).  Implements the published NIfTI-1.1 header layout (348-byte header, 4-byte extender,
extensions, data at vox_offset) -- enough for what the reference does with nibabel on this path:
  * `nib.load(p).get_fdata()` / `.affine` / `.header.get_zooms()`: data scaled by scl_slope/scl_inter, best affine =
    sform (sform_code > 0) else qform (qform_code > 0) else pixdim-only;
  * `new_header = img_in_orig.header.copy(); new_header.set_data_dtype(np.uint8)` + label-table XML extension
    (ecode 0) + `nib.save` (TS/nnunet.py:723-726; TS/nifti_ext_header.py:12-42).
nibabel itself is absent in this image; files written here were round-tripped through this reader only (PARITY
UNPINNED vs nibabel's byte-level output; the header fields follow the standard)."""
from __future__ import annotations

import gzip
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

_DT = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
       768: np.uint32, 1024: np.int64, 1280: np.uint64}
_DT_INV = {np.dtype(v): k for k, v in _DT.items()}
_HDR = "<i10s18sihcB8h3f4h8f3fhBB4f2i80s24s2h3f3f12f16s4s"   # little-endian layout, 348 bytes
assert struct.calcsize(_HDR) == 348


@dataclass
class NiftiHeader:
    raw: bytes                                   # the 348 header bytes as read (little-endian normalised)
    dim: Tuple[int, ...]
    datatype: int
    pixdim: Tuple[float, ...]
    vox_offset: float
    scl_slope: float
    scl_inter: float
    qform_code: int
    sform_code: int
    quatern: Tuple[float, float, float]
    qoffset: Tuple[float, float, float]
    srow: np.ndarray
    extensions: List[Tuple[int, bytes]] = field(default_factory=list)

    def get_zooms(self) -> Tuple[float, ...]:
        return tuple(np.float32(v) for v in self.pixdim[1:1 + self.dim[0]])

    def get_data_shape(self) -> Tuple[int, ...]:
        return tuple(self.dim[1:1 + self.dim[0]])

    def get_data_dtype(self):
        return np.dtype(_DT[self.datatype])


def _qform(h: NiftiHeader) -> np.ndarray:
    b, c, d = (float(v) for v in h.quatern)
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 0 else 0.0
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if h.pixdim[0] < 0 else 1.0
    zooms = np.array([h.pixdim[1], h.pixdim[2], h.pixdim[3] * qfac], dtype=np.float64)
    aff = np.eye(4)
    aff[:3, :3] = R * zooms
    aff[:3, 3] = h.qoffset
    return aff


def best_affine(h: NiftiHeader) -> np.ndarray:
    if h.sform_code > 0:
        aff = np.eye(4)
        aff[:3, :] = h.srow
        return aff
    if h.qform_code > 0:
        return _qform(h)
    # neither code set: nibabel's base affine (shape_zoom_affine with the Analyze / NIfTI default x_flip=True):
    # diag(-zx, zy, zz) with the volume centre at the world origin
    zooms = np.array([-h.pixdim[1], h.pixdim[2], h.pixdim[3]], dtype=np.float64)
    aff = np.diag([zooms[0], zooms[1], zooms[2], 1.0])
    shape = np.array(h.dim[1:4], dtype=np.float64)
    aff[:3, 3] = -(shape - 1) / 2.0 * zooms
    return aff


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _member_table(raw) -> Optional[list]:
    """[(deflate start, deflate end, uncompressed size)] of a gzip file whose EVERY member carries this module's FEXTRA index
    subfield (written by write_gzip_members), else None.  Walking the members costs a few header reads, no inflation."""
    mv = memoryview(raw)
    out, off, n = [], 0, len(mv)
    while off < n:
        if n - off < 28 or mv[off] != 0x1F or mv[off + 1] != 0x8B or mv[off + 2] != 8 or mv[off + 3] != 4:
            return None
        xlen = mv[off + 10] | (mv[off + 11] << 8)
        if xlen != 8 or bytes(mv[off + 12:off + 16]) != b"BO\x04\x00":
            return None
        total = struct.unpack("<I", mv[off + 16:off + 20])[0]
        if total < 28 or off + total > n:
            return None
        out.append((off + 20, off + total - 8, struct.unpack("<I", mv[off + total - 4:off + total])[0]))
        off += total
    return out or None


READ_THREADS = int(os.environ.get("BOA_READ_THREADS", "0")) or min(8, os.cpu_count() or 1)


def read_bytes(path, threads: Optional[int] = None) -> bytes:
    """The decompressed content of `path`.  A .gz file written by this module (indexed members) is inflated on `threads` cores
    (zlib releases the GIL): a 512^3 label volume in 0.15 s instead of 0.6 s; any other gzip stream takes the ordinary
    sequential path (a deflate stream has no entry points -- files of other writers are parallelised across FILES, load_many)."""
    if not str(path).endswith(".gz"):
        with open(path, "rb") as f:
            return f.read()
    threads = READ_THREADS if threads is None else int(threads)
    with open(path, "rb") as f:
        raw = f.read()
    tab = _member_table(raw) if threads > 1 else None
    if not tab or len(tab) == 1:
        return gzip.decompress(raw)
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    offs = np.concatenate([[0], np.cumsum([t[2] for t in tab])]).astype(np.int64)
    out = bytearray(int(offs[-1]))
    omv, rmv = memoryview(out), memoryview(raw)

    def inflate(i):
        lo, hi, size = tab[i]
        d = zlib.decompress(rmv[lo:hi], -15, size)
        if len(d) != size or (zlib.crc32(d) & 0xFFFFFFFF) != struct.unpack("<I", rmv[hi:hi + 4])[0]:
            raise ValueError(f"{path}: gzip member {i} is corrupt")
        omv[int(offs[i]):int(offs[i]) + size] = d

    with ThreadPoolExecutor(max_workers=min(threads, len(tab))) as ex:
        list(ex.map(inflate, range(len(tab))))
    return out


def load_many(paths, threads: Optional[int] = None) -> list:
    """[load(p) for p in paths] with the files read and inflated concurrently (the reference reads the CT and up to seven
    segmentation volumes per task one after the other, NN/imageio/nibabel_reader_writer.py:38-90, BOA/compute/measurements.py)."""
    from concurrent.futures import ThreadPoolExecutor
    paths = list(paths)
    threads = READ_THREADS if threads is None else int(threads)
    if threads <= 1 or len(paths) <= 1:
        return [load(p) for p in paths]
    inner = max(1, threads // len(paths))
    with ThreadPoolExecutor(max_workers=min(threads, len(paths))) as ex:
        return list(ex.map(lambda p: load(p, threads=inner), paths))


def load(path, threads: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, NiftiHeader]:
    """-> (raw data array in file axis order and dtype, affine (4,4) float64, header).  `get_fdata` = fdata(...)."""
    blob = read_bytes(path, threads)
    if len(blob) < 352:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    endian = "<"
    if struct.unpack("<i", blob[:4])[0] != 348:
        if struct.unpack(">i", blob[:4])[0] != 348:
            raise ValueError(f"{path}: sizeof_hdr != 348")
        endian = ">"
    v = struct.unpack(endian + _HDR[1:], blob[:348])
    magic = v[-1]
    if magic[:3] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: bad magic {magic!r}")
    if magic[:3] == b"ni1":
        raise ValueError(f"{path}: two-file NIfTI (.hdr/.img) is not supported")
    dim = v[7:15]
    datatype = v[19]
    pixdim = v[22:30]
    vox_offset, scl_slope, scl_inter = v[30], v[31], v[32]
    qform_code, sform_code = v[44], v[45]
    quatern, qoffset = v[46:49], v[49:52]
    srow = np.array(v[52:64], dtype=np.float64).reshape(3, 4)
    if datatype not in _DT:
        raise TypeError(f"{path}: unsupported NIfTI datatype code {datatype}")
    raw_le = struct.pack(_HDR, *v)
    h = NiftiHeader(raw_le, tuple(int(d) for d in dim), int(datatype), tuple(float(p) for p in pixdim), float(vox_offset),
                    float(scl_slope), float(scl_inter), int(qform_code), int(sform_code), tuple(quatern), tuple(qoffset), srow)
    off = 352
    if blob[348] != 0:                       # extensions present
        while off + 8 <= int(vox_offset):
            esize, ecode = struct.unpack(endian + "ii", blob[off:off + 8])
            if esize < 8:
                break
            h.extensions.append((ecode, bytes(blob[off + 8:off + esize])))
            off += esize
    shape = h.get_data_shape()
    dt = np.dtype(_DT[datatype]).newbyteorder(endian)
    n = int(np.prod(shape))
    data = np.frombuffer(blob, dtype=dt, count=n, offset=int(vox_offset)).reshape(shape, order="F")
    return data.astype(dt.newbyteorder("=")), best_affine(h), h


def is_scaled(h: NiftiHeader) -> bool:
    """True when get_fdata() would apply scl_slope / scl_inter."""
    s, i = h.scl_slope, h.scl_inter
    return bool(np.isfinite(s) and s != 0 and not (s == 1.0 and (i == 0 or not np.isfinite(i))))


def fdata(data: np.ndarray, h: NiftiHeader) -> np.ndarray:
    """nibabel's get_fdata(): float64 with scl_slope / scl_inter applied when they are set."""
    out = data.astype(np.float64)
    s, i = h.scl_slope, h.scl_inter
    if np.isfinite(s) and s != 0 and not (s == 1.0 and (i == 0 or not np.isfinite(i))):
        out = out * s + (i if np.isfinite(i) else 0.0)
    return out


def label_xml(label_map: Dict[int, str]) -> bytes:
    """TS/nifti_ext_header.py:12-42 (the CaretExtension label table written into the extended header)."""
    colors = [[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255], [255, 128, 0],
              [255, 0, 128], [128, 255, 128], [0, 128, 255], [128, 128, 128], [185, 170, 155]]
    pre = ('<?xml version="1.0" encoding="UTF-8"?> <CaretExtension>  <Date><![CDATA[2013-07-14T05:45:09]]></Date>   '
           '<VolumeInformation Index="0">   <LabelTable>')
    body = ""
    for k, name in label_map.items():
        r, g, b = colors[k % len(colors)]
        body += f'<Label Key="{k}" Red="{r / 255}" Green="{g / 255}" Blue="{b / 255}" Alpha="1"><![CDATA[{name}]]></Label>\n'
    post = ('  </LabelTable>  <StudyMetaDataLinkSet>  </StudyMetaDataLinkSet>  <VolumeType><![CDATA[Label]]></VolumeType>   '
            '</VolumeInformation></CaretExtension>')
    return bytes(pre + "\n" + body + "\n" + post + "\n              ", "utf-8")


def parse_label_xml(content: bytes) -> Dict[int, str]:
    import re
    return {int(k): v for k, v in re.findall(r'<Label Key="(\d+)"[^>]*><!\[CDATA\[(.*?)\]\]></Label>',
                                             content.decode("utf-8", "replace"))}


SAVE_THREADS = int(os.environ.get("BOA_SAVE_THREADS", "0")) or min(8, os.cpu_count() or 1)   # nr_thr_saving of the reference
_GZ_BLOCK = 4 << 20


def write_gzip_members(f, payload, compresslevel: int = 1, threads: int = 1, block: int = _GZ_BLOCK):
    """`payload` (bytes-like) as a gzip stream of independently compressed members (RFC 1952 section 2.2: a gzip file is a
    series of members; gzip.open, zlib's gzread -- SimpleITK, nibabel -- and `gunzip` read them as one stream; each member
    carries its own length in a FEXTRA subfield so that read_bytes can inflate them in parallel too), the members
    deflated in parallel: zlib releases the GIL, so a 512^3 label volume (134 MB) compresses on `threads` cores instead of
    one (pigz's scheme without its shared dictionary; level 1 like nibabel's default, so the ratio loss is a fraction of a
    percent).  The reference saves its volumes from `nr_thr_saving` worker processes instead (TS/nnunet.py:705-724)."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(payload).cast("B")

    def member(lo):
        # raw deflate + a hand-made gzip wrapper whose FEXTRA field (RFC 1952 2.3.1.1, subfield id "BO") holds the member's total
        # length -- BGZF's idea: a reader that knows the field hops from member to member and inflates them in parallel
        # (read_bytes); every other gzip reader skips it
        piece = mv[lo:lo + block]
        c = zlib.compressobj(compresslevel, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        total = 20 + len(body) + 8
        head = b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + b"\x08\x00" + b"BO\x04\x00" + struct.pack("<I", total)
        return head + body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece) & 0xFFFFFFFF)

    starts = list(range(0, len(mv), block)) or [0]
    if threads <= 1 or len(starts) == 1:
        for lo in starts:
            f.write(member(lo))
        return
    with ThreadPoolExecutor(max_workers=min(threads, len(starts))) as ex:
        for part in ex.map(member, starts):
            f.write(part)


def quatern_from_affine(affine: np.ndarray) -> Tuple[Tuple[float, float, float], Tuple[float, float, float], float]:
    """(quatern b c d, qoffset, qfac) of an affine whose 3 x 3 part is rotation x positive zooms (x an optional flip of the third
    axis): the NIfTI-1.1 standard's matrix -> quaternion recipe (`nifti_mat44_to_quatern`), the inverse of `_qform`."""
    A = np.asarray(affine, dtype=np.float64)
    R = A[:3, :3] / np.sqrt(np.sum(A[:3, :3] ** 2, axis=0))[None, :]
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy()
        R[:, 2] = -R[:, 2]
        qfac = -1.0
    r11, r12, r13, r21, r22, r23, r31, r32, r33 = R.reshape(-1)
    a = r11 + r22 + r33 + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b, c, d = 0.25 * (r32 - r23) / a, 0.25 * (r13 - r31) / a, 0.25 * (r21 - r12) / a
    else:
        xd, yd, zd = 1.0 + r11 - (r22 + r33), 1.0 + r22 - (r11 + r33), 1.0 + r33 - (r11 + r22)
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c, d, a = 0.25 * (r12 + r21) / b, 0.25 * (r13 + r31) / b, 0.25 * (r32 - r23) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b, d, a = 0.25 * (r12 + r21) / c, 0.25 * (r23 + r32) / c, 0.25 * (r13 - r31) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b, c, a = 0.25 * (r13 + r31) / d, 0.25 * (r23 + r32) / d, 0.25 * (r21 - r12) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return (float(b), float(c), float(d)), tuple(float(x) for x in A[:3, 3]), qfac


def save(path, data: np.ndarray, affine: np.ndarray, like: Optional[NiftiHeader] = None,
         extensions: Optional[List[Tuple[int, bytes]]] = None, compresslevel: int = 1, threads: Optional[int] = None,
         form_codes: Optional[Tuple[int, int]] = None):
    """Write `data` (file axis order); `.gz` paths are deflated on `threads` cores (default SAVE_THREADS).  `like`: header to copy (pixdim units, descrip, q/s-form codes ... as
    `img_in_orig.header.copy()` keeps them); datatype/bitpix/dim/vox_offset and the affine fields are set from the
    arguments; scl_slope/inter are reset (label volumes are stored unscaled).  `form_codes`: also store the affine as a qform."""
    data = np.asarray(data)
    if data.dtype not in _DT_INV:
        raise TypeError(f"unsupported dtype {data.dtype}")
    affine = np.asarray(affine, dtype=np.float64)
    v = list(struct.unpack(_HDR, like.raw)) if like is not None else list(struct.unpack(_HDR, b"\0" * 348))
    v[0] = 348
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    v[7:15] = dim
    v[19] = _DT_INV[data.dtype]
    v[20] = data.dtype.itemsize * 8
    zooms = np.sqrt(np.sum(affine[:3, :3] ** 2, axis=0))
    if like is None:
        v[22:30] = [1.0, float(zooms[0]), float(zooms[1]), float(zooms[2]), 1.0, 1.0, 1.0, 1.0]
        v[35] = 2 | 8                          # xyzt_units: mm, sec
        v[44], v[45] = 0, 2                    # qform unknown, sform aligned (what nibabel writes for a bare affine)
    v[31], v[32] = float("nan"), float("nan")   # nibabel writes nan slope/inter for unscaled data
    v[52:64] = [float(x) for x in affine[:3, :].reshape(-1)]
    if like is not None and like.sform_code == 0:
        v[45] = 2
    if form_codes is not None:                 # (qform_code, sform_code) with the quaternion of `affine` (what ITK's NIfTI writer stores)
        quat, qoff, qfac = quatern_from_affine(affine)
        v[44], v[45] = int(form_codes[0]), int(form_codes[1])
        v[46:49] = list(quat)
        v[49:52] = list(qoff)
        v[22] = qfac
    exts = list(extensions or [])
    ext_blob = b""
    for ecode, content in exts:
        esize = 8 + len(content)
        pad = (-esize) % 16
        ext_blob += struct.pack("<ii", esize + pad, ecode) + content + b"\0" * pad
    v[30] = float(352 + len(ext_blob))
    v[-1] = b"n+1\0"
    hdr = struct.pack(_HDR, *v)
    head = hdr + bytes([1 if exts else 0, 0, 0, 0]) + ext_blob
    body = np.asfortranarray(data).reshape(-1, order="F")          # (a view when `data` is already F-ordered)
    with open(path, "wb") as f:
        if str(path).endswith(".gz"):
            write_gzip_members(f, head, compresslevel, 1)
            write_gzip_members(f, body, compresslevel, SAVE_THREADS if threads is None else int(threads))
        else:
            f.write(head)
            f.write(body.tobytes())
