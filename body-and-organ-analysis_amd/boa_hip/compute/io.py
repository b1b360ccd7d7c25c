"""Mirror of the input side of BOA/compute/io.py for the hot path: `get_image_info` (:326-383) -- a folder with one DICOM CT
series -> `<output_folder>/image.nii.gz` + the `ct_info` name / value list the Excel / JSON writers consume.  The DICOM-SEG /
PACS / SMB half of that file (`store_dicoms`, `store_excel`) is the reference's control plane and stays there (DESIGN.md
section 7).  Reader: boa_hip/dicom.py (uncompressed little-endian CT only; parity unpinned vs SimpleITK / GDCM)."""
from __future__ import annotations

import pathlib
from typing import Any, Dict, List, Tuple

from .. import dicom, nifti


def get_image_info(input_folder: pathlib.Path, output_folder: pathlib.Path) -> Tuple[pathlib.Path, List[Dict[str, Any]]]:
    """Same contract as the reference: raises ValueError with `validate_dicom`'s message for a series that is not an axial CT
    acquisition of >= 10 slices; writes the volume as `image.nii.gz` (LPS geometry converted to the RAS s/q-form as ITK's NIfTI
    writer does); returns (path, ct_info)."""
    input_folder, output_folder = pathlib.Path(input_folder), pathlib.Path(output_folder)
    files = dicom.series_file_names(input_folder)
    dcm = dicom.read_file(files[0], stop_before_pixels=True)
    message = dicom.validate_dicom(dcm, len(files))
    if message:
        raise ValueError(message)
    data, geom, _ = dicom.load_series(input_folder)
    output_folder.mkdir(parents=True, exist_ok=True)
    nifti_path = output_folder / "image.nii.gz"
    nifti.save(nifti_path, data, geom["affine"], form_codes=(1, 1))   # NIFTI_XFORM_SCANNER_ANAT for both forms, as ITK writes them
    return nifti_path, dicom.ct_info_from_dataset(dcm)
