"""Label tables of the TotalSegmentator `total` task as data (structure names, whitespace separated, in label
order starting at 1).  Same content as TS/map_to_binary.py `class_map["total"]` (:116-234), `class_map_5_parts`
(:808-959) and `map_taskid_to_partname_ct` (:1054-1062); pinned by tests/golden/g7_label_tables.json, which is
generated from the reference tables."""

_TOTAL = """
    spleen kidney_right kidney_left gallbladder liver stomach pancreas adrenal_gland_right adrenal_gland_left
    lung_upper_lobe_left lung_lower_lobe_left lung_upper_lobe_right lung_middle_lobe_right
    lung_lower_lobe_right esophagus trachea thyroid_gland small_bowel duodenum colon urinary_bladder prostate
    kidney_cyst_left kidney_cyst_right sacrum vertebrae_S1 vertebrae_L5 vertebrae_L4 vertebrae_L3 vertebrae_L2
    vertebrae_L1 vertebrae_T12 vertebrae_T11 vertebrae_T10 vertebrae_T9 vertebrae_T8 vertebrae_T7 vertebrae_T6
    vertebrae_T5 vertebrae_T4 vertebrae_T3 vertebrae_T2 vertebrae_T1 vertebrae_C7 vertebrae_C6 vertebrae_C5
    vertebrae_C4 vertebrae_C3 vertebrae_C2 vertebrae_C1 heart aorta pulmonary_vein brachiocephalic_trunk
    subclavian_artery_right subclavian_artery_left common_carotid_artery_right common_carotid_artery_left
    brachiocephalic_vein_left brachiocephalic_vein_right atrial_appendage_left superior_vena_cava
    inferior_vena_cava portal_vein_and_splenic_vein iliac_artery_left iliac_artery_right iliac_vena_left
    iliac_vena_right humerus_left humerus_right scapula_left scapula_right clavicula_left clavicula_right
    femur_left femur_right hip_left hip_right spinal_cord gluteus_maximus_left gluteus_maximus_right
    gluteus_medius_left gluteus_medius_right gluteus_minimus_left gluteus_minimus_right autochthon_left
    autochthon_right iliopsoas_left iliopsoas_right brain skull rib_left_1 rib_left_2 rib_left_3 rib_left_4
    rib_left_5 rib_left_6 rib_left_7 rib_left_8 rib_left_9 rib_left_10 rib_left_11 rib_left_12 rib_right_1
    rib_right_2 rib_right_3 rib_right_4 rib_right_5 rib_right_6 rib_right_7 rib_right_8 rib_right_9
    rib_right_10 rib_right_11 rib_right_12 sternum costal_cartilages
"""

_PARTS = {
    291: """
    spleen kidney_right kidney_left gallbladder liver stomach pancreas adrenal_gland_right adrenal_gland_left
    lung_upper_lobe_left lung_lower_lobe_left lung_upper_lobe_right lung_middle_lobe_right
    lung_lower_lobe_right esophagus trachea thyroid_gland small_bowel duodenum colon urinary_bladder prostate
    kidney_cyst_left kidney_cyst_right
""",
    292: """
    sacrum vertebrae_S1 vertebrae_L5 vertebrae_L4 vertebrae_L3 vertebrae_L2 vertebrae_L1 vertebrae_T12
    vertebrae_T11 vertebrae_T10 vertebrae_T9 vertebrae_T8 vertebrae_T7 vertebrae_T6 vertebrae_T5 vertebrae_T4
    vertebrae_T3 vertebrae_T2 vertebrae_T1 vertebrae_C7 vertebrae_C6 vertebrae_C5 vertebrae_C4 vertebrae_C3
    vertebrae_C2 vertebrae_C1
""",
    293: """
    heart aorta pulmonary_vein brachiocephalic_trunk subclavian_artery_right subclavian_artery_left
    common_carotid_artery_right common_carotid_artery_left brachiocephalic_vein_left
    brachiocephalic_vein_right atrial_appendage_left superior_vena_cava inferior_vena_cava
    portal_vein_and_splenic_vein iliac_artery_left iliac_artery_right iliac_vena_left iliac_vena_right
""",
    294: """
    humerus_left humerus_right scapula_left scapula_right clavicula_left clavicula_right femur_left
    femur_right hip_left hip_right spinal_cord gluteus_maximus_left gluteus_maximus_right gluteus_medius_left
    gluteus_medius_right gluteus_minimus_left gluteus_minimus_right autochthon_left autochthon_right
    iliopsoas_left iliopsoas_right brain skull
""",
    295: """
    rib_left_1 rib_left_2 rib_left_3 rib_left_4 rib_left_5 rib_left_6 rib_left_7 rib_left_8 rib_left_9
    rib_left_10 rib_left_11 rib_left_12 rib_right_1 rib_right_2 rib_right_3 rib_right_4 rib_right_5
    rib_right_6 rib_right_7 rib_right_8 rib_right_9 rib_right_10 rib_right_11 rib_right_12 sternum
    costal_cartilages
""",
}

TOTAL_NAMES = _TOTAL.split()
CLASS_MAP_TOTAL = {i + 1: n for i, n in enumerate(TOTAL_NAMES)}
CLASS_MAP_TOTAL_INV = {n: i for i, n in CLASS_MAP_TOTAL.items()}
PART_TASK_IDS = (291, 292, 293, 294, 295)
PART_NAMES = {291: "class_map_part_organs", 292: "class_map_part_vertebrae", 293: "class_map_part_cardiac",
              294: "class_map_part_muscles", 295: "class_map_part_ribs"}
CLASS_MAP_PARTS = {tid: {i + 1: n for i, n in enumerate(txt.split())} for tid, txt in _PARTS.items()}


def part_lut(task_id: int):
    """uint8 LUT local argmax index -> global `total` label (index 0 = background stays 0)."""
    import numpy as np
    pm = CLASS_MAP_PARTS[task_id]
    lut = np.zeros(max(pm) + 1, dtype=np.uint8)
    for j, name in pm.items():
        lut[j] = CLASS_MAP_TOTAL_INV[name]
    return lut


# ---- label tables of the other models BOA measures (data file generated from the reference tables, pinned by G11) ----
_DATA = None


def _data() -> dict:
    global _DATA
    if _DATA is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "measurement_label_maps.json")) as f:
            _DATA = json.load(f)
    return _DATA


def measurement_label_map(model_name: str) -> dict:
    """{region name: label id} exactly as compute_measurements derives it (BOA/compute/measurements.py:289-293): every
    TotalSegmentator class_map task whose name starts with `model_name` contributes (for "total" also the `total_mr` /
    `total_v1` tables as `mr_*` / `v1_*` regions -- a quirk of the reference that its JSON carries)."""
    return {k: int(v) for k, v in _data()["label_maps"][model_name]}


def class_map(task: str) -> dict:
    """{label id: structure name} of a task (TS/map_to_binary.py class_map)."""
    return {int(k): v for k, v in _data()["class_maps"][task].items()}


def output_name(model_name: str) -> str:
    """File stem compute_measurements looks for (BOA/compute/util.py:6-14, measurements.py:263-268)."""
    return _data()["output_names"].get(model_name, model_name)


def cnr_adjusted_regions() -> dict:
    return {k: set(v) for k, v in _data()["cnr_adjusted_regions"].items()}
