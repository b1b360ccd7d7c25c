"""Model-folder contract (SURVEY 8b "Model-folder contract"): find and load nnU-Net v2 model folders below
`$nnUNet_results` (= TOTALSEG_WEIGHTS_PATH, or ~/.totalsegmentator/nnunet/results; TS/config.py:26-51):
    DatasetNNN_<name>/<trainer>__nnUNetPlans__<config>/{dataset.json, plans.json, fold_k/checkpoint_final.pth}
(NN/utilities/file_path_utilities.py:19-26; dataset dir found by its `DatasetNNN` prefix,
NN/utilities/dataset_name_id_conversion.py:21-35).  Nothing is downloaded: a missing folder raises."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import plans as P

# task tables: TS/python_api.py:168-189 (`total`, fast variant), BCA/tasks.py:15-48
TASKS: Dict[str, dict] = {
    "total": {"task_id": [291, 292, 293, 294, 295], "resample": 1.5, "trainer": "nnUNetTrainerNoMirroring", "folds": [0]},
    "total_fast": {"task_id": [297], "resample": 3.0, "trainer": "nnUNetTrainer_4000epochs_NoMirroring", "folds": [0]},
    "body_parts": {"task_id": [543], "resample": 5.0, "trainer": "nnUNetTrainer_1500epochs_NoMirroring",
                   "folds": [0, 1, 2, 3, 4], "resample_only_thickness": True},
    "body_regions": {"task_id": [542], "resample": 5.0, "trainer": "nnUNetTrainerNoMirroring", "folds": [0, 1, 2, 3, 4],
                     "resample_only_thickness": True},
    # rough organ segmentation for the crop cascade (TS/python_api.py:685-707): single model, `total` label map
    "total_6mm": {"task_id": [298], "resample": 6.0, "trainer": "nnUNetTrainer_4000epochs_NoMirroring", "folds": [0]},
    # crop-cascade tasks of `--models all` (TS/python_api.py:236-259,311-329): native resolution (resample None), cropped
    # to the listed `total` structures (+ crop_addon mm); folds None = every fold_k present (nnU-Net's default)
    "lung_vessels": {"task_id": [258], "resample": None, "trainer": "nnUNetTrainer", "folds": [0],
                     "crop": ["lung_upper_lobe_left", "lung_lower_lobe_left", "lung_upper_lobe_right",
                              "lung_middle_lobe_right", "lung_lower_lobe_right"], "crop_addon": [3, 3, 3]},
    "cerebral_bleed": {"task_id": [150], "resample": None, "trainer": "nnUNetTrainer", "folds": [0], "crop": ["brain"],
                       "crop_addon": [3, 3, 3]},
    "hip_implant": {"task_id": [260], "resample": None, "trainer": "nnUNetTrainer", "folds": [0],
                    "crop": ["femur_left", "femur_right", "hip_left", "hip_right"], "crop_addon": [3, 3, 3]},
    "pleural_pericard_effusion": {"task_id": [315], "resample": None, "trainer": "nnUNetTrainer", "folds": None,
                                  "crop": ["lung_upper_lobe_left", "lung_lower_lobe_left", "lung_upper_lobe_right",
                                           "lung_middle_lobe_right", "lung_lower_lobe_right"], "crop_addon": [50, 50, 50]},
    "liver_vessels": {"task_id": [8], "resample": None, "trainer": "nnUNetTrainer", "folds": [0], "crop": ["liver"],
                      "crop_addon": [20, 20, 20]},
    # licensed (TS/python_api.py:493-505): robust_crop = the 3 mm `total` model (Dataset297) makes the crop mask, and the
    # result is cleared outside the 10 mm-dilated union of heart / aorta / inferior vena cava (TS/postprocessing.py:101-131)
    "heartchambers_highres": {"task_id": [301], "resample": None, "trainer": "nnUNetTrainer", "folds": [0], "crop": ["heart"],
                              "crop_addon": [5, 5, 5], "robust_crop": True,
                              "remove_outside": ["heart", "aorta", "inferior_vena_cava"], "remove_outside_dilation": 10},
}
CASCADE_MODELS = ("lung_vessels", "cerebral_bleed", "hip_implant", "pleural_pericard_effusion", "liver_vessels",
                  "heartchambers_highres")


def rough_model_key(task: str) -> str:
    """The model that makes the crop mask of a cascade task (TS/python_api.py:679-699): Dataset298 at 6 mm, or with
    `robust_crop` Dataset297 at 3 mm."""
    return "total_fast" if TASKS[task].get("robust_crop") else "total_6mm"


def effective_crop_addon(task: str) -> List[int]:
    """Crop margin (mm) the reference actually applies.  TASKS[task]["crop_addon"] is the value of the task table
    (TS/python_api.py:236-259,311-329), but whenever the crop mask comes from the default rough `total` model
    (`crop_model is None`, true for every task here: none of them names a `crop_model`) TS/python_api.py:726 overrides it:
        crop_addon = [20,20,20] if crop_model is None else crop_addon"""
    info = TASKS[task]
    if "crop" not in info:
        raise KeyError(f"{task} is not a crop-cascade task")
    return [20, 20, 20] if info.get("crop_model") is None else list(info["crop_addon"])


def results_dir() -> str:
    for var in ("nnUNet_results", "TOTALSEG_WEIGHTS_PATH"):
        if os.environ.get(var):
            return os.environ[var]
    return os.path.join(os.path.expanduser("~"), ".totalsegmentator", "nnunet", "results")


def find_dataset_dir(dataset_id: int, root: str = None) -> str:
    root = root or results_dir()
    prefix = "Dataset%03d" % dataset_id
    if not os.path.isdir(root):
        raise FileNotFoundError(f"nnUNet_results folder {root!r} does not exist (weights are never downloaded here)")
    hits = sorted(d for d in os.listdir(root) if d.startswith(prefix) and os.path.isdir(os.path.join(root, d)))
    if len(hits) != 1:
        raise RuntimeError(f"Found {'no' if not hits else 'more than one'} dataset folder for {prefix} in {root}: {hits}")
    return os.path.join(root, hits[0])


def load_task_models(task: str, fast_bca: bool = False, root: str = None, configuration: str = "3d_fullres"
                     ) -> List[Tuple[int, P.ModelConfig, List[np.ndarray]]]:
    """-> [(task_id, ModelConfig, [weight blob per fold])] for SegmentationTask."""
    info = TASKS[task]
    folds = [0] if (fast_bca and task in ("body_parts", "body_regions")) else info["folds"]
    out = []
    for tid in info["task_id"]:
        folder = os.path.join(find_dataset_dir(tid, root), f"{info['trainer']}__nnUNetPlans__{configuration}")
        cfg = P.load_model_folder(folder, configuration)
        if folds is None:  # nnU-Net: use every fold present (predict_from_raw_data.py:68-73 auto-detection)
            folds = sorted(int(d[5:]) for d in os.listdir(folder) if d.startswith("fold_") and d[5:].isdigit() and
                           os.path.isfile(os.path.join(folder, d, "checkpoint_final.pth")))
            if not folds:
                raise FileNotFoundError(f"no fold_k/checkpoint_final.pth below {folder}")
        blobs = []
        for f in folds:
            ck = os.path.join(folder, f"fold_{f}", "checkpoint_final.pth")
            blobs.append(P.weight_blob_from_state_dict(cfg.geometry, P.load_checkpoint(ck)))
        out.append((tid, cfg, blobs))
    return out


def write_model_folder(root: str, dataset_id: int, name: str, trainer: str, plans: dict, dataset_json: dict,
                       state_dicts: Sequence[Dict[str, np.ndarray]], configuration: str = "3d_fullres") -> str:
    """Write a model folder in the layout above (used for synthetic weights in tests / benchmarks)."""
    import json
    import torch
    folder = os.path.join(root, "Dataset%03d_%s" % (dataset_id, name), f"{trainer}__nnUNetPlans__{configuration}")
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "plans.json"), "w") as f:
        json.dump(plans, f)
    with open(os.path.join(folder, "dataset.json"), "w") as f:
        json.dump(dataset_json, f)
    for k, sd in enumerate(state_dicts):
        os.makedirs(os.path.join(folder, f"fold_{k}"), exist_ok=True)
        torch.save({"network_weights": {n: torch.from_numpy(np.asarray(v)) for n, v in sd.items()},
                    "trainer_name": trainer, "init_args": {"configuration": configuration},
                    "inference_allowed_mirroring_axes": None},
                   os.path.join(folder, f"fold_{k}", "checkpoint_final.pth"))
    return folder
