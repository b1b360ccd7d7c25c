"""Model sets (BOA/compute/constants.py:16-35)."""
BASE_MODELS = {"bca", "body_regions", "body_parts"}
ALL_MODELS = {"bca", "body_parts", "body_regions", "cerebral_bleed", "hip_implant", "liver_vessels", "lung_vessels",
              "pleural_pericard_effusion", "total"}
LICENSE_MODELS = {"heartchambers_highres"}
AVAILABLE_MODELS = ALL_MODELS | LICENSE_MODELS
ADDITIONAL_MODELS_OUTPUT_NAME = {  # BOA/compute/util.py:6-14
    "lung_vessels": "lung_vessels_airways", "cerebral_bleed": "cerebral_bleed", "hip_implant": "hip_implant",
    "coronary_arteries": "coronary_arteries", "pleural_pericard_effusion": "pleural_pericard_effusion",
    "liver_vessels": "liver_vessels", "heartchambers_highres": "heartchambers",
}
