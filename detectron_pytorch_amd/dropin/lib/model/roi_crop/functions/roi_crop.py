"""Overlay for lib/model/roi_crop/functions/roi_crop.py."""
from detectron_pytorch_amd.roi_crop import RoICropFunction  # noqa: F401
