"""Overlay for lib/model/roi_crop/modules/roi_crop.py."""
from detectron_pytorch_amd.roi_crop import _RoICrop_Module as _RoICrop  # noqa: F401
