"""Overlay for lib/model/roi_pooling/functions/roi_pool.py."""
from detectron_pytorch_amd.roi_pool import RoIPoolFunction  # noqa: F401
