"""Overlay for lib/model/roi_pooling/modules/roi_pool.py."""
from detectron_pytorch_amd.roi_pool import _RoIPooling  # noqa: F401
