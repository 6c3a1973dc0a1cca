"""Overlay for lib/model/nms/nms_wrapper.py."""
from detectron_pytorch_amd.nms import nms  # noqa: F401
