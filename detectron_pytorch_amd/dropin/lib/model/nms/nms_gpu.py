"""Overlay for lib/model/nms/nms_gpu.py."""
from detectron_pytorch_amd.nms import nms_gpu  # noqa: F401
