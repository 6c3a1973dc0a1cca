"""ctypes binding of libmi_detectron_ops.so -- the only way this package reaches the GPU.

No torch C++ headers are involved: tensors cross the boundary as `data_ptr()` integers and the
current HIP stream as `torch.cuda.current_stream().cuda_stream`, exactly the information the
reference's C glue pulled out of THCudaTensor / THCState (roi_align_cuda.c:10-31).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_detectron_ops.so")

MI_OK = 0
ABI_VERSION = 8  # MI_ABI_VERSION of include/mi_detectron_ops.h this binding was written against
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
ROI_ALIGN_CAFFE2, ROI_ALIGN_LEGACY = 0, 1
NMS_GE_ORIG_ASC, NMS_GT_SORTED_POS = 0, 1
ROI_ALIGN_RECORDS_READY, ROI_ALIGN_OVERWRITE = 1, 2

_c_int, _c_float, _c_void_p, _c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/mi_detectron_ops.h declaration by declaration
SIGNATURES = {
    "mi_abi_version": (_c_int, []),
    "mi_last_error": (ctypes.c_char_p, []),
    "mi_roi_align_forward": (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_float] + [_c_int] * 3 + [_c_void_p]),
    "mi_roi_align_forward_workspace_bytes": (_c_size_t, [_c_int]),
    "mi_roi_align_forward_ws": (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_float] + [_c_int] * 3
                                + [_c_void_p, _c_size_t, _c_void_p]),
    "mi_roi_align_backward": (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_float] + [_c_int] * 3 + [_c_void_p]),
    "mi_roi_align_backward_ws": (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_float] + [_c_int] * 3
                                 + [_c_void_p, _c_size_t, _c_int, _c_void_p]),
    "mi_roi_align_backward_overwrites": (_c_int, [_c_int] * 8),
    "mi_roi_align_forward_writes_records": (_c_int, [_c_int] * 8),
    "mi_roi_pool_forward": (_c_int, [_c_void_p] * 4 + [_c_int] * 7 + [_c_float, _c_void_p]),
    "mi_roi_pool_backward": (_c_int, [_c_void_p] * 4 + [_c_int] * 7 + [_c_float, _c_void_p]),
    "mi_roi_crop_forward": (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_void_p]),
    "mi_roi_crop_backward": (_c_int, [_c_void_p] * 4 + [_c_int] * 7 + [_c_void_p]),
    "mi_roi_crop_backward_workspace_bytes": (_c_size_t, [_c_int]),
    "mi_roi_crop_backward_ws": (_c_int, [_c_void_p] * 4 + [_c_int] * 7 + [_c_void_p, _c_size_t, _c_void_p]),
    "mi_nms_workspace_bytes": (_c_size_t, [_c_int]),
    "mi_nms": (_c_int, [_c_void_p, _c_int, _c_float, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "mi_nms_batched_workspace_bytes": (_c_size_t, [_c_int, _c_void_p]),
    "mi_nms_batched": (_c_int, [_c_int, _c_void_p, _c_void_p, _c_float, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_size_t,
                               _c_void_p]),
    "mi_nms_segmented_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "mi_nms_segmented": (_c_int, [_c_void_p, ctypes.c_longlong, ctypes.c_longlong, _c_void_p, ctypes.c_longlong,
                                 ctypes.c_longlong, _c_int, _c_int, _c_float, _c_float, _c_void_p, _c_void_p, _c_void_p,
                                 _c_void_p, _c_size_t, _c_void_p]),
    "mi_detection_select": (_c_int, [_c_void_p] * 5 + [_c_int] * 4 + [_c_void_p] * 4),
    "mi_topk_batched_workspace_bytes": (_c_size_t, [_c_int, _c_void_p, _c_void_p]),
    "mi_topk_batched": (_c_int, [_c_int] + [_c_void_p] * 6 + [_c_size_t, _c_void_p]),
    "mi_rpn_collect_candidates": (_c_int, [_c_int] + [_c_void_p] * 6 + [_c_int, _c_void_p, _c_void_p, _c_void_p]),
    "mi_fpn_level_index_from_restore": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "mi_rpn_collect_finish": (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_float, _c_float] + [_c_void_p] * 4),
    "mi_roi_align_fpn_supported": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "mi_keypoint_nms_oks": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, ctypes.c_double, _c_void_p, _c_void_p, _c_void_p]),
    "mi_box_voting": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_float, _c_int, _c_float,
                              _c_void_p, _c_void_p]),
    "mi_polys_to_masks_wrt_boxes": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int,
                                            _c_int, _c_void_p]),
    "mi_roi_align_backward_workspace_bytes": (_c_size_t, [_c_void_p, _c_int, _c_int]),
    "mi_roi_align_forward_fpn": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                         _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "mi_roi_align_forward_fpn_records": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                                 _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "mi_rpn_collect_finish_records": (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_float, _c_float] + [_c_void_p] * 4
                                      + [_c_int] * 6 + [_c_void_p, _c_size_t, _c_void_p]),
    "mi_roi_align_backward_fpn": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                          _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_int, _c_void_p]),
    "mi_rpn_decode_proposals": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                        _c_int, _c_int, ctypes.c_double, _c_float, ctypes.c_double, _c_void_p, _c_void_p,
                                        _c_void_p]),
    "mi_soft_nms": (_c_int, [_c_void_p, _c_int, _c_float, _c_float, _c_float, _c_int, _c_void_p, _c_void_p, _c_void_p,
                            _c_void_p]),
    "mi_soft_nms_segmented": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int,
                                      _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "mi_bbox_overlaps": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "mi_affine_channel_forward": (_c_int, [_c_void_p] * 5 + [_c_int] * 6 + [_c_void_p]),
    "mi_affine_channel_backward": (_c_int, [_c_void_p] * 5 + [_c_int] * 6 + [_c_void_p]),
    "mi_mask_paste_rle": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_void_p,
                                  _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "mi_keypoint_decode": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "mi_dbg_roi_align_timeline": (None, [_c_void_p]),
    "mi_dbg_reload_tuning": (None, []),
    "mi_dbg_copy_float4": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_void_p]),
}



class FpnLevels(ctypes.Structure):
    """mi_fpn_levels (include/mi_detectron_ops.h)."""
    _fields_ = [("num_levels", ctypes.c_int), ("features", ctypes.c_void_p * 4), ("grads", ctypes.c_void_p * 4),
                ("height", ctypes.c_int * 4), ("width", ctypes.c_int * 4), ("spatial_scale", ctypes.c_float * 4)]


_lib = None


class MiOpsError(RuntimeError):
    pass


def lib():
    """Load the HIP library; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
        # It must be in the process BEFORE our library is dlopen()ed, so that our NEEDED
        # libamdhip64.so.7 binds to the same runtime instance that owns torch's allocations and
        # streams; loading ours first would pull in /opt/rocm's copy as a second runtime (observed:
        # "no ROCm-capable device is detected" from our launches).
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise MiOpsError(
                "%s not found: build it with `python -m detectron_pytorch_amd.build` "
                "(or __graft_entry__.build()); there is no CPU fallback for these ops" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.mi_abi_version() != ABI_VERSION:
            raise MiOpsError("ABI version mismatch: library %d, binding %d -- rebuild with "
                             "`python -m detectron_pytorch_amd.build --force`" % (handle.mi_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc, what):
    if rc != MI_OK:
        msg = lib().mi_last_error().decode(errors="replace")
        raise MiOpsError("%s failed (code %d): %s" % (what, rc, msg))


def current_stream_handle(device):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name):
    """The reference raises NotImplementedError for CPU features (functions/roi_align.py:29-30)."""
    if not t.is_cuda:
        raise NotImplementedError("%s must be a GPU tensor: this op has no CPU implementation" % name)
