"""CPU tests of the C-ABI boundary: the HIP library builds (hipcc cross-compiles gfx950 without a
GPU), loads, exports exactly the symbols include/mi_detectron_ops.h declares, and rejects bad
arguments before touching the device.  No compute is launched here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi_detectron_ops.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for required in ["mi_roi_align_forward", "mi_roi_align_forward_ws", "mi_roi_align_forward_workspace_bytes",
                     "mi_roi_align_backward", "mi_roi_align_backward_ws", "mi_roi_align_backward_overwrites", "mi_roi_pool_forward", "mi_roi_pool_backward",
                     "mi_roi_crop_forward", "mi_roi_crop_backward", "mi_roi_crop_backward_ws", "mi_roi_crop_backward_workspace_bytes", "mi_nms", "mi_nms_workspace_bytes", "mi_nms_batched", "mi_nms_batched_workspace_bytes", "mi_soft_nms", "mi_rpn_decode_proposals", "mi_roi_align_fpn_supported", "mi_roi_align_forward_fpn", "mi_roi_align_backward_fpn", "mi_soft_nms_segmented", "mi_roi_align_forward_writes_records",
                     "mi_bbox_overlaps", "mi_last_error", "mi_abi_version"]:
        assert required in syms


def test_library_exports_every_declared_symbol(hip_lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", hip_lib_path]).decode()
    exported = set(re.findall(r" T (mi_[a-z0-9_]+)", out))
    assert set(declared_symbols()) == exported


def test_library_loads_and_binding_matches_header(hip_lib_path):
    from detectron_pytorch_amd import _lib

    handle = _lib.lib()
    assert handle.mi_abi_version() == _lib.ABI_VERSION
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    assert handle.mi_nms_workspace_bytes(0) >= 16
    # workspace grows with n * ceil(n/64) mask words
    assert handle.mi_nms_workspace_bytes(2000) > 2000 * 32 * 8


def test_bad_arguments_are_rejected_without_a_device(hip_lib_path):
    from detectron_pytorch_amd import _lib

    h = _lib.lib()
    null = ctypes.c_void_p(0)
    rc = h.mi_roi_align_forward(null, null, null, 1, 1, 4, 4, 1, 0, 7, 0.25, 2, 0, 0, null)
    assert rc == 1 and b"aligned size" in h.mi_last_error()
    rc = h.mi_roi_align_forward(null, null, null, 1, 1, 4, 4, 1, 7, 7, 0.25, 2, 9, 0, null)
    assert rc == 1 and b"variant" in h.mi_last_error()
    rc = h.mi_roi_align_forward(null, null, null, 1, 1, 4, 4, 1, 7, 7, 0.25, 2, 0, 0, null)
    assert rc == 1 and b"null" in h.mi_last_error()
    rc = h.mi_nms(null, -1, 0.5, 0, null, null, null, 0, null)
    assert rc == 1
    rc = h.mi_roi_crop_forward(null, null, null, 4, 1, 4, 4, 2, 7, 7, null)
    assert rc == 1 and b"RoIs-per-image" in h.mi_last_error()
    with pytest.raises(_lib.MiOpsError):
        _lib.check(rc, "mi_roi_crop_forward")


def test_cpu_tensors_raise_like_the_reference(hip_lib_path):
    import torch

    from detectron_pytorch_amd.roi_align import RoIAlignFunction
    from detectron_pytorch_amd.roi_pool import RoIPoolFunction

    with pytest.raises(NotImplementedError):  # roi_xfrom/roi_align/functions/roi_align.py:29-30
        RoIAlignFunction(7, 7, 0.25, 2)(torch.zeros(1, 2, 8, 8), torch.zeros(1, 5))
    with pytest.raises(NotImplementedError):
        RoIPoolFunction(7, 7, 0.25)(torch.zeros(1, 2, 8, 8), torch.zeros(1, 5))


def test_dropin_overlay_resolves_reference_import_paths(hip_lib_path):
    import importlib
    import sys

    overlay = os.path.join(ROOT, "detectron_pytorch_amd", "dropin", "lib")
    sys.path.insert(0, overlay)
    try:
        for mod, names in [("modeling.roi_xfrom.roi_align.functions.roi_align", ["RoIAlignFunction"]),
                           ("model.roi_align.functions.roi_align", ["RoIAlignFunction"]),
                           ("model.roi_pooling.functions.roi_pool", ["RoIPoolFunction"]),
                           ("model.roi_crop.functions.roi_crop", ["RoICropFunction"]),
                           ("model.nms.nms_gpu", ["nms_gpu"]), ("model.nms.nms_wrapper", ["nms"]),
                           ("utils.cython_nms", ["nms", "soft_nms"]), ("utils.cython_bbox", ["bbox_overlaps"])]:
            m = importlib.import_module(mod)
            for n in names:
                assert callable(getattr(m, n))
        # same constructor signatures as the reference call sites (model_builder.py:279,286,290)
        from modeling.roi_xfrom.roi_align.functions.roi_align import RoIAlignFunction
        from model.roi_align.functions.roi_align import RoIAlignFunction as Legacy
        RoIAlignFunction(7, 7, 0.25, 2)
        Legacy(7, 7, 0.25)
    finally:
        sys.path.remove(overlay)
        for k in [k for k in sys.modules if k.split(".")[0] in ("modeling", "model", "utils")]:
            del sys.modules[k]


def test_dropin_overlay_merges_with_the_reference_tree(tmp_path, monkeypatch):
    """With dropin/lib in front of the reference's lib/ on sys.path, modules the overlay does not provide must still be
    found in the reference tree, and the ones it provides must win."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    overlay = os.path.join(root, "detectron_pytorch_amd", "dropin", "lib")
    fake = tmp_path / "lib"
    for pkg in ("modeling", "utils", "core"):
        (fake / pkg).mkdir(parents=True)
        (fake / pkg / "__init__.py").write_text("")
    (fake / "modeling" / "model_builder.py").write_text("WHO = 'reference model_builder'\n")
    (fake / "modeling" / "generate_anchors.py").write_text("WHO = 'reference generate_anchors'\n")
    (fake / "utils" / "blob.py").write_text("WHO = 'reference blob'\n")
    for name in [m for m in sys.modules if m.split(".")[0] in ("modeling", "utils", "core", "model")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.setattr(sys, "path", [overlay, str(fake)] + sys.path)
    assert importlib.import_module("modeling.model_builder").WHO == "reference model_builder"
    assert importlib.import_module("utils.blob").WHO == "reference blob"
    anchors_mod = importlib.import_module("modeling.generate_anchors")
    assert not hasattr(anchors_mod, "WHO") and anchors_mod.generate_anchors().shape == (15, 4)
    assert importlib.import_module("utils.cython_nms").soft_nms.__module__ == "detectron_pytorch_amd.nms"
    for name in [m for m in sys.modules if m.split(".")[0] in ("modeling", "utils", "core", "model")]:
        monkeypatch.delitem(sys.modules, name, raising=False)


def test_deterministic_mode_asks_for_the_unplanned_backward(hip_lib_path):
    """torch.use_deterministic_algorithms(True): the autograd Functions size their scratch for the records only, which makes
    the backward run without list slices and their fp32 atomics (host-side sizing only: no launch here)."""
    import torch

    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    fwd = _lib.lib().mi_roi_align_forward_workspace_bytes(1024)
    assert _backward_workspace_bytes([(200, 336)], 2, 1024) > fwd
    torch.use_deterministic_algorithms(True)
    try:
        assert _backward_workspace_bytes([(200, 336)], 2, 1024) == fwd
    finally:
        torch.use_deterministic_algorithms(False)
