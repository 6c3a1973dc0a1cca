"""TEST INFRASTRUCTURE -- imports the REFERENCE's own Python model code (lib/modeling, lib/roi_data, lib/core/config,
...) from /root/reference so that tests and fixture generators can execute it on the CPU.  Nothing here is shipped or
measured; the product package never imports this module.  Only usable where /root/reference exists (this container,
not the GPU box): `available()` says so and callers skip.

The reference targets PyTorch 0.3/0.4, Python 3.6, numpy 1.x and compiled cffi / Cython / pycocotools extensions.  What
is shimmed, and only that (no arithmetic of the reference is replaced):
  * removed PyTorch internals its DataParallel imports (`torch._six`, `dataloader.numpy_type_map`) -- dummies, that code
    path is never run;
  * `yaml.load` without a Loader (config.py:1037) -> SafeLoader;
  * `np.float` / `np.int` aliases removed in numpy 1.24 (generate_anchors.py:63-65) -> float / int;
  * the compiled operators: `utils.cython_nms` / `utils.cython_bbox` are bound to the builds of the reference's own .pyx
    in oracle/_ref (oracle/build_ref.py); `RoIAlignFunction` is bound to an autograd Function around the reference's own
    .cu kernels compiled for the host (oracle/_ref/libref_roi_align.so); RoIPool / RoICrop are not used by the FPN yamls;
  * pycocotools / cv2 are absent: modules that import them get empty stand-ins; `segm_utils.polys_to_mask_wrt_box`
    (needs pycocotools' frPyObjects) is bound to the rectangle rasteriser the synthetic data uses;
  * `.cuda(device_id)` on CPU tensors (model_builder.py:276,302; *_heads.py losses) -> identity, `get_device()` -> 0.
"""
import collections.abc
import os
import sys
import types

REFERENCE = "/root/reference"
_state = {}


def available():
    from . import ref

    return os.path.isdir(os.path.join(REFERENCE, "lib", "modeling")) and ref.available()


def _stub(name, **attrs):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
    for k, v in attrs.items():
        setattr(sys.modules[name], k, v)
    return sys.modules[name]


def load():
    """Make `import core.config`, `import modeling.model_builder` ... resolve to the reference's files; returns the
    reference's global `cfg`."""
    if "cfg" in _state:
        return _state["cfg"]
    import numpy as np
    import torch
    import yaml

    from . import ref

    lib = os.path.join(REFERENCE, "lib")
    if lib not in sys.path:
        sys.path.insert(0, lib)
    six = types.ModuleType("torch._six")
    six.string_classes, six.int_classes, six.container_abcs = (str,), (int,), collections.abc
    sys.modules["torch._six"] = six
    import torch.utils.data.dataloader as dl

    if not hasattr(dl, "numpy_type_map"):
        dl.numpy_type_map = {}
    if not getattr(yaml.load, "_mi_patched", False):
        orig = yaml.load

        def load_yaml(stream, Loader=None):
            return orig(stream, Loader=Loader or yaml.SafeLoader)

        load_yaml._mi_patched = True
        yaml.load = load_yaml
    for alias, typ in (("float", float), ("int", int), ("bool", bool)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)

    # compiled operators of the reference, built from its own sources for the host
    sys.modules["utils.cython_nms"] = ref._mod("cython_nms")
    sys.modules["utils.cython_bbox"] = ref._mod("cython_bbox")

    class _RoIAlign(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, rois, ah, aw, scale, sr):
            ctx.cfg = (ah, aw, scale, sr, tuple(features.shape))
            ctx.save_for_backward(rois)
            out = ref.roi_align_forward(features.detach().numpy(), rois.detach().numpy(), ah, aw, scale, sr)
            return torch.from_numpy(out)

        @staticmethod
        def backward(ctx, grad):
            ah, aw, scale, sr, shape = ctx.cfg
            (rois,) = ctx.saved_tensors
            g = ref.roi_align_backward(grad.contiguous().numpy(), rois.numpy(), shape, scale, sr)
            return torch.from_numpy(g), None, None, None, None, None

    class RoIAlignFunction(object):   # functions/roi_align.py:7-48 call convention around the host build of its kernels
        def __init__(self, aligned_height, aligned_width, spatial_scale, sampling_ratio):
            self.cfg = (int(aligned_height), int(aligned_width), float(spatial_scale), int(sampling_ratio))

        def __call__(self, features, rois):
            return _RoIAlign.apply(features, rois, *self.cfg)

    _stub("modeling.roi_xfrom.roi_align.functions.roi_align", RoIAlignFunction=RoIAlignFunction)
    _stub("model.roi_pooling.functions.roi_pool", RoIPoolFunction=object)
    _stub("model.roi_crop.functions.roi_crop", RoICropFunction=object)
    _stub("pycocotools.mask")
    _stub("pycocotools.coco", COCO=object)
    _stub("pycocotools.cocoeval", COCOeval=object)
    _stub("cv2")
    # the stubs above created empty 'model' / 'modeling' packages: point them at the reference's directories so that the
    # real sub-modules (modeling.FPN, modeling.model_builder, ...) still import from source
    sys.modules["modeling"].__path__ = [os.path.join(lib, "modeling")]
    sys.modules["model"].__path__ = [os.path.join(lib, "model")]
    sys.modules["modeling.roi_xfrom"].__path__ = [os.path.join(lib, "modeling", "roi_xfrom")]

    if not getattr(torch.Tensor.cuda, "_mi_patched", False):
        orig_cuda = torch.Tensor.cuda

        def cuda(self, *a, **k):
            return self if not torch.cuda.is_available() else orig_cuda(self, *a, **k)

        cuda._mi_patched = True
        torch.Tensor.cuda = cuda
        orig_get_device = torch.Tensor.get_device
        torch.Tensor.get_device = lambda self: 0 if not self.is_cuda else orig_get_device(self)

    from core.config import cfg

    _state["cfg"] = cfg
    return cfg


def configure(yaml_relpath, **overrides):
    """Reset-free configuration helper: merge a reference yaml (path relative to /root/reference) and the overrides
    (dotted keys) into the reference's global cfg.  Call once per process."""
    cfg = load()
    from core.config import assert_and_infer_cfg, merge_cfg_from_file, merge_cfg_from_list

    merge_cfg_from_file(os.path.join(REFERENCE, yaml_relpath))
    flat = []
    for k, v in overrides.items():
        flat += [k.replace("__", "."), v]
    if flat:
        merge_cfg_from_list(flat)
    assert_and_infer_cfg(make_immutable=False)
    return cfg


def build_model(seed=None):
    """`Generalized_RCNN()` of the reference under the current cfg (optionally seeded like tools/train_net_step.py does
    through cfg.RNG_SEED)."""
    import torch

    load()
    import modeling.model_builder as mb

    if seed is not None:
        torch.manual_seed(seed)
    return mb.Generalized_RCNN()


def bind_rect_rasterizer(fn):
    """Replace `utils.segms.polys_to_mask_wrt_box` (pycocotools) by `fn(polygons, box, M)` -- see the header."""
    load()
    import utils.segms as segm_utils

    segm_utils.polys_to_mask_wrt_box = fn
