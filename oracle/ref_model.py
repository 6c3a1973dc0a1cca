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
import contextlib
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
    # the reference's top-level package names are generic ('utils', 'nn', 'core', ...): anything another test left under
    # them in sys.modules (e.g. the drop-in overlay's 'utils') would shadow the reference's files
    for name in list(sys.modules):
        top = name.split(".")[0]
        if top in ("utils", "modeling", "model", "core", "nn", "datasets", "roi_data"):
            f = getattr(sys.modules[name], "__file__", None) or ""
            if not f.startswith(lib):
                del sys.modules[name]
    if lib in sys.path:
        sys.path.remove(lib)
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


@contextlib.contextmanager
def pycocotools_polygon_calls():
    """While active, the stand-in `pycocotools.mask` answers the two calls utils/segms.py:114-115 (and :66-67) makes for
    polygons -- frPyObjects, decode -- with the restatement of maskApi.c in oracle/mask_targets.py; everything else of the
    reference's polygon handling then runs from its own source."""
    load()
    import utils.segms as segm_utils

    from . import mask_targets as oracle_segms

    mu = segm_utils.mask_util
    keep = (mu.__dict__.get("frPyObjects"), mu.__dict__.get("decode"))
    mu.frPyObjects = lambda polys, h, w: (list(polys), h, w)
    mu.decode = lambda rle: oracle_segms.fr_poly_decode(rle[0], rle[1], rle[2])
    try:
        yield
    finally:
        for k, v in zip(("frPyObjects", "decode"), keep):
            if v is None:
                mu.__dict__.pop(k, None)
            else:
                setattr(mu, k, v)


def mask_rcnn_blobs_from_polygons(labels_int32, sampled_boxes, segms, gt_classes, im_scale=1.0, batch_idx=0, resolution=28):
    """The reference's OWN `add_mask_rcnn_blobs` (roi_data/mask_rcnn.py:34-107) and `utils/segms.py` executed on a roidb
    entry with polygon `segms` (pycocotools_polygon_calls() active).  Returns the blobs dict (mask_rois,
    roi_has_mask_int32, masks_int32 -- class-agnostic [n_fg, M * M]) and the boxes enclosing the polygons (:44)."""
    import numpy as np

    cfg = load()
    import roi_data.mask_rcnn as mrcnn
    import utils.segms as segm_utils

    keep = (cfg.MRCNN.RESOLUTION, cfg.MRCNN.CLS_SPECIFIC_MASK)
    cfg.MRCNN.RESOLUTION, cfg.MRCNN.CLS_SPECIFIC_MASK = resolution, False
    try:
        with pycocotools_polygon_calls():
            entry = {"gt_classes": np.asarray(gt_classes, dtype=np.int32), "is_crowd": np.zeros(len(segms), dtype=bool),
                     "segms": segms}
            blobs = {"labels_int32": np.asarray(labels_int32, dtype=np.int32)}
            mrcnn.add_mask_rcnn_blobs(blobs, np.array(sampled_boxes, dtype=np.float32), entry, im_scale, batch_idx)
            blobs["boxes_from_polys"] = segm_utils.polys_to_boxes(segms)
    finally:
        cfg.MRCNN.RESOLUTION, cfg.MRCNN.CLS_SPECIFIC_MASK = keep
    return blobs


# ---- executing the reference's training forward on the CPU --------------------------------------------------------------
def roidb_entry(height, width, boxes, classes, num_classes, keypoints=None, segms=None):
    """A ground-truth-only roidb entry as datasets/json_dataset.py:178-262 builds it (rectangular polygon masks unless
    `segms` -- per instance a list of polygons -- is given)."""
    import numpy as np
    import scipy.sparse

    boxes = np.asarray(boxes, dtype=np.float32)
    classes = np.asarray(classes, dtype=np.int32)
    n = boxes.shape[0]
    ov = np.zeros((n, num_classes), dtype=np.float32)
    ov[np.arange(n), classes] = 1.0
    if segms is None:
        segms = [[[float(b[0]), float(b[1]), float(b[2]), float(b[1]), float(b[2]), float(b[3]), float(b[0]), float(b[3])]]
                 for b in boxes]
    entry = dict(height=height, width=width, flipped=False, boxes=boxes, segms=segms,
                 seg_areas=((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).astype(np.float32),
                 gt_classes=classes, gt_overlaps=scipy.sparse.csr_matrix(ov), is_crowd=np.zeros(n, dtype=bool),
                 box_to_gt_ind_map=np.arange(n, dtype=np.int32))
    if keypoints is not None:          # json_dataset.py:264-272: [n, 3, K] int32 (x, y, visibility)
        entry["gt_keypoints"] = np.asarray(keypoints, dtype=np.int32)
        entry["has_visible_keypoints"] = bool((entry["gt_keypoints"][:, 2, :] > 0).any())
    return entry


def rpn_blobs(entries, im_scales, seed):
    """roi_data/rpn.py:40-113 executed from source: (blobs dict incl. 'im_info' and the minimal 'roidb')."""
    import numpy as np

    load()
    import roi_data.rpn as rpn

    blobs = {k: [] for k in rpn.get_rpn_blob_names(is_training=True)}
    np.random.seed(seed)
    rpn.add_rpn_blobs(blobs, im_scales, entries)
    return blobs


def train_forward(model, data, blobs, priority, rasterizer):
    """`Generalized_RCNN.forward` in training mode on CPU tensors, with the two things that cannot run here replaced:
    `npr.choice(inds, size, replace=False)` in roi_data/fast_rcnn.py:146,160 draws "the first `size` of the permutation
    given by `priority`" (global candidate numbering: all gt boxes, then the collected proposals in collect order), and
    pycocotools' polygon rasteriser is `rasterizer(polygons, box, M)` (None: the reference's own polys_to_mask_wrt_box
    with pycocotools_polygon_calls() active -- polygon `segms` in the roidb entries).
    Returns (return_dict, captured) with captured['rois'] = the collected proposals and captured['blobs'] = the
    labelled RoI blobs (numpy)."""
    import pickle

    import numpy as np
    import torch

    load()
    import datasets.json_dataset as json_dataset
    import roi_data.fast_rcnn as frcn
    import utils.segms as segm_utils

    captured, state = {}, {}
    minimal = blobs["roidb"]
    n_gt = [int((e["gt_classes"] > 0).sum()) for e in minimal]
    gt_off = np.concatenate([[0], np.cumsum(n_gt)])
    orig_add, orig_sample, orig_npr, orig_rast = (json_dataset.add_proposals, frcn._sample_rois, frcn.npr,
                                                  segm_utils.polys_to_mask_wrt_box)
    orig_add_blobs = frcn.add_fast_rcnn_blobs

    def add_proposals(roidb, rois, scales, crowd_thresh):
        captured["rois"] = rois.copy()
        state["pos"] = [np.where(rois[:, 0] == i)[0] for i in range(len(roidb))]
        return orig_add(roidb, rois, scales, crowd_thresh)

    def sample_rois(entry, im_scale, batch_idx):
        state["im"] = batch_idx
        return orig_sample(entry, im_scale, batch_idx)

    class Npr(object):
        @staticmethod
        def choice(a, size=None, replace=True):
            assert replace is False
            i = state["im"]
            a = np.asarray(a)
            glob = np.where(a < n_gt[i], gt_off[i] + a, gt_off[-1] + state["pos"][i][np.maximum(a - n_gt[i], 0)])
            return a[np.argsort(priority[glob], kind="stable")[:int(size)]]

    def add_blobs(b, im_scales, roidb):
        out = orig_add_blobs(b, im_scales, roidb)
        captured["blobs"] = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}
        return out

    json_dataset.add_proposals, frcn._sample_rois, frcn.npr = add_proposals, sample_rois, Npr
    segm_utils.polys_to_mask_wrt_box, frcn.add_fast_rcnn_blobs = rasterizer or orig_rast, add_blobs
    polygon_calls = pycocotools_polygon_calls() if rasterizer is None else contextlib.nullcontext()
    # roi_data/keypoint_rcnn.py:52-54 calls np.random.choice directly: same permutation rule
    import roi_data.keypoint_rcnn as kprcnn

    class _Random(object):
        choice = staticmethod(Npr.choice)

    class _NumpyProxy(object):
        random = _Random()

        def __getattr__(self, name):
            return getattr(np, name)

    orig_kp_np, kprcnn.np = kprcnn.np, _NumpyProxy()
    try:
        roidb_in = [np.frombuffer(pickle.dumps([e]), dtype=np.uint8).astype(np.float32) for e in minimal]   # blob.py:165-169
        kwargs = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in blobs.items() if k.startswith("rpn_")}
        with polygon_calls:
            ret = model(data, torch.from_numpy(blobs["im_info"]), roidb_in, **kwargs)
    finally:
        kprcnn.np = orig_kp_np
        json_dataset.add_proposals, frcn._sample_rois, frcn.npr = orig_add, orig_sample, orig_npr
        segm_utils.polys_to_mask_wrt_box, frcn.add_fast_rcnn_blobs = orig_rast, orig_add_blobs
    return ret, captured
