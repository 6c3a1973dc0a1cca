"""RoIAlign -- host-side mirror of the reference's autograd Functions, backed by mi_roi_align_*.

Reference interface kept (same names, argument order and behaviour):
  * lib/modeling/roi_xfrom/roi_align/functions/roi_align.py:7-48
        RoIAlignFunction(aligned_height, aligned_width, spatial_scale, sampling_ratio)(features, rois)
  * lib/model/roi_align/functions/roi_align.py:7-47   (legacy jwyang arithmetic, no sampling_ratio)
        RoIAlignFunction(aligned_height, aligned_width, spatial_scale)(features, rois)
  * modules RoIAlign / RoIAlignAvg / RoIAlignMax: roi_xfrom/roi_align/modules/roi_align.py:6-45 and
    model/roi_align/modules/roi_align.py:6-42

The reference uses legacy instance-style autograd Functions (config in __init__, tensors in
__call__), which PyTorch >= 1.3 rejects.  Each public name here is a small callable whose
__call__ invokes the modern static `_RoIAlign.apply(features, rois, ah, aw, scale, sr, variant)`;
that `.apply` signature is the frozen one.

Behaviour preserved: CPU `features` -> NotImplementedError (:29-30); no gradient w.r.t. rois
(backward returns (grad_input, None), :48); fp32 only; output [R, C, ah, aw] in the features' dtype.
New: features stored channels_last are consumed in place (layout flag of the C-ABI) instead of
being copied to NCHW.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.nn.modules.module import Module
from torch.nn.functional import avg_pool2d, max_pool2d

from . import _lib


def _layout_of(features):
    """Return (tensor usable by the kernels, layout flag)."""
    if features.is_contiguous():
        return features, _lib.LAYOUT_NCHW
    if features.dim() == 4 and features.is_contiguous(memory_format=torch.channels_last):
        return features, _lib.LAYOUT_NHWC
    return features.contiguous(), _lib.LAYOUT_NCHW


def _check_inputs(features, rois):
    _lib.require_cuda(features, "features")
    if features.dtype != torch.float32 or rois.dtype != torch.float32:
        raise TypeError("RoIAlign supports float32 features and rois only (as the reference)")
    if features.dim() != 4:
        raise ValueError("features must be [N, C, H, W]")
    if rois.dim() != 2 or rois.size(1) != 5:
        # the reference C glue returns 0 here and Python ignores it (roi_align_cuda.c:19-22)
        raise ValueError("rois must be [R, 5] (batch_index, x1, y1, x2, y2)")
    if rois.device != features.device:
        raise ValueError("rois and features must be on the same device")


def roi_align_forward(features, rois, aligned_height, aligned_width, spatial_scale, sampling_ratio,
                      variant=_lib.ROI_ALIGN_CAFFE2, return_workspace=False, want_backward=None):
    """Raw forward (no autograd): returns a new [R, C, ah, aw] tensor and, on request, the device scratch holding the
    per-RoI records a backward over the same rois can reuse -- None when the forward wrote none (shapes only the generic
    kernel serves).  want_backward (default: return_workspace): size the scratch for the
    planned backward, which also makes the forward write the records' backward block (2 us; an inference call skips it)."""
    if want_backward is None:
        want_backward = return_workspace
    _check_inputs(features, rois)
    features, layout = _layout_of(features)
    if variant == _lib.ROI_ALIGN_LEGACY and layout != _lib.LAYOUT_NCHW:
        features, layout = features.contiguous(), _lib.LAYOUT_NCHW
    rois = rois.contiguous()
    n, c, h, w = features.shape
    r = rois.size(0)
    # every element is written by the kernel: no zero fill (the reference's .zero_() at :23 is redundant)
    output = torch.empty((r, c, aligned_height, aligned_width), dtype=features.dtype, device=features.device)
    lib = _lib.lib()
    records = variant == _lib.ROI_ALIGN_CAFFE2 and bool(lib.mi_roi_align_forward_writes_records(
        c, h, w, r, int(aligned_height), int(aligned_width), int(variant), layout))
    # device scratch (a free-list pop of the caching allocator): the per-RoI records of the fast paths
    ws_bytes = lib.mi_roi_align_forward_workspace_bytes(r)
    if records and want_backward:
        ws_bytes = max(ws_bytes, _backward_workspace_bytes([(h, w)], n, r))  # a backward reuses this scratch
    workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=features.device)
    with torch.cuda.device(features.device):
        rc = lib.mi_roi_align_forward_ws(
            features.data_ptr(), rois.data_ptr(), output.data_ptr(), n, c, h, w, r,
            int(aligned_height), int(aligned_width), float(spatial_scale), int(sampling_ratio),
            int(variant), layout, workspace.data_ptr(), ws_bytes, _lib.current_stream_handle(features.device))
    _lib.check(rc, "mi_roi_align_forward_ws")
    return (output, workspace if records else None) if return_workspace else output


def _backward_workspace_bytes(sizes, batch, num_rois):
    """mi_roi_align_backward_workspace_bytes for gradient maps of the given (height, width)s: records + backward plan.

    Under `torch.use_deterministic_algorithms(True)` only the records' size is returned: the backward then runs unplanned --
    one workgroup sums a tile's whole RoI list in sweep order, no atomics, bit-reproducible from run to run -- where the
    planned backward adds the slices of a long list with fp32 atomics in whatever order they finish."""
    if torch.are_deterministic_algorithms_enabled():
        return int(_lib.lib().mi_roi_align_forward_workspace_bytes(int(num_rois)))
    t = _lib.FpnLevels()
    t.num_levels = len(sizes)
    for i, (h, w) in enumerate(sizes):
        t.height[i], t.width[i] = int(h), int(w)
    return int(_lib.lib().mi_roi_align_backward_workspace_bytes(ctypes.byref(t), int(batch), int(num_rois)))


def _aligned16(t):
    """The backward kernels fetch gradient blocks in 16-byte pieces: a contiguous view that starts in the middle of an
    allocation (possible for a slice along the channel / bin axes flattened by the caller) is copied once."""
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def roi_align_backward(grad_output, rois, feature_size, aligned_height, aligned_width, spatial_scale,
                       sampling_ratio, variant=_lib.ROI_ALIGN_CAFFE2, channels_last=False, workspace=None):
    """Raw backward: returns grad w.r.t. features, shape `feature_size` (zero-filled then accumulated,
    functions/roi_align.py:39-44)."""
    _lib.require_cuda(grad_output, "grad_output")
    grad_output = _aligned16(grad_output.contiguous())
    rois = rois.contiguous()
    n, c, h, w = feature_size
    fmt = torch.channels_last if (channels_last and variant == _lib.ROI_ALIGN_CAFFE2) else torch.contiguous_format
    layout = _lib.LAYOUT_NHWC if fmt is torch.channels_last else _lib.LAYOUT_NCHW
    lib = _lib.lib()
    r = rois.size(0)
    # The NCHW tile path writes every element itself; everywhere else the kernels accumulate into a zero-filled
    # buffer like the reference (functions/roi_align.py:39-44).
    overwrite = bool(lib.mi_roi_align_backward_overwrites(c, h, w, r, int(aligned_height), int(aligned_width),
                                                          int(variant), layout))
    grad_input = torch.empty((n, c, h, w), dtype=grad_output.dtype, device=grad_output.device, memory_format=fmt)
    if not overwrite:
        grad_input.zero_()
    ws_bytes = lib.mi_roi_align_forward_workspace_bytes(r)
    # `workspace`: the scratch of the forward over the same rois (records are reused); else a fresh one
    ready = workspace is not None and workspace.numel() >= ws_bytes and workspace.device == grad_output.device
    if not ready:
        workspace = torch.empty(_backward_workspace_bytes([(h, w)], n, r) if variant == _lib.ROI_ALIGN_CAFFE2 else ws_bytes,
                                dtype=torch.uint8, device=grad_output.device)
    flags = (_lib.ROI_ALIGN_RECORDS_READY if ready else 0) | (_lib.ROI_ALIGN_OVERWRITE if overwrite else 0)
    with torch.cuda.device(grad_output.device):
        rc = lib.mi_roi_align_backward_ws(
            grad_output.data_ptr(), rois.data_ptr(), grad_input.data_ptr(), n, c, h, w, r,
            int(aligned_height), int(aligned_width), float(spatial_scale), int(sampling_ratio),
            int(variant), layout, workspace.data_ptr(), workspace.numel(), flags,
            _lib.current_stream_handle(grad_output.device))
    _lib.check(rc, "mi_roi_align_backward_ws")
    return grad_input


class _RoIAlign(Function):
    """Modern static autograd Function; `.apply(features, rois, ah, aw, scale, sampling_ratio, variant)`."""

    @staticmethod
    def forward(ctx, features, rois, aligned_height, aligned_width, spatial_scale, sampling_ratio, variant):
        ctx.cfg = (int(aligned_height), int(aligned_width), float(spatial_scale), int(sampling_ratio), int(variant))
        ctx.feature_size = tuple(features.shape)
        ctx.channels_last = (features.dim() == 4 and not features.is_contiguous()
                             and features.is_contiguous(memory_format=torch.channels_last))
        # records a forward left in its scratch (channels-last features) serve the backward over the same rois
        output, workspace = roi_align_forward(features, rois, *ctx.cfg, return_workspace=True,
                                              want_backward=bool(ctx.needs_input_grad[0]))
        ctx.save_for_backward(rois, workspace if workspace is not None else rois.new_empty(0))
        return output

    @staticmethod
    def backward(ctx, grad_output):
        rois, workspace = ctx.saved_tensors
        ah, aw, scale, sr, variant = ctx.cfg
        grad_input = roi_align_backward(grad_output, rois, ctx.feature_size, ah, aw, scale, sr, variant,
                                        channels_last=ctx.channels_last,
                                        workspace=workspace if workspace.numel() > 0 else None)
        return grad_input, None, None, None, None, None, None


def _fpn_table(tensors, scales, grads=False):
    t = _lib.FpnLevels()
    t.num_levels = len(tensors)
    for i, (x, sc) in enumerate(zip(tensors, scales)):
        (t.grads if grads else t.features)[i] = x.data_ptr()
        t.height[i], t.width[i], t.spatial_scale[i] = int(x.size(2)), int(x.size(3)), float(sc)
    return t


def _common_layout(features):
    """LAYOUT_NCHW / LAYOUT_NHWC when every map is stored that way (dense), else None."""
    if all(f.is_contiguous() for f in features):
        return _lib.LAYOUT_NCHW
    if all(f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) for f in features):
        return _lib.LAYOUT_NHWC
    return None


def roi_align_fpn_supported(features, num_rois, aligned_height, aligned_width):
    """True when mi_roi_align_forward_fpn / _backward_fpn serve these maps in one call (at most four levels, same
    batch, channels and storage layout -- all NCHW or all channels_last); otherwise the caller loops over the levels."""
    if not (1 <= len(features) <= 4) or num_rois <= 0:
        return False
    f0 = features[0]
    for f in features:
        if (not f.is_cuda or f.dtype != torch.float32 or f.dim() != 4 or f.size(0) != f0.size(0)
                or f.size(1) != f0.size(1) or f.device != f0.device):
            return False
    layout = _common_layout(features)
    if layout is None:
        return False
    table = _fpn_table(features, [1.0] * len(features))
    return bool(_lib.lib().mi_roi_align_fpn_supported(ctypes.byref(table), int(f0.size(1)), int(num_rois),
                                                      int(aligned_height), int(aligned_width), layout))


class _RoIAlignFPN(Function):
    """RoIAlign over all FPN levels in one call: `.apply(rois, roi_levels, ah, aw, sampling_ratio, scales, *features)`.
    roi_levels[i] = index into `features` of the map RoI i is pooled from; the output is in the order of `rois`."""

    @staticmethod
    def forward(ctx, rois, roi_levels, aligned_height, aligned_width, sampling_ratio, scales, *features):
        lib = _lib.lib()
        rois = rois.contiguous()
        roi_levels = roi_levels.to(dtype=torch.int32).contiguous()
        n, c = features[0].size(0), features[0].size(1)
        r = rois.size(0)
        output = torch.empty((r, c, aligned_height, aligned_width), dtype=torch.float32, device=rois.device)
        ws_bytes = lib.mi_roi_align_forward_workspace_bytes(r)
        if any(ctx.needs_input_grad[6:]):  # room for the planned backward; the forward then writes its tables too
            ws_bytes = max(ws_bytes, _backward_workspace_bytes([(f.size(2), f.size(3)) for f in features], n, r))
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=rois.device)
        table = _fpn_table(features, scales)
        layout = _common_layout(features)
        if layout is None:
            raise ValueError("the FPN maps must share one storage layout (all contiguous or all channels_last)")
        with torch.cuda.device(rois.device):
            rc = lib.mi_roi_align_forward_fpn(ctypes.byref(table), rois.data_ptr(), roi_levels.data_ptr(),
                                              output.data_ptr(), n, c, r, int(aligned_height), int(aligned_width),
                                              int(sampling_ratio), layout, workspace.data_ptr(), ws_bytes,
                                              _lib.current_stream_handle(rois.device))
        _lib.check(rc, "mi_roi_align_forward_fpn")
        ctx.records_ready = True   # the fused forward leaves the records of its rois in the workspace
        ctx.cfg = (int(aligned_height), int(aligned_width), int(sampling_ratio), tuple(float(s) for s in scales))
        ctx.shapes = [tuple(f.shape) for f in features]
        ctx.layout = layout
        ctx.save_for_backward(rois, roi_levels, workspace)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.lib()
        rois, roi_levels, workspace = ctx.saved_tensors
        ah, aw, sr, scales = ctx.cfg
        grad_output = _aligned16(grad_output.contiguous())
        fmt = torch.channels_last if ctx.layout == _lib.LAYOUT_NHWC else torch.contiguous_format
        grads = [torch.empty(shape, dtype=torch.float32, device=grad_output.device, memory_format=fmt)
                 for shape in ctx.shapes]
        n, c = ctx.shapes[0][0], ctx.shapes[0][1]
        table = _fpn_table(grads, scales, grads=True)
        with torch.cuda.device(grad_output.device):  # every element of every level is written: no zero fill
            rc = lib.mi_roi_align_backward_fpn(ctypes.byref(table), grad_output.data_ptr(), rois.data_ptr(),
                                               roi_levels.data_ptr(), n, c, rois.size(0), ah, aw, sr, ctx.layout,
                                               workspace.data_ptr(), workspace.numel(),
                                               (_lib.ROI_ALIGN_RECORDS_READY if ctx.records_ready else 0)
                                               | _lib.ROI_ALIGN_OVERWRITE,
                                               _lib.current_stream_handle(grad_output.device))
        _lib.check(rc, "mi_roi_align_backward_fpn")
        return (None, None, None, None, None, None) + tuple(grads)


class PreparedRecords(object):
    """The RoIAlign records of a RoI blob, written by the stage that PRODUCES the blob (mi_rpn_collect_finish_records:
    fpn_proposals.generate_and_collect(..., records=this)) instead of by the first launch of the pooling call -- the producer
    holds every RoI in a launch of its own, so the pooling call that follows starts with its gather kernel.

    Built before the proposals exist, from what the pooling call will be: the pyramid (`features`, coarsest map first, and
    their `scales`), the pooled size, the sampling ratio and the number of rows of the (static) RoI blob.  `matches()` is
    what roi_align_fpn checks before it trusts the records: same maps (addresses), same geometry, and the very tensor the
    producer wrote the RoIs into.  Inference only (no backward block is asked for)."""

    def __init__(self, features, scales, aligned_height, aligned_width, sampling_ratio, num_rois):
        self.features = list(features)
        self.scales = tuple(float(s) for s in scales)
        self.cfg = (int(aligned_height), int(aligned_width), int(sampling_ratio))
        self.num_rois = int(num_rois)
        self.layout = _common_layout(self.features)
        self.batch, self.channels = int(features[0].size(0)), int(features[0].size(1))
        self.supported = self.layout is not None and roi_align_fpn_supported(self.features, self.num_rois, *self.cfg[:2])
        self.rois = None          # set by the producer: the blob these records describe
        self.workspace = None
        if self.supported:
            self.table = _fpn_table(self.features, self.scales)
            self.workspace = torch.empty(int(_lib.lib().mi_roi_align_forward_workspace_bytes(self.num_rois)),
                                         dtype=torch.uint8, device=features[0].device)

    def matches(self, features, scales, rois, aligned_height, aligned_width, sampling_ratio):
        return (self.supported and self.rois is not None and rois is self.rois and rois.size(0) == self.num_rois
                and (int(aligned_height), int(aligned_width), int(sampling_ratio)) == self.cfg
                and len(features) == len(self.features)
                and all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(features, self.features))
                and tuple(float(s) for s in scales) == self.scales and _common_layout(features) == self.layout)


def roi_align_fpn(features, scales, rois, roi_levels, aligned_height, aligned_width, sampling_ratio, prepared=None):
    """Pool `rois` [R,5] from the FPN maps `features` (list, any order; `scales` alike) in one call; roi_levels [R]
    (int tensor on the device) indexes into `features`.  Differentiable w.r.t. every map.
    `prepared`: a PreparedRecords whose producer already wrote the records of exactly this call (inference): the records
    launch is skipped (mi_roi_align_forward_fpn_records); anything that does not match is ignored."""
    for f in features:
        _check_inputs(f, rois)
    if (prepared is not None and not torch.is_grad_enabled()
            and prepared.matches(features, scales, rois, aligned_height, aligned_width, sampling_ratio)):
        lib = _lib.lib()
        roi_levels = roi_levels.to(dtype=torch.int32).contiguous()
        output = torch.empty((rois.size(0), prepared.channels, int(aligned_height), int(aligned_width)), dtype=torch.float32,
                             device=rois.device)
        with torch.cuda.device(rois.device):
            rc = lib.mi_roi_align_forward_fpn_records(
                ctypes.byref(prepared.table), rois.data_ptr(), roi_levels.data_ptr(), output.data_ptr(), prepared.batch,
                prepared.channels, rois.size(0), int(aligned_height), int(aligned_width), int(sampling_ratio), prepared.layout,
                prepared.workspace.data_ptr(), prepared.workspace.numel(), _lib.current_stream_handle(rois.device))
        _lib.check(rc, "mi_roi_align_forward_fpn_records")
        return output
    return _RoIAlignFPN.apply(rois, roi_levels, int(aligned_height), int(aligned_width), int(sampling_ratio),
                              tuple(scales), *features)


class RoIAlignFunction(object):
    """Drop-in for modeling.roi_xfrom.roi_align.functions.roi_align.RoIAlignFunction (Caffe2 semantics).

    Call site kept verbatim: `RoIAlignFunction(resolution, resolution, sc, sampling_ratio)(bl_in, rois)`
    (lib/modeling/model_builder.py:290-291, 321-322).
    """

    def __init__(self, aligned_height, aligned_width, spatial_scale, sampling_ratio):
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)

    def __call__(self, features, rois):
        return _RoIAlign.apply(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale,
                               self.sampling_ratio, _lib.ROI_ALIGN_CAFFE2)


class LegacyRoIAlignFunction(object):
    """Drop-in for model.roi_align.functions.roi_align.RoIAlignFunction (legacy jwyang semantics)."""

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def __call__(self, features, rois):
        return _RoIAlign.apply(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale, 0,
                               _lib.ROI_ALIGN_LEGACY)


# ---- modules: roi_xfrom/roi_align/modules/roi_align.py:6-45 ---------------------------------
class RoIAlign(Module):
    def __init__(self, aligned_height, aligned_width, spatial_scale, sampling_ratio):
        super(RoIAlign, self).__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)

    def forward(self, features, rois):
        return RoIAlignFunction(self.aligned_height, self.aligned_width, self.spatial_scale,
                                self.sampling_ratio)(features, rois)


class RoIAlignAvg(RoIAlign):
    def forward(self, features, rois):
        x = RoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale,
                             self.sampling_ratio)(features, rois)
        return avg_pool2d(x, kernel_size=2, stride=1)


class RoIAlignMax(RoIAlign):
    def forward(self, features, rois):
        x = RoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale,
                             self.sampling_ratio)(features, rois)
        return max_pool2d(x, kernel_size=2, stride=1)


# ---- legacy modules: model/roi_align/modules/roi_align.py:6-42 -------------------------------
class LegacyRoIAlign(Module):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super(LegacyRoIAlign, self).__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return LegacyRoIAlignFunction(self.aligned_height, self.aligned_width, self.spatial_scale)(features, rois)


class LegacyRoIAlignAvg(LegacyRoIAlign):
    def forward(self, features, rois):
        x = LegacyRoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale)(features, rois)
        return avg_pool2d(x, kernel_size=2, stride=1)


class LegacyRoIAlignMax(LegacyRoIAlign):
    def forward(self, features, rois):
        x = LegacyRoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale)(features, rois)
        return max_pool2d(x, kernel_size=2, stride=1)
