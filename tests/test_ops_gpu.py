"""GPU parity tests (-m gpu): the HIP kernels, called through the C-ABI via the host-side mirror of the
reference's Functions, against the CPU oracle and the committed golden vectors.

Bars (BASELINE.json north_star): NMS kept indices bit-exact; RoIAlign features / gradients within 1e-4.
Integer / index outputs (argmax, kept indices) bit-exact.  Forward ops follow the reference's fp32
operation order with FMA contraction off, so they are checked bit-exactly too where the summation
order is deterministic; backward ops use fp32 atomics (order unspecified in the reference as well)
and are checked to 1e-4.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden
from detectron_pytorch_amd import synthetic as syn

pytestmark = pytest.mark.gpu

ATOL = 1e-4  # north_star tolerance for RoIAlign features / gradients
RTOL = 1e-4
# The NCHW fast path of RoIAlign forward reads the reference's taps with the reference's weights but sums them
# separably with FMAs (roi_align_records.hip): fp32 rounding differences only.  Bar used below: 1e-5 (10x inside
# the contract); the generic direct kernel keeps the reference operation order and is checked bit-exactly.
FAST_ATOL = 1e-5


def dev():
    return torch.device("cuda", 0)


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def assert_close(actual, expected, what, atol=ATOL, rtol=RTOL):
    actual = actual.detach().cpu().numpy() if isinstance(actual, torch.Tensor) else actual
    err = np.abs(actual.astype(np.float64) - expected.astype(np.float64))
    tol = atol + rtol * np.abs(expected)
    assert (err <= tol).all(), "%s: max abs err %.3e (tol %.1e)" % (what, err.max(), atol)


def assert_fwd(actual, expected, what, exact):
    """exact: the direct kernel (reference operation order); otherwise the separable-FMA fast path."""
    actual = actual.detach().cpu().numpy() if isinstance(actual, torch.Tensor) else actual
    if exact:
        assert np.array_equal(actual, expected), "%s: forward of the direct kernel is expected to be bit-exact" % what
    else:
        assert_close(actual, expected, what, atol=FAST_ATOL, rtol=FAST_ATOL)


@pytest.fixture
def tuning_env(monkeypatch):
    """Set MI_ROI_ALIGN_* variables for one test.  The library reads them once and never again on the launch path, so a
    change is made visible through the explicit debug entry `mi_dbg_reload_tuning` (and undone the same way)."""
    from detectron_pytorch_amd import _lib

    def set_env(**env):
        for k, v in env.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        _lib.lib().mi_dbg_reload_tuning()

    yield set_env
    monkeypatch.undo()
    _lib.lib().mi_dbg_reload_tuning()


@pytest.fixture(params=["stream", "stream_unplanned", "stream_sliced", "direct"])
def roi_align_impl(request, tuning_env):
    """Run a test against the RoIAlign implementations behind mi_roi_align_*: the default fast paths ("stream": the
    record-driven forward, the planned backward with list slices of 32 RoIs; "stream_unplanned": MI_ROI_ALIGN_BWD_SLICE=0,
    one workgroup walks a tile's whole list; "stream_sliced": slices of 2 RoIs, so that nearly every tile is summed by
    several workgroups with atomics) and the generic direct kernels (MI_ROI_ALIGN_IMPL=direct)."""
    stream = request.param.startswith("stream")
    tuning_env(MI_ROI_ALIGN_IMPL=None if stream else request.param,
               MI_ROI_ALIGN_BWD_SLICE={"stream_unplanned": 0, "stream_sliced": 2}.get(request.param))
    return request.param


def _fwd_is_exact(impl, channels):
    """The direct kernels keep the reference's operation order; the fast paths (FMA, separable) decline channel counts
    that are no multiple of their channel tile (32 for the record-driven kernels, 8 for the records-free forward that
    takes their place) and fall back to them."""
    return impl == "direct" or channels % 8 != 0


def test_extension_is_loaded_not_a_fallback(hip_lib_path):
    from detectron_pytorch_amd import _lib

    assert _lib.lib().mi_abi_version() == _lib.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libmi_detectron_ops.so" in maps


# ---- RoIAlign (Caffe2 semantics) ---------------------------------------------------------------
def _roi_align_gpu(feat, rois, res, scale, sr, gtop=None, channels_last=False):
    from detectron_pytorch_amd.roi_align import RoIAlignFunction

    f = to_dev(feat).requires_grad_(True)
    if channels_last:
        f = f.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = RoIAlignFunction(res, res, scale, sr)(f, to_dev(rois))
    grad = None
    if gtop is not None:
        out.backward(to_dev(gtop))
        grad = f.grad
    return out, grad


@pytest.mark.parametrize("res,sr,channels,channels_last", [(16, 2, 64, False), (10, 0, 32, False), (3, 2, 96, False),
                                                         (16, 2, 64, True), (21, 2, 32, False)])
def test_roi_align_backward_pooled_sizes_beyond_the_heads(oracle_mod, res, sr, channels, channels_last):
    """The tile backward takes a RoI's gradient block as it lies in memory (16-byte LDS-DMA pieces); pooled sizes other
    than the heads' 7 x 7 / 14 x 14 -- 16 x 16 and 21 x 21 could not take the tile kernel at all before -- run the generic
    (run-time size) instance: gradients against the oracle, two images, RoIs from tiny to map-sized."""
    h, w, scale = 50, 84, 1.0 / 16
    feat = syn.feature_map(2, channels, h, w, seed=res)
    rois = syn.rois_adversarial(96, 2, h, w, scale, seed=res + 1)
    gtop = np.random.RandomState(res).randn(96, channels, res, res).astype(np.float32)
    out, grad = _roi_align_gpu(feat, rois, res, scale, sr, gtop, channels_last=channels_last)
    assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=8), "fwd %d" % res, exact=False)
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, sr, threads=8), "bwd %d" % res)


def test_roi_align_backward_with_a_misaligned_top_gradient(oracle_mod):
    """The tile backward fetches gradient blocks in 16-byte pieces.  A top_grad that is only dword-aligned is a legal argument
    at the C boundary: such a call takes the generic kernel (reference mapping, atomics) instead of being refused, with and
    without the OVERWRITE contract; the autograd Function copies such a gradient once and stays on the tile kernel."""
    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import roi_align_backward

    h, w, scale = 50, 84, 1.0 / 16
    rois = syn.rois_canonical(40, 1, seed=3, side=(32.0, 300.0))
    shape = (40, 64, 7, 7)
    gtop = np.random.RandomState(4).randn(*shape).astype(np.float32)
    flat = torch.zeros(int(np.prod(shape)) + 1, device=dev())
    flat[1:] = to_dev(gtop).reshape(-1)
    odd = flat[1:].view(shape)                      # contiguous, 4 bytes past a 16-byte boundary
    assert odd.is_contiguous() and odd.data_ptr() % 16 == 4
    want = oracle_mod.roi_align_backward(gtop, rois, (1, 64, h, w), scale, 2, threads=8)
    grad = roi_align_backward(odd, to_dev(rois), (1, 64, h, w), 7, 7, scale, 2)
    assert_close(grad, want, "misaligned top_grad through the autograd wrapper")
    lib = _lib.lib()
    ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(40), dtype=torch.uint8, device=dev())
    for flags, fill in ((0, 0.0), (_lib.ROI_ALIGN_OVERWRITE, float("nan"))):   # accumulate into zeros / overwrite garbage
        gin = torch.full((1, 64, h, w), fill, device=dev())
        rc = lib.mi_roi_align_backward_ws(odd.data_ptr(), to_dev(rois).data_ptr(), gin.data_ptr(), 1, 64, h, w, 40, 7, 7, scale,
                                          2, 0, 0, ws.data_ptr(), ws.numel(), flags, _lib.current_stream_handle(dev()))
        assert rc == 0, lib.mi_last_error()
        assert_close(gin, want, "misaligned top_grad at the C-ABI, flags %d" % flags)


def test_roi_align_golden():
    g = load_golden("roi_align.npz")
    feat, rois, scale = g["feat"], g["rois"], float(g["scale"])
    for key in [k[4:] for k in g.files if k.startswith("fwd_")]:
        sr, res = int(key.split("_")[0][2:]), int(key.split("_")[1][1:])
        out, grad = _roi_align_gpu(feat, rois, res, scale, sr, g["gtop_" + key])
        assert_fwd(out, g["fwd_" + key], "fwd " + key, exact=False)
        assert_close(grad, g["bwd_" + key], "bwd " + key)


@pytest.mark.parametrize("shape,res,sr,nrois", [((2, 8, 25, 42), 7, 2, 64), ((1, 16, 50, 84), 14, 2, 40),
                                                ((3, 5, 13, 21), 7, 0, 50), ((1, 3, 7, 9), 3, 3, 33),
                                                ((2, 64, 100, 168), 7, 2, 128), ((1, 70, 30, 40), 7, 1, 20),
                                                ((2, 32, 50, 84), 7, 0, 96), ((1, 96, 25, 42), 14, 2, 64),
                                                ((3, 32, 200, 336), 7, 2, 200), ((1, 32, 9, 70), 6, 3, 40)])
def test_roi_align_vs_oracle_adversarial_rois(oracle_mod, roi_align_impl, shape, res, sr, nrois):
    n, c, h, w = shape
    scale = 1.0 / 16
    feat = syn.feature_map(n, c, h, w, seed=res + sr)
    rois = syn.rois_adversarial(nrois, n, h, w, scale, seed=nrois)
    gtop = np.random.RandomState(7).randn(nrois, c, res, res).astype(np.float32)
    out, grad = _roi_align_gpu(feat, rois, res, scale, sr, gtop)
    ref_out = oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=8)
    assert_fwd(out, ref_out, "fwd", exact=_fwd_is_exact(roi_align_impl, c))
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, sr, threads=8), "bwd")


def test_roi_align_channels_last_storage(oracle_mod, roi_align_impl):
    n, c, h, w, scale = 2, 32, 25, 42, 1.0 / 32
    feat = syn.feature_map(n, c, h, w, seed=1)
    rois = syn.rois_adversarial(48, n, h, w, scale, seed=2)
    gtop = np.random.RandomState(3).randn(48, c, 7, 7).astype(np.float32)
    out, grad = _roi_align_gpu(feat, rois, 7, scale, 2, gtop, channels_last=True)
    assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, 7, 7, scale, 2), "fwd nhwc",
               exact=(roi_align_impl == "direct"))
    assert grad.is_contiguous(memory_format=torch.channels_last)
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, 2), "bwd nhwc")


@pytest.mark.parametrize("vec", ["4", "2", "1"])
@pytest.mark.parametrize("shape,res,sr,nrois,scale", [
    ((1, 256, 50, 84), 7, 2, 160, 1.0 / 16),    # 64 lanes x float4, the config-2 kernel
    ((2, 512, 25, 42), 7, 2, 96, 1.0 / 32),     # two chunks per RoI, two images
    ((2, 64, 50, 84), 14, 2, 64, 1.0 / 16),     # mask resolution (14 waves, one channel per lane)
    ((1, 256, 50, 84), 7, 0, 80, 1.0 / 16),     # adaptive sampling grid
    ((2, 6, 25, 42), 7, 2, 50, 1.0 / 32),       # C < 64 and C * bins not a multiple of 4 (dword copy-out)
    ((1, 100, 30, 40), 7, 3, 40, 1.0 / 16),     # partial last chunk, sampling_ratio 3 (generic loops)
    ((3, 128, 13, 21), 14, 1, 30, 1.0 / 32),    # tiny map, one sample per bin
])
def test_roi_align_nhwc_kernel_vs_oracle(oracle_mod, tuning_env, vec, shape, res, sr, nrois, scale):
    """roi_align_fwd_nhwc (channels-last features, NCHW output) with every channels-per-lane variant, on RoIs that
    include the ones the record tables cannot describe (reference-order path inside the kernel)."""
    tuning_env(MI_ROI_ALIGN_NHWC_V=vec)
    n, c, h, w = shape
    feat = syn.feature_map(n, c, h, w, seed=res + sr)
    rois = np.vstack([syn.rois_adversarial(nrois // 2, n, h, w, scale, seed=nrois),
                      syn.rois_canonical(nrois - nrois // 2, n, seed=3, side=(8.0 / scale / 4, 0.6 * h / scale),
                                         im_h=int(h / scale), im_w=int(w / scale))])
    out, _ = _roi_align_gpu(feat, rois, res, scale, sr, channels_last=True)
    assert out.is_contiguous()
    assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=8), "nhwc fwd", exact=False)


def test_roi_align_nhwc_config2_full_shape(oracle_mod, tuning_env):
    """BASELINE configs[1] with the features stored channels-last: forward through roi_align_fwd_nhwc (both tap
    batch sizes), backward through the NCHW tile kernel + layout change."""
    feat = syn.feature_map(1, 256, 200, 336, seed=0)
    rois = syn.rois_canonical(512, 1, seed=0)
    gtop = np.random.RandomState(1).randn(512, 256, 7, 7).astype(np.float32)
    threads = oracle_mod.num_threads_available()
    ref_out = oracle_mod.roi_align_forward(feat, rois, 7, 7, 0.25, 2, threads=threads)
    out, grad = _roi_align_gpu(feat, rois, 7, 0.25, 2, gtop, channels_last=True)
    assert_fwd(out, ref_out, "config-2 nhwc fwd", exact=False)
    assert grad.is_contiguous(memory_format=torch.channels_last)
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, 0.25, 2, threads=threads), "nhwc bwd")
    tuning_env(MI_ROI_ALIGN_NHWC_PB=7)
    out7, _ = _roi_align_gpu(feat, rois, 7, 0.25, 2, channels_last=True)
    assert torch.equal(out7, out.detach())  # same arithmetic, different load batching
    nchw, _ = _roi_align_gpu(feat, rois, 7, 0.25, 2)
    assert_close(out, nchw.detach().cpu().numpy(), "nhwc vs nchw fast path", atol=FAST_ATOL, rtol=FAST_ATOL)


def test_roi_align_config2_full_shape(oracle_mod, roi_align_impl):
    """BASELINE configs[1]: 512 RoIs x 256 ch x 7x7 on P2 (200x336), sampling_ratio 2, fwd + bwd."""
    feat = syn.feature_map(1, 256, 200, 336, seed=0)
    rois = syn.rois_canonical(512, 1, seed=0)
    gtop = np.random.RandomState(1).randn(512, 256, 7, 7).astype(np.float32)
    out, grad = _roi_align_gpu(feat, rois, 7, 0.25, 2, gtop)
    ref_out = oracle_mod.roi_align_forward(feat, rois, 7, 7, 0.25, 2, threads=oracle_mod.num_threads_available())
    assert_fwd(out, ref_out, "config-2 fwd", exact=(roi_align_impl == "direct"))
    ref_grad = oracle_mod.roi_align_backward(gtop, rois, feat.shape, 0.25, 2,
                                             threads=oracle_mod.num_threads_available())
    assert_close(grad, ref_grad, "config-2 bwd")
    # size-independent properties: mass conservation (all canonical RoIs are interior) and linearity
    assert abs(float(grad.sum()) - float(gtop.astype(np.float64).sum())) < 1e-2 * np.abs(gtop).sum() ** 0.5
    out2, _ = _roi_align_gpu(2.0 * feat, rois, 7, 0.25, 2)
    assert torch.equal(out2, 2.0 * out.detach())


@pytest.mark.parametrize("path", ["records", "channels_last", "no_workspace", "direct"])
@pytest.mark.parametrize("res,sr", [(7, 2), (6, 0)])
def test_roi_align_non_finite_border_pixels_propagate_as_in_the_reference(oracle_mod, tuning_env, path, res, sr):
    """A sample clamped to the last row / column reads the border pixel twice in the reference (weights 1 and 0): an Inf
    there gives NaN, an Inf in the second-to-last row gives nothing.  Every forward path keeps that: the record kernels
    encode the pair as (size - 1, size) and read `size` from the clamped address (they used to encode (size - 2, size - 1)
    with weights (0, 1), which moved the NaN to the wrong bins)."""
    from detectron_pytorch_amd.roi_align import roi_align_forward

    n, c, h, w, scale = 2, 64, 50, 84, 1.0 / 16
    feat = syn.feature_map(n, c, h, w, seed=3)
    rois = syn.rois_adversarial(64, n, h, w, scale, seed=11)
    feat[:, :, -1, :] = np.inf
    feat[:, :, :, -1] = -np.inf
    feat[:, :, 0, 0] = np.nan
    feat[:, :, -2, 5] = np.inf   # second-to-last row: must NOT leak into bins clamped to the last row
    feat[:, :, 7, -2] = np.nan
    ref = oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=8)
    tuning_env(MI_ROI_ALIGN_IMPL="direct" if path == "direct" else None,
               MI_ROI_ALIGN_NO_WS="1" if path == "no_workspace" else None)
    f = to_dev(feat)
    if path == "channels_last":
        f = f.contiguous(memory_format=torch.channels_last)
    out = roi_align_forward(f, to_dev(rois), res, res, scale, sr).cpu().numpy()
    assert np.isnan(ref).any() and np.isinf(ref).any()
    assert np.array_equal(np.isnan(out), np.isnan(ref)) and np.array_equal(np.isinf(out), np.isinf(ref))
    fin = np.isfinite(ref)
    assert np.array_equal(np.sign(out[~fin & ~np.isnan(ref)]), np.sign(ref[~fin & ~np.isnan(ref)]))
    assert np.abs(out[fin] - ref[fin]).max() <= FAST_ATOL


def _edge_rois(batch, height, width, scale, seed):
    """RoIs inside the image whose windows reach the last row, the last column and the bottom-right corner of the LAST image
    (where a 16-byte group of the window copy runs past the row end, the map and the allocation), a few interior ones
    around, fractional coordinates throughout (no tap with weight exactly 0)."""
    rng = np.random.RandomState(seed)
    im_w, im_h = width / scale, height / scale
    last = batch - 1
    rows = [[last, im_w - 0.37 / scale - rng.uniform(3, 9) / scale, im_h * 0.31, im_w - 0.37 / scale, im_h * 0.31 + rng.uniform(4, 9) / scale],
            [last, im_w * 0.27, im_h - 0.41 / scale - rng.uniform(3, 8) / scale, im_w * 0.27 + rng.uniform(4, 9) / scale, im_h - 0.41 / scale],
            [last, im_w - 0.23 / scale - 5.3 / scale, im_h - 0.29 / scale - 4.7 / scale, im_w - 0.23 / scale, im_h - 0.29 / scale],
            [last, im_w - 1.0 - 2.6 / scale, im_h - 1.0 - 2.2 / scale, im_w - 1.0, im_h - 1.0],     # the reference's clamped last sample
            [0, 0.19 / scale, 0.23 / scale, 6.6 / scale, 5.4 / scale]]
    for _ in range(11):
        x1, y1 = rng.uniform(1.1, width - 12.3) / scale, rng.uniform(1.1, height - 12.3) / scale
        rows.append([rng.randint(0, batch), x1, y1, x1 + rng.uniform(2.3, 10.7) / scale, y1 + rng.uniform(2.3, 10.7) / scale])
    return np.asarray(rows, np.float32)


@pytest.mark.parametrize("path", ["records", "channels_last", "direct"])
@pytest.mark.parametrize("channels,height,width", [(32, 25, 42), (256, 50, 84), (32, 200, 336)])
def test_roi_align_consumes_nothing_outside_its_windows(oracle_mod, tuning_env, path, channels, height, width):
    """The NCHW forward copies window rows in 16-byte groups that run past the row end (into the next row, the next channel,
    or behind the descriptor) and lays them out on a padded pitch.  Nothing of that may reach a sum: every feature pixel
    no sample taps is NaN here, the features are the TAIL of a NaN-filled allocation (the last channel of the last image
    ends with the buffer) with NaN guards in front, and the RoIs sit on the last row / column / corner of the last image.
    Forward and backward through the C-ABI on caller-placed buffers; the gradient map sits between sentinel guards."""
    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    n, res, sr, scale = 2, 7, 2, 0.25
    rois = _edge_rois(n, height, width, scale, seed=channels + width)
    r = rois.shape[0]
    # pixels some sample taps: where the gradient of an all-ones top gradient is non-zero (one channel is enough)
    tapped = oracle_mod.roi_align_backward(np.ones((r, 1, res, res), np.float32), rois, (n, 1, height, width), scale, sr,
                                           threads=4)[:, 0] != 0
    assert tapped[n - 1, height - 1, width - 1] and tapped[n - 1, height - 1].any() and tapped[n - 1, :, width - 1].any()
    feat = syn.feature_map(n, channels, height, width, seed=7)
    feat[np.broadcast_to(~tapped[:, None], feat.shape)] = np.nan
    gtop = np.random.RandomState(3).randn(r, channels, res, res).astype(np.float32)
    ref = oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=4)
    ref_grad = oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, sr, threads=4)
    assert np.isfinite(ref).all()

    tuning_env(MI_ROI_ALIGN_IMPL="direct" if path == "direct" else None)
    lib, d = _lib.lib(), dev()
    layout = _lib.LAYOUT_NHWC if path == "channels_last" else _lib.LAYOUT_NCHW
    f_np = np.ascontiguousarray(feat.transpose(0, 2, 3, 1)) if path == "channels_last" else feat
    guard = 4096
    fbuf = torch.full((guard + feat.size,), float("nan"), device=d)
    fbuf[guard:] = to_dev(f_np).reshape(-1)                                   # features = the tail of the allocation
    gbuf = torch.full((guard + gtop.size,), float("nan"), device=d)
    gbuf[guard:] = to_dev(gtop).reshape(-1)
    sentinel = 12345.0
    dbuf = torch.full((guard + feat.size + guard,), sentinel, device=d)       # gradient map between two guards
    obuf = torch.full((guard + ref.size + guard,), sentinel, device=d)
    rois_d = to_dev(rois)
    ws = torch.empty(_backward_workspace_bytes([(height, width)], n, r), dtype=torch.uint8, device=d)
    stream = _lib.current_stream_handle(d)
    f_ptr, g_ptr = fbuf.data_ptr() + 4 * guard, gbuf.data_ptr() + 4 * guard
    o_ptr, d_ptr = obuf.data_ptr() + 4 * guard, dbuf.data_ptr() + 4 * guard
    _lib.check(lib.mi_roi_align_forward_ws(f_ptr, rois_d.data_ptr(), o_ptr, n, channels, height, width, r, res, res, scale,
                                           sr, _lib.ROI_ALIGN_CAFFE2, layout, ws.data_ptr(), ws.numel(), stream), "forward")
    flags = _lib.ROI_ALIGN_OVERWRITE if lib.mi_roi_align_backward_overwrites(channels, height, width, r, res, res,
                                                                            _lib.ROI_ALIGN_CAFFE2, layout) else 0
    if not flags:
        dbuf[guard:guard + feat.size] = 0
    _lib.check(lib.mi_roi_align_backward_ws(g_ptr, rois_d.data_ptr(), d_ptr, n, channels, height, width, r, res, res, scale,
                                            sr, _lib.ROI_ALIGN_CAFFE2, layout, ws.data_ptr(), ws.numel(), flags, stream),
               "backward")
    torch.cuda.synchronize()
    out = obuf[guard:guard + ref.size].cpu().numpy().reshape(ref.shape)
    grad = dbuf[guard:guard + feat.size].cpu().numpy().reshape(f_np.shape)
    if path == "channels_last":
        grad = grad.transpose(0, 3, 1, 2)
    assert np.isfinite(out).all(), "%d outputs took a value from outside their windows" % int((~np.isfinite(out)).sum())
    assert_fwd(out, ref, "forward on the map's last row / column / corner", exact=(path == "direct"))
    assert_close(grad, ref_grad, "backward on the map's last row / column / corner")
    for buf, size in ((obuf, ref.size), (dbuf, feat.size)):
        assert bool((buf[:guard] == sentinel).all()) and bool((buf[guard + size:] == sentinel).all()), "guard overwritten"


@pytest.mark.parametrize("variant", ["records", "channels_last"])
@pytest.mark.parametrize("case", ["adversarial", "piled", "nonfinite", "fpn", "sr0"])
def test_roi_align_forward_kernels_on_hard_inputs(oracle_mod, tuning_env, variant, case):
    """The record-driven forward kernels (NCHW: roi_align_fwd_records; channels-last: roi_align_fwd_nhwc) on inputs that reach
    every branch: RoIs the tables cannot describe (reference-order path inside the kernel) and RoIs of no image, a pile of
    300 RoIs on one spot, non-finite features at the borders (a clamped sample reads the border pixel twice, as the
    reference), an FPN pyramid in one call, an adaptive sampling grid (generic kernel)."""
    from detectron_pytorch_amd.roi_align import roi_align_forward, roi_align_fpn

    nhwc = variant == "channels_last"

    def to_maps(a):
        t = to_dev(a)
        return t.contiguous(memory_format=torch.channels_last) if nhwc else t

    if case == "fpn":
        frois, flv = syn.rois_fpn_distributed(400, batch=2, seed=5)
        maps = [syn.feature_map(2, 64, syn.FPN_LEVELS[l][0], syn.FPN_LEVELS[l][1], seed=l) for l in (5, 4, 3, 2)]
        scales = [syn.FPN_LEVELS[l][2] for l in (5, 4, 3, 2)]
        idx = np.array([(5, 4, 3, 2).index(int(l)) for l in flv], dtype=np.int32)
        out = roi_align_fpn([to_maps(m) for m in maps], scales, to_dev(frois), to_dev(idx), 7, 7, 2).cpu().numpy()
        for li in range(4):
            sel = np.nonzero(idx == li)[0]
            ref = oracle_mod.roi_align_forward(maps[li], frois[sel], 7, 7, scales[li], 2, threads=8)
            assert np.abs(out[sel] - ref).max() <= FAST_ATOL
        return
    n, c, h, w, scale, res, sr = 2, 64, 50, 84, 1.0 / 16, 7, 2
    feat = syn.feature_map(n, c, h, w, seed=3)
    if case == "adversarial":
        rois = syn.rois_adversarial(96, n, h, w, scale, seed=9)
    elif case == "piled":
        rois = syn.rois_canonical(300, n, seed=4, side=(32.0, 200.0))
        rois[:, 0] = 1.0
        rois[:, 1:] = rois[:1, 1:] + np.random.RandomState(0).uniform(-20, 20, (300, 4)).astype(np.float32)
    elif case == "nonfinite":
        rois = syn.rois_adversarial(64, n, h, w, scale, seed=11)
        feat[:, :, -1, :] = np.inf
        feat[:, :, :, -1] = -np.inf
        feat[:, :, 0, 0] = np.nan
        feat[:, :, -2, 5] = np.inf   # second-to-last row: must NOT leak into bins clamped to the last row
        feat[:, :, 7, -2] = np.nan
    else:
        rois = syn.rois_adversarial(80, n, h, w, scale, seed=13)
        res, sr = 6, 0
    ref = oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=8)
    if case == "adversarial":  # RoIs of no image (padding rows of the training path) pool zeros
        rois[5, 0], rois[6, 0] = -1.0, 7.0
        ref[5:7] = 0.0
    out = roi_align_forward(to_maps(feat), to_dev(rois), res, res, scale, sr).cpu().numpy()
    if case == "nonfinite":
        assert np.array_equal(np.isnan(out), np.isnan(ref)) and np.array_equal(np.isinf(out), np.isinf(ref))
        fin = np.isfinite(ref)
        assert np.array_equal(np.sign(out[~fin & ~np.isnan(ref)]), np.sign(ref[~fin & ~np.isnan(ref)]))
        assert np.abs(out[fin] - ref[fin]).max() <= FAST_ATOL
    else:
        assert np.abs(out - ref).max() <= FAST_ATOL


def test_roi_align_mask_head_shape_and_multi_image(oracle_mod, roi_align_impl):
    """config-2 variant (i)/(iii): 14x14 mask resolution, two images per rank."""
    feat = syn.feature_map(2, 256, 100, 168, seed=4)
    rois = syn.rois_canonical(128, 2, seed=5, side=(32.0, 300.0))
    gtop = np.random.RandomState(6).randn(128, 256, 14, 14).astype(np.float32)
    out, grad = _roi_align_gpu(feat, rois, 14, 0.125, 2, gtop)
    assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, 14, 14, 0.125, 2, threads=8), "mask fwd",
               exact=(roi_align_impl == "direct"))
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, 0.125, 2, threads=8), "mask bwd")


def test_roi_align_stream_path_small_and_fpn_sized_rois(oracle_mod):
    """RoIs the LDS ring serves directly (FPN-assigned sizes on every level) plus degenerate ones."""
    rois_all, lvls = syn.rois_fpn_distributed(400, batch=2, seed=3)
    for lvl, (h, w, scale) in syn.FPN_LEVELS.items():
        rois = rois_all[lvls == lvl]
        feat = syn.feature_map(2, 64, h, w, seed=lvl)
        gtop = np.random.RandomState(lvl).randn(len(rois), 64, 7, 7).astype(np.float32)
        out, grad = _roi_align_gpu(feat, rois, 7, scale, 2, gtop)
        assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, 7, 7, scale, 2, threads=8), "lvl %d" % lvl,
                   exact=False)
        assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, 2, threads=8), "lvl %d" % lvl)


def test_roi_align_empty_and_autograd_contract():
    from detectron_pytorch_amd.roi_align import RoIAlignFunction

    f = torch.randn(1, 4, 10, 10, device=dev(), requires_grad=True)
    out = RoIAlignFunction(7, 7, 0.25, 2)(f, torch.zeros(0, 5, device=dev()))
    assert out.shape == (0, 4, 7, 7)
    rois = torch.tensor([[0, 4, 4, 30, 30]], device=dev(), dtype=torch.float32, requires_grad=True)
    out = RoIAlignFunction(7, 7, 0.25, 2)(f, rois)
    out.sum().backward()
    assert rois.grad is None and f.grad is not None  # backward returns (grad_input, None)
    with pytest.raises(ValueError):
        RoIAlignFunction(7, 7, 0.25, 2)(f, torch.zeros(3, 4, device=dev()))


def test_roi_align_fpn_without_rois_gives_zero_gradients():
    """An empty RoI set (no foreground RoI for the mask head) must hand back zero-filled gradients for every level, not
    the uninitialised memory an OVERWRITE-mode backward would otherwise leave (mi_roi_align_backward_fpn, R == 0)."""
    from detectron_pytorch_amd.roi_align import roi_align_fpn

    scales = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4]
    feats = [torch.randn(2, 8, int(np.ceil(256 * s)), int(np.ceil(320 * s)), device=dev(), requires_grad=True) for s in scales]
    poison = [torch.full_like(f, float("nan")) for f in feats for _ in range(4)]      # dirty the allocator's free blocks
    del poison
    out = roi_align_fpn(feats, scales, torch.zeros((0, 5), device=dev()), torch.zeros((0,), dtype=torch.int32, device=dev()),
                        7, 7, 2)
    assert out.shape == (0, 8, 7, 7)
    out.sum().backward()
    for f in feats:
        assert f.grad is not None and float(f.grad.abs().sum()) == 0.0


def test_roi_align_forward_large_lds_caps(oracle_mod, tuning_env):
    """MI_ROI_ALIGN_CAP = 448 / 640 ask for more than 64 KB of dynamic LDS: the launcher opts in (it used to fail)."""
    feat = syn.feature_map(1, 32, 100, 168, seed=2)
    rois = syn.rois_canonical(64, 1, seed=3, im_h=800, im_w=1344)
    want = oracle_mod.roi_align_forward(feat, rois, 7, 7, 0.125, 2, threads=8)
    for cap in ("448", "640", "256", "192"):
        tuning_env(MI_ROI_ALIGN_CAP=cap)
        out, _ = _roi_align_gpu(feat, rois, 7, 0.125, 2)
        assert_fwd(out, want, "cap " + cap, exact=False)
    feat14 = syn.feature_map(2, 64, 50, 84, seed=4)
    rois14 = syn.rois_canonical(48, 2, seed=5, im_h=800, im_w=1344)
    want14 = oracle_mod.roi_align_forward(feat14, rois14, 14, 14, 1.0 / 16, 2, threads=8)
    for cap in ("256", "448"):
        tuning_env(MI_ROI_ALIGN_CAP=cap)
        out, _ = _roi_align_gpu(feat14, rois14, 14, 1.0 / 16, 2)
        assert_fwd(out, want14, "14x14 cap " + cap, exact=False)


def test_roi_align_forward_stages_dealt_to_several_workgroups(oracle_mod, tuning_env):
    """A launch of few multi-stage items deals every item's stages to two workgroups (MI_ROI_ALIGN_FWD_SPLIT forces 1 / 2 /
    4): the same stage code runs either way, so the outputs are bit-equal between the settings and equal to the oracle --
    pooled heights whose stage count is not a multiple of the split included, and a RoI of no image / a RoI the records
    cannot describe among them (only part 0 of such an item may write)."""
    for ah, hw, scale in ((14, (50, 84), 1.0 / 16), (9, (64, 96), 0.125), (21, (40, 56), 1.0 / 16)):
        feat = syn.feature_map(2, 64, hw[0], hw[1], seed=6 + ah)
        rois = syn.rois_canonical(40, 2, seed=7 + ah, im_h=int(hw[0] / scale), im_w=int(hw[1] / scale))
        rois[3, 0] = 5.0                                             # no such image: zeros
        rois[4, 1:] = np.array([-3000.0, -3000.0, 9000.0, 9000.0], np.float32)   # far wider than any LDS window
        valid = rois[:, 0] < 2                                       # the oracle refuses rows of no image: zeros there
        want = np.zeros((rois.shape[0], 64, ah, ah), np.float32)
        want[valid] = oracle_mod.roi_align_forward(feat, rois[valid], ah, ah, scale, 2, threads=8)
        outs = {}
        for split in ("1", "2", "4", None):
            tuning_env(MI_ROI_ALIGN_FWD_SPLIT=split)
            out, _ = _roi_align_gpu(feat, rois, ah, scale, 2)
            assert_fwd(out, want, "%dx%d split %s" % (ah, ah, split), exact=False)
            outs[split] = out.detach().cpu().numpy()
        for split in ("2", "4", None):
            np.testing.assert_array_equal(outs[split], outs["1"], err_msg="%dx%d split %s" % (ah, ah, split))


# ---- RoIAlign (legacy) ----------------------------------------------------------------------------
def test_roi_align_forward_partial_wait_equals_the_full_wait(tuning_env):
    """The stage pipeline of roi_align_fwd_records waits with s_waitcnt vmcnt(#stores of the previous stage) for the next
    window (roi_align_records.hip, wait_vmcnt_at_most): correct as long as vmcnt retires in order and the store count is
    an upper bound of what the compiler emitted.  MI_ROI_ALIGN_FWD_FULL_WAIT=1 makes the same (release) kernel wait with
    vmcnt(0); on shapes whose items have several stages -- 14 x 14 heads, large windows, one or two workgroups per item --
    both must give the same bits."""
    from detectron_pytorch_amd.roi_align import roi_align_forward

    h, w, scale = syn.FPN_LEVELS[2]
    feat = to_dev(syn.feature_map(1, 64, h, w, seed=5))
    for res, rois_np in ((14, syn.rois_canonical(96, 1, seed=6)), (7, syn.rois_canonical(96, 1, seed=7, side=(120.0, 420.0))),
                         (14, syn.rois_canonical(40, 1, seed=8, side=(200.0, 700.0)))):
        rois = to_dev(rois_np)
        outs = {}
        for split in (1, 2):
            for full in (None, 1):
                tuning_env(MI_ROI_ALIGN_FWD_FULL_WAIT=full, MI_ROI_ALIGN_FWD_SPLIT=split)
                outs[(split, full)] = roi_align_forward(feat, rois, res, res, scale, 2).clone()
        ref = outs[(1, 1)]
        for key, o in outs.items():
            assert torch.equal(o, ref), "res %d, (split, full wait) %r differs from the full-wait single-workgroup result" % (res, key)


def test_roi_align_legacy_golden_and_oracle(oracle_mod):
    from detectron_pytorch_amd.roi_align import LegacyRoIAlignFunction

    g = load_golden("roi_align_legacy.npz")
    f = to_dev(g["feat"]).requires_grad_(True)
    out = LegacyRoIAlignFunction(7, 7, float(g["scale"]))(f, to_dev(g["rois"]))
    assert np.array_equal(out.detach().cpu().numpy(), g["fwd"])
    out.backward(to_dev(g["gtop"]))
    assert_close(f.grad, g["bwd"], "legacy bwd")
    feat = syn.feature_map(2, 12, 30, 44, seed=9)
    rois = syn.rois_canonical(40, 2, seed=10, im_h=30 * 16, im_w=44 * 16)
    out = LegacyRoIAlignFunction(8, 8, 1.0 / 16)(to_dev(feat), to_dev(rois))
    assert np.array_equal(out.cpu().numpy(), oracle_mod.roi_align_legacy_forward(feat, rois, 8, 8, 1.0 / 16))


# ---- RoIPool ---------------------------------------------------------------------------------------
def test_roi_pool_golden_and_oracle(oracle_mod):
    from detectron_pytorch_amd.roi_pool import RoIPoolFunction, roi_pool_forward

    g = load_golden("roi_pool.npz")
    scale = float(g["scale"])
    f = to_dev(g["feat"]).requires_grad_(True)
    out = RoIPoolFunction(7, 7, scale)(f, to_dev(g["rois"]))
    assert np.array_equal(out.detach().cpu().numpy(), g["fwd"])
    _, argmax = roi_pool_forward(to_dev(g["feat"]), to_dev(g["rois"]), 7, 7, scale)
    assert np.array_equal(argmax.cpu().numpy(), g["argmax"])
    out.backward(to_dev(g["gtop"]))
    assert np.array_equal(f.grad.cpu().numpy(), g["bwd"])   # the tile backward adds in the reference's order: bit-equal
    feat = syn.feature_map(2, 40, 38, 50, seed=12)
    rois = syn.rois_adversarial(100, 2, 38, 50, 1.0 / 16, seed=13)
    gtop = np.random.RandomState(14).randn(100, 40, 7, 7).astype(np.float32)
    f = to_dev(feat).requires_grad_(True)
    out = RoIPoolFunction(7, 7, 1.0 / 16)(f, to_dev(rois))
    ref_out, ref_arg = oracle_mod.roi_pool_forward(feat, rois, 7, 7, 1.0 / 16, threads=8)
    assert np.array_equal(out.detach().cpu().numpy(), ref_out)
    out.backward(to_dev(gtop))
    assert np.array_equal(f.grad.cpu().numpy(), oracle_mod.roi_pool_backward(gtop, rois, ref_arg, feat.shape, 1.0 / 16, threads=8))


@pytest.mark.parametrize("shape,res,scale,nrois", [
    ((1, 32, 50, 84), (7, 7), 1.0 / 16, 64),      # stride-16 map: whole-image RoIs, bins wider than eight columns
    ((2, 70, 25, 42), (7, 7), 1.0 / 32, 48),      # 70 channels: a ragged last tile of 6
    ((1, 32, 40, 400), (7, 7), 1.0 / 4, 40),      # bins of up to 58 columns: several column blocks per row
    ((1, 64, 60, 90), (14, 14), 1.0 / 8, 40),     # 196 bins per channel
    ((1, 32, 30, 30), (3, 60), 1.0 / 16, 24),     # more bin columns than pixels
    ((2, 8, 9, 11), (2, 3), 1.0 / 16, 20)])       # fewer channels than a tile, tiny map
def test_roi_pool_kernel_shapes_vs_oracle(oracle_mod, shape, res, scale, nrois):
    """roi_pool_fwd reads a bin four rows x eight columns at a time (masked past the bin) and takes wide bins row by row in
    column blocks: values AND flat int32 argmax bit-equal to the oracle (roi_pooling_kernel.cu:24-93) on narrow, wide and
    tall bins, ragged channel tiles, RoIs outside the map, malformed ones and RoIs of no image."""
    from detectron_pytorch_amd.roi_pool import roi_pool_forward

    n, c, h, w = shape
    feat = syn.feature_map(n, c, h, w, seed=31)
    feat[0, 0, :3, :3] = 7.5          # ties: the FIRST maximum in row-major order must win
    feat[-1, -1] = -np.inf            # a plane nothing beats -FLT_MAX in: value -FLT_MAX, argmax -1
    rois = syn.rois_adversarial(nrois, n, h, w, scale, seed=32)
    ref_out, ref_arg = oracle_mod.roi_pool_forward(feat, rois, res[0], res[1], scale, threads=8)
    rois[-1, 0] = n + 3               # RoIs of no image: the reference (and its oracle) would read past the input; here: empty bins
    rois[-2, 0] = -1
    out, argmax = roi_pool_forward(to_dev(feat), to_dev(rois), res[0], res[1], scale)
    ok = (rois[:, 0] >= 0) & (rois[:, 0] < n)
    assert np.array_equal(out.cpu().numpy()[ok], ref_out[ok])
    assert np.array_equal(argmax.cpu().numpy()[ok], ref_arg[ok])
    assert not out.cpu().numpy()[~ok].any() and (argmax.cpu().numpy()[~ok] == -1).all()


@pytest.mark.parametrize("shape,res,scale,nrois,kind", [
    ((1, 32, 50, 84), (7, 7), 1.0 / 16, 64, "adversarial"),    # whole-image RoIs: every tile, long lists
    ((2, 70, 25, 42), (7, 7), 1.0 / 32, 48, "adversarial"),    # 70 channels: ragged channel group and ragged wave; W % 4 != 0
    ((1, 40, 60, 90), (14, 14), 1.0 / 8, 40, "adversarial"),   # 14 bin rows: two row blocks per RoI, the 16-column register block
    ((1, 32, 30, 30), (3, 60), 1.0 / 16, 24, "adversarial"),   # pooled width beyond the register block: one bin at a time
    ((2, 8, 9, 11), (2, 3), 1.0 / 16, 20, "adversarial"),      # a map smaller than a tile
    ((1, 16, 40, 70), (7, 7), 1.0 / 4, 300, "tiny"),           # RoIs of 1-6 pixels: a pixel in up to seven bins of an axis
    ((2, 32, 48, 64), (7, 7), 1.0 / 8, 600, "clustered"),      # three rounds of the RoI scan, hundreds of RoIs on the same tiles
    ((1, 8, 35, 1), (4, 1), 1.0 / 16, 12, "adversarial"),      # width 1
    ((1, 128, 200, 336), (7, 7), 1.0 / 4, 96, "adversarial"),  # 275 tiles x 4 channel groups: the 8-channels-per-wave instance
    ((1, 64, 200, 336), (7, 7), 1.0 / 4, 96, "adversarial"),   # 275 tiles x 4 groups of 16: the 4-channels-per-wave instance
    ((1, 128, 200, 336), (7, 7), 1.0 / 4, 200, "tiny"),        # the one-element-at-a-time path of both
    ((1, 64, 200, 336), (7, 7), 1.0 / 4, 200, "tiny")])
def test_roi_pool_backward_tiles_bit_equal_to_the_reference_order(oracle_mod, shape, res, scale, nrois, kind):
    """roi_pool_bwd_tiles (LDS accumulators per 8 x 32 tile, RoIs in ascending index, a pixel's terms in four ordered passes or
    one element at a time; 2 channels per wave on these small maps, 8 / 4 on the last two shapes) against the oracle of ROIPoolBackward (roi_pooling_kernel.cu:128-203), which adds a pixel's terms by
    ascending (RoI, ph, pw): BIT-equal, every element of a NaN-filled gradient map overwritten (:202), through the raw C-ABI."""
    from detectron_pytorch_amd import _lib

    n, c, h, w = shape
    rng = np.random.RandomState(77)
    feat = syn.feature_map(n, c, h, w, seed=51)
    feat = np.round(feat * 4) / 4          # many ties: overlapping bins of one RoI pick the SAME pixel (the order matters)
    if kind == "adversarial":
        rois = syn.rois_adversarial(nrois, n, h, w, scale, seed=52)
    elif kind == "tiny":
        x1 = rng.uniform(0, w / scale, nrois)
        y1 = rng.uniform(0, h / scale, nrois)
        rois = np.stack([np.zeros(nrois), x1, y1, x1 + rng.uniform(0, 6 / scale, nrois), y1 + rng.uniform(0, 6 / scale, nrois)], 1).astype(np.float32)
    else:
        cx, cy = rng.uniform(0.3, 0.7, 2)
        x1 = (cx + rng.normal(0, 0.05, nrois)) * w / scale
        y1 = (cy + rng.normal(0, 0.05, nrois)) * h / scale
        rois = np.stack([rng.randint(0, n, nrois), x1, y1, x1 + rng.uniform(8, 200, nrois), y1 + rng.uniform(8, 200, nrois)], 1).astype(np.float32)
    ref_out, ref_arg = oracle_mod.roi_pool_forward(feat, rois, res[0], res[1], scale, threads=8)
    gtop = rng.randn(nrois, c, *res).astype(np.float32)
    want = oracle_mod.roi_pool_backward(gtop, rois, ref_arg, feat.shape, scale, threads=8)
    lib = _lib.lib()
    got = torch.full((n, c, h, w), float("nan"), device=dev())
    d_gtop, d_rois, d_arg = to_dev(gtop), to_dev(rois), to_dev(ref_arg)
    rc = lib.mi_roi_pool_backward(d_gtop.data_ptr(), d_rois.data_ptr(), d_arg.data_ptr(), got.data_ptr(), n, c, h, w,
                                  nrois, res[0], res[1], scale, _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.abs(want).sum() > 0
    # a fabricated argmax (other channels, other images, pixels outside the rectangle, out of range): the reference adds a
    # term only where the index is the pixel's own (:193); no RoI at all: zeros
    fake = (ref_arg.astype(np.int64) + rng.choice([0, 0, 0, 1, -1, w, -w, h * w, -h * w, c * h * w, -c * h * w, 10 ** 9], ref_arg.shape))
    fake = np.clip(fake, -3, 2 ** 31 - 1).astype(np.int32)
    want = oracle_mod.roi_pool_backward(gtop, rois, fake, feat.shape, scale, threads=8)
    got.fill_(float("nan"))
    d_fake = to_dev(fake)
    rc = lib.mi_roi_pool_backward(d_gtop.data_ptr(), d_rois.data_ptr(), d_fake.data_ptr(), got.data_ptr(), n, c, h, w,
                                  nrois, res[0], res[1], scale, _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    assert np.array_equal(got.cpu().numpy(), want)
    got.fill_(float("nan"))
    rc = lib.mi_roi_pool_backward(None, None, None, got.data_ptr(), n, c, h, w, 0, res[0], res[1], scale, _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    assert not got.cpu().numpy().any()


@pytest.mark.parametrize("shape,grid_hw,nrois,span", [
    ((2, 32, 50, 84), (14, 14), 16, 1.25),   # 196 points: four groups of 64
    ((1, 70, 33, 47), (7, 7), 9, 1.6),       # ragged channel tile, grids largely outside the image
    ((1, 32, 40, 400), (7, 7), 8, 1.0),      # grids spanning a 400-pixel-wide map
    ((3, 8, 6, 5), (3, 2), 12, 1.3)])        # tiny
def test_roi_crop_kernel_shapes_vs_oracle(oracle_mod, shape, grid_hw, nrois, span):
    """roi_crop_fwd / _bwd work from a per-workgroup table of the grid points (64 per group): forward bit-equal to the oracle
    (roi_crop_cuda_kernel.cu:47-109), unwritten elements stay as the caller left them; the backward's sums to 1e-4."""
    from detectron_pytorch_amd import _lib

    n, c, h, w = shape
    feat = syn.feature_map(n, c, h, w, seed=41)
    grid = syn.crop_grid(nrois, grid_hw[0], grid_hw[1], seed=42, span=span)
    grid[0] = np.random.RandomState(43).uniform(-1.1, 1.1, grid[0].shape)   # a grid that is no box: arbitrary points
    if shape[3] == 400:
        grid[1, ..., 1] = np.linspace(-1, 1, grid_hw[1])[None, :]           # spans the whole width
    gtop = np.random.RandomState(44).randn(nrois, c, *grid_hw).astype(np.float32)
    lib = _lib.lib()
    out = torch.full((nrois, c) + tuple(grid_hw), 123.0, device=dev())
    f, gr = to_dev(feat), to_dev(grid)
    rc = lib.mi_roi_crop_forward(f.data_ptr(), gr.data_ptr(), out.data_ptr(), n, c, h, w, nrois, grid_hw[0], grid_hw[1],
                                 _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    ref = oracle_mod.roi_crop_forward(feat, grid)           # zero where nothing is written
    ref_sentinel = oracle_mod.roi_crop_forward(feat + 1e3, grid)  # which elements ARE written: those that moved
    written = ref_sentinel != ref
    got = out.cpu().numpy()
    assert np.array_equal(got[written], ref[written])
    nw = ~written   # not written by the reference (all four taps outside), or written with in-image weights that sum to 0
    assert ((got[nw] == 123.0) | (got[nw] == ref[nw])).all()
    if span > 1.2:
        assert (got == 123.0).any(), "some points of these grids have no tap in the image: they must stay unwritten"
    gin = torch.zeros_like(f)
    rc = lib.mi_roi_crop_backward(f.data_ptr(), gr.data_ptr(), to_dev(gtop).data_ptr(), gin.data_ptr(), n, c, h, w, nrois,
                                  grid_hw[0], grid_hw[1], _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    want = oracle_mod.roi_crop_backward(feat, grid, gtop)
    assert_close(gin, want, "roi_crop bwd")
    # the tile form (mi_roi_crop_backward_ws): no atomics, OVERWRITES a NaN-filled gradient, same sums in another order
    gin2 = torch.full_like(f, float("nan"))
    ws = torch.empty(lib.mi_roi_crop_backward_workspace_bytes(nrois), dtype=torch.uint8, device=dev())
    d_gtop = to_dev(gtop)
    rc = lib.mi_roi_crop_backward_ws(f.data_ptr(), gr.data_ptr(), d_gtop.data_ptr(), gin2.data_ptr(), n, c, h, w, nrois,
                                     grid_hw[0], grid_hw[1], ws.data_ptr(), ws.numel(), _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    assert_close(gin2, want, "roi_crop bwd tiles")
    assert lib.mi_roi_crop_backward_ws(f.data_ptr(), gr.data_ptr(), d_gtop.data_ptr(), gin2.data_ptr(), n, c, h, w, nrois,
                                       grid_hw[0], grid_hw[1], ws.data_ptr(), 8, _lib.current_stream_handle(dev())) != 0   # workspace too small


def test_roi_crop_backward_tiles_dense_grids_and_no_rois(oracle_mod):
    """Grids denser than the pixels (several points of a RoI share their top-left pixel: the entries that take LDS atomics
    instead of plain read / add / write), grids of one pixel, RoIs of a second image, and no RoI at all (zeros)."""
    from detectron_pytorch_amd import _lib

    lib = _lib.lib()
    n, c, h, w, nrois, gh, gw = 2, 40, 21, 45, 12, 7, 7
    rng = np.random.RandomState(5)
    feat = syn.feature_map(n, c, h, w, seed=61)
    grid = np.empty((nrois, gh, gw, 2), np.float32)
    for r in range(nrois):
        cy, cx = rng.uniform(-0.9, 0.9, 2)
        span = [0.0, 0.01, 0.05, 0.2][r % 4]            # 0: all 49 points in one pixel; 0.05: ~2 x 1 pixels
        grid[r, ..., 0] = cy + span * np.linspace(-1, 1, gh)[:, None]
        grid[r, ..., 1] = cx + span * np.linspace(-1, 1, gw)[None, :]
    gtop = rng.randn(nrois, c, gh, gw).astype(np.float32)
    want = oracle_mod.roi_crop_backward(feat, grid, gtop)
    f, gr, gt = to_dev(feat), to_dev(grid), to_dev(gtop)
    got = torch.full_like(f, float("nan"))
    ws = torch.empty(lib.mi_roi_crop_backward_workspace_bytes(nrois), dtype=torch.uint8, device=dev())
    rc = lib.mi_roi_crop_backward_ws(f.data_ptr(), gr.data_ptr(), gt.data_ptr(), got.data_ptr(), n, c, h, w, nrois, gh, gw,
                                     ws.data_ptr(), ws.numel(), _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    assert_close(got, want, "dense grids")
    assert np.abs(want[1]).sum() > 0
    got.fill_(float("nan"))
    rc = lib.mi_roi_crop_backward_ws(f.data_ptr(), None, None, got.data_ptr(), n, c, h, w, 0, gh, gw, ws.data_ptr(), ws.numel(),
                                     _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    assert not got.cpu().numpy().any()


def _guarded(shape, fill, dtype=torch.float32, guard=4096):
    """A tensor of `shape` in the middle of a larger allocation whose ends hold a sentinel: (view, check)."""
    size = int(np.prod(shape))
    buf = torch.full((guard + size + guard,), fill, dtype=dtype, device=dev())
    sentinel = 1234567 if dtype == torch.int32 else 12345.5
    buf[:guard] = sentinel
    buf[guard + size:] = sentinel

    def check():
        assert bool((buf[:guard] == sentinel).all()) and bool((buf[guard + size:] == sentinel).all()), "written outside the buffer"
    return buf[guard:guard + size].view(shape), check


@pytest.mark.parametrize("channels,height,width", [(40, 37, 53), (8, 200, 336)])
def test_roi_pool_and_roi_crop_stay_inside_their_buffers_on_non_finite_geometry(oracle_mod, channels, height, width):
    """NaN / Inf / 1e30 RoI corners, image indices and sampling-grid points (a diverged network; the reference converts them
    to int, which is undefined there): every round-6 RoIPool / RoICrop kernel returns, writes nothing outside its output
    (sentinel-guarded allocations), leaves the device healthy, and the rows of the WELL-FORMED RoIs / grids still equal the
    oracle bit for bit."""
    from detectron_pytorch_amd import _lib

    lib, stream = _lib.lib(), _lib.current_stream_handle(dev())
    n, c, h, w, scale, ph, pw = 2, channels, height, width, 1.0 / 8, 7, 7
    rng = np.random.RandomState(9)
    feat = syn.feature_map(n, c, h, w, seed=71)
    rois = syn.rois_adversarial(96, n, h, w, scale, seed=72)
    bad_values = [np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9]
    bad_rows = np.arange(0, 96, 2)
    for k, r in enumerate(bad_rows):
        cols = [1 + k % 4] if k < 28 else ([0] if k < 36 else [1, 2, 3, 4])
        for col in cols:
            rois[r, col] = bad_values[(k + col) % len(bad_values)]
    good = np.ones(96, bool)
    good[bad_rows] = False
    ref_out, ref_arg = oracle_mod.roi_pool_forward(feat, rois[good], ph, pw, scale, threads=8)
    f, d_rois = to_dev(feat), to_dev(rois)
    out, chk_out = _guarded((96, c, ph, pw), float("nan"))
    arg, chk_arg = _guarded((96, c, ph, pw), -7, dtype=torch.int32)
    rc = lib.mi_roi_pool_forward(f.data_ptr(), d_rois.data_ptr(), out.data_ptr(), arg.data_ptr(), n, c, h, w, 96, ph, pw, scale, stream)
    assert rc == 0, lib.mi_last_error()
    torch.cuda.synchronize()
    chk_out(), chk_arg()
    assert np.array_equal(out.cpu().numpy()[good], ref_out)
    assert np.array_equal(arg.cpu().numpy()[good], ref_arg)
    a = arg.cpu().numpy()
    assert ((a >= -1) & (a < n * c * h * w)).all(), "an argmax outside the input"
    gtop = to_dev(rng.randn(96, c, ph, pw).astype(np.float32))
    gin, chk_gin = _guarded((n, c, h, w), float("nan"))
    rc = lib.mi_roi_pool_backward(gtop.data_ptr(), d_rois.data_ptr(), arg.data_ptr(), gin.data_ptr(), n, c, h, w, 96, ph, pw, scale, stream)
    assert rc == 0, lib.mi_last_error()
    torch.cuda.synchronize()
    chk_gin()
    assert bool(torch.isfinite(gin).all()), "every element is overwritten with a finite sum"

    # RoICrop: grids with non-finite / huge points in every other RoI
    nrois = 24
    grid = syn.crop_grid(nrois, ph, pw, seed=73, span=1.2)
    bad = np.arange(0, nrois, 2)
    for k, r in enumerate(bad):
        m = rng.rand(ph, pw) < (1.0 if k % 3 == 0 else 0.3)
        grid[r, ..., k % 2][m] = bad_values[k % len(bad_values)]
    good = np.ones(nrois, bool)
    good[bad] = False
    d_grid = to_dev(grid)
    out, chk_out = _guarded((nrois, c, ph, pw), 123.0)
    rc = lib.mi_roi_crop_forward(f.data_ptr(), d_grid.data_ptr(), out.data_ptr(), n, c, h, w, nrois, ph, pw, stream)
    assert rc == 0, lib.mi_last_error()
    torch.cuda.synchronize()
    chk_out()
    # RoI r samples image r // (nrois / n): the oracle over ALL grids with the bad ones made harmless gives the good rows
    safe = grid.copy()
    safe[bad] = 5.0   # far outside: nothing written
    ref = oracle_mod.roi_crop_forward(feat, safe)
    ref_moved = oracle_mod.roi_crop_forward(feat + 1e3, safe)
    written = (ref_moved != ref)
    got = out.cpu().numpy()
    assert np.array_equal(got[good][written[good]], ref[good][written[good]])
    gt = to_dev(rng.randn(nrois, c, ph, pw).astype(np.float32))
    gin, chk_gin = _guarded((n, c, h, w), 0.0)
    rc = lib.mi_roi_crop_backward(f.data_ptr(), d_grid.data_ptr(), gt.data_ptr(), gin.data_ptr(), n, c, h, w, nrois, ph, pw, stream)
    assert rc == 0, lib.mi_last_error()
    torch.cuda.synchronize()
    chk_gin()
    gin2, chk_gin2 = _guarded((n, c, h, w), float("nan"))
    ws, chk_ws = _guarded((lib.mi_roi_crop_backward_workspace_bytes(nrois) // 4,), 0, dtype=torch.int32)
    rc = lib.mi_roi_crop_backward_ws(f.data_ptr(), d_grid.data_ptr(), gt.data_ptr(), gin2.data_ptr(), n, c, h, w, nrois, ph, pw,
                                     ws.data_ptr(), ws.numel() * 4, stream)
    assert rc == 0, lib.mi_last_error()
    torch.cuda.synchronize()
    chk_gin2(), chk_ws()
    assert not bool(torch.isnan(gin2).all()), "the tile form overwrites"
    assert float(torch.ones(4, device=dev()).sum()) == 4.0


# ---- RoICrop ---------------------------------------------------------------------------------------
def test_roi_crop_golden_and_oracle(oracle_mod):
    from detectron_pytorch_amd.roi_crop import RoICropFunction

    g = load_golden("roi_crop.npz")
    f = to_dev(g["feat"]).requires_grad_(True)
    grid = to_dev(g["grid"]).requires_grad_(True)
    out = RoICropFunction()(f, grid)
    assert np.array_equal(out.detach().cpu().numpy(), g["fwd"])
    out.backward(to_dev(g["gtop"]))
    assert_close(f.grad, g["bwd"], "roi_crop bwd")
    assert grid.grad is not None and not grid.grad.any()  # the reference never writes the grid gradient
    feat = syn.feature_map(3, 20, 33, 47, seed=15)
    grid_np = syn.crop_grid(12, 7, 7, seed=16)
    gtop = np.random.RandomState(17).randn(12, 20, 7, 7).astype(np.float32)
    f = to_dev(feat).requires_grad_(True)
    out = RoICropFunction()(f, to_dev(grid_np))
    assert np.array_equal(out.detach().cpu().numpy(), oracle_mod.roi_crop_forward(feat, grid_np))
    out.backward(to_dev(gtop))
    assert_close(f.grad, oracle_mod.roi_crop_backward(feat, grid_np, gtop), "roi_crop bwd oracle")


# ---- NMS: bit-exact kept indices ------------------------------------------------------------------------
def test_nms_golden_bit_exact():
    from detectron_pytorch_amd import nms as mi_nms

    g = load_golden("nms.npz")
    checked = 0
    for key in g.files:
        if not key.startswith("cython_"):
            continue
        name, t = key[len("cython_"):].rsplit("_t", 1)
        dets, thresh = g["dets_" + name], int(t) / 100.0
        keep = mi_nms.cython_nms(dets, thresh)
        assert keep.dtype == np.int64 and np.array_equal(keep, g[key]), key
        sorted_dets, _ = syn.sort_by_score(dets)
        keep_gpu = mi_nms.nms_gpu(to_dev(sorted_dets), thresh)
        assert keep_gpu.dtype == torch.int32 and keep_gpu.shape[1] == 1
        assert np.array_equal(keep_gpu.view(-1).cpu().numpy(), g["gpu_" + name + "_t" + t]), key
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize("gen", [syn.boxes_uniform, syn.boxes_clustered])
@pytest.mark.parametrize("n,thresh", [(1, 0.5), (2, 0.5), (63, 0.5), (64, 0.7), (65, 0.3), (1000, 0.5), (1000, 0.7),
                                      (2000, 0.7), (4096, 0.7), (4097, 0.5), (6000, 0.7), (12000, 0.7)])
def test_nms_vs_oracle_bit_exact(oracle_mod, gen, n, thresh):
    from detectron_pytorch_amd import nms as mi_nms

    dets = gen(n, seed=n + 1)
    keep = mi_nms.cython_nms(dets, thresh)
    ref_keep = oracle_mod.nms_cython(dets, thresh)
    assert np.array_equal(keep, ref_keep), "GE_ORIG_ASC n=%d" % n
    if n <= 6000:
        sorted_dets, _ = syn.sort_by_score(dets)
        keep_gpu = mi_nms.nms_gpu(to_dev(sorted_dets), thresh).view(-1).cpu().numpy()
        assert np.array_equal(keep_gpu, oracle_mod.nms_gpu_semantics(sorted_dets, thresh)), "GT_SORTED_POS n=%d" % n
    # size-independent properties: ascending output, idempotence (NMS of the kept set keeps everything)
    assert (np.diff(keep) > 0).all()
    again = mi_nms.cython_nms(dets[keep], thresh)
    assert np.array_equal(again, np.arange(len(keep)))


def test_nms_hand_cases_and_threshold_equality():
    from detectron_pytorch_amd import nms as mi_nms

    d = np.array([[0, 0, 9, 9, 0.9], [9, 0, 18, 9, 0.8]], np.float32)
    iou = float(np.float32(10.0) / np.float32(190.0))
    assert mi_nms.cython_nms(d, iou).tolist() == [0]  # >= at equality (cython_nms.pyx:84)
    assert mi_nms.nms_gpu(to_dev(d), iou).view(-1).tolist() == [0, 1]  # strict > (nms_cuda_kernel.cu:78)
    d = np.array([[100, 100, 120, 120, 0.1], [0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.9]], np.float32)
    assert mi_nms.cython_nms(d, 0.5).tolist() == [0, 2]
    assert mi_nms.cython_nms(np.zeros((0, 5), np.float32), 0.5).tolist() == []
    assert mi_nms.nms(torch.zeros(0, 5, device=dev()), 0.5) == []  # nms_wrapper.py:13-14
    # tied scores: documented rule = higher original index first
    d = np.array([[0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.5], [50, 50, 60, 60, 0.5]], np.float32)
    assert mi_nms.cython_nms(d, 0.5).tolist() == [1, 2]
    keep, num = mi_nms.nms_device(to_dev(d), 0.5)
    assert keep.is_cuda and num.is_cuda and int(num) == 2


def test_nms_is_asynchronous_on_the_current_stream():
    from detectron_pytorch_amd import nms as mi_nms

    dets = to_dev(syn.boxes_clustered(2000, seed=3))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep, num = mi_nms.nms_device(dets, 0.7)
    s.synchronize()
    ref, refnum = mi_nms.nms_device(dets, 0.7)
    torch.cuda.synchronize()
    assert int(num) == int(refnum) and torch.equal(keep[:int(num)], ref[:int(refnum)])


def test_bbox_overlaps_bit_exact(oracle_mod):
    from detectron_pytorch_amd import nms as mi_nms

    g = load_golden("bbox_overlaps.npz")
    assert np.array_equal(mi_nms.bbox_overlaps(g["boxes"], g["query"]), g["overlaps"])
    boxes = syn.boxes_uniform(2000, seed=8)[:, :4]
    query = syn.boxes_clustered(8, seed=9)[:, :4]
    assert np.array_equal(mi_nms.bbox_overlaps(boxes, query), oracle_mod.bbox_overlaps(boxes, query))


def test_nms_many_matches_serial(oracle_mod):
    """nms_device_many (independent problems in one batched call, or fanned out over HIP streams) == serial calls."""
    from detectron_pytorch_amd import nms as mi_nms

    # batched entry point (every problem <= 4096 boxes, incl. an empty one and 40 problems > one table of 32)
    sets = [syn.boxes_clustered(500 + 137 * i, seed=20 + i) for i in range(11)] + [np.zeros((0, 5), np.float32)]
    sets += [syn.boxes_uniform(64 * i + 1, seed=i) for i in range(28)]
    outs = mi_nms.nms_device_many([to_dev(d) for d in sets], 0.7)
    torch.cuda.synchronize()
    for d, (keep, num) in zip(sets, outs):
        k = keep[:int(num.item())].cpu().numpy()
        assert np.array_equal(k, oracle_mod.nms_cython(d, 0.7) if len(d) else np.zeros(0, np.int64))
    # GT_SORTED_POS through the batched path
    from detectron_pytorch_amd import _lib
    srt = [syn.sort_by_score(d)[0] for d in sets[:5]]
    outs = mi_nms.nms_device_many([to_dev(d) for d in srt], 0.7, _lib.NMS_GT_SORTED_POS)
    for d, (keep, num) in zip(srt, outs):
        assert np.array_equal(keep[:int(num.item())].cpu().numpy(), oracle_mod.nms_gpu_semantics(d, 0.7))
    # stream fan-out (a problem above the batched limit forces it)
    big = [syn.boxes_clustered(5000, seed=3), syn.boxes_clustered(700, seed=4)]
    outs = mi_nms.nms_device_many([to_dev(d) for d in big], 0.7)
    torch.cuda.synchronize()
    for d, (keep, num) in zip(big, outs):
        assert np.array_equal(keep[:int(num.item())].cpu().numpy(), oracle_mod.nms_cython(d, 0.7))


# ---- roi_feature_transform: the caller of the RoI operators (model_builder.py:252-324) ---------------------------
def _fpn_inputs(channels=32, num_rois=300, batch=2, seed=41):
    from detectron_pytorch_amd import roi_xform

    rois, lvls = syn.rois_fpn_distributed(num_rois, batch=batch, seed=seed)
    blobs = roi_xform.add_multilevel_roi_blobs({}, "rois", rois, lvls, 2, 5)
    scales = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4]          # coarsest level first, as the reference orders them
    feats = [syn.feature_map(batch, channels, int(np.ceil(syn.IM_H * s)), int(np.ceil(syn.IM_W * s)), seed=seed + i)
             for i, s in enumerate(scales)]
    return rois, lvls, blobs, scales, feats


def test_roi_feature_transform_fpn_roi_align_vs_oracle(oracle_mod):
    from detectron_pytorch_amd import roi_xform

    rois, lvls, blobs, scales, feats = _fpn_inputs()
    dev_feats = [to_dev(f).requires_grad_(True) for f in feats]
    out = roi_xform.roi_feature_transform(dev_feats, blobs, "rois", "RoIAlign", 7, scales, 2)
    assert out.shape == (rois.shape[0], feats[0].shape[1], 7, 7)
    gtop = np.random.RandomState(5).randn(*out.shape).astype(np.float32)
    out.backward(to_dev(gtop))
    expected = np.empty(out.shape, np.float32)
    for lvl in range(2, 6):
        idx = np.nonzero(lvls == lvl)[0]
        if idx.size == 0:
            continue
        feat, sc = feats[5 - lvl], scales[5 - lvl]
        expected[idx] = oracle_mod.roi_align_forward(feat, rois[idx], 7, 7, sc, 2)
        grad = oracle_mod.roi_align_backward(gtop[idx], rois[idx], feat.shape, sc, 2)
        assert_close(dev_feats[5 - lvl].grad, grad, "fpn level %d grad" % lvl)
    assert_fwd(out, expected, "fpn roi_feature_transform", exact=False)


@pytest.mark.parametrize("why", ["direct", "no_ws", "fused_off", "roi_pool"])
def test_roi_feature_transform_level_vector_without_the_fused_call(oracle_mod, tuning_env, why):
    """The device-side producers (rcnn.targets, fpn_proposals, inference) hand over `rois` + `rois_levels` only.  When the
    fused FPN call declines (MI_ROI_ALIGN_IMPL=direct, MI_ROI_ALIGN_NO_WS, fused=False) or the method is not RoIAlign, the
    levels are split on the spot: same result as the fused call, gradients on every level, padding rows (image -1, level
    out of range) give zeros -- instead of the KeyError('rois_fpn2') of round 2."""
    from detectron_pytorch_amd import roi_xform

    rois, lvls, _, scales, feats = _fpn_inputs(num_rois=200)
    rois = np.concatenate([rois, np.array([[-1, 0, 0, 0, 0], [-1, 5, 5, 50, 50]], np.float32)])   # padding rows
    lvls = np.concatenate([lvls, np.array([2, 9], lvls.dtype)])
    blobs = {"rois": to_dev(rois), "rois_levels": to_dev(lvls.astype(np.int64))}
    method = "RoIPoolF" if why == "roi_pool" else "RoIAlign"
    if why == "direct":
        tuning_env(MI_ROI_ALIGN_IMPL="direct")
    elif why == "no_ws":
        tuning_env(MI_ROI_ALIGN_NO_WS="1")
    dev_feats = [to_dev(f).requires_grad_(True) for f in feats]
    out = roi_xform.roi_feature_transform(dev_feats, blobs, "rois", method, 7, scales, 2, fused=(why != "fused_off"))
    assert out.shape == (rois.shape[0], feats[0].shape[1], 7, 7)
    out.sum().backward()
    got = out.detach().cpu().numpy()
    assert (got[-2:] == 0).all()
    for lvl in range(2, 6):
        idx = np.nonzero(lvls[:-2] == lvl)[0]
        feat, sc = feats[5 - lvl], scales[5 - lvl]
        if method == "RoIAlign":
            want = oracle_mod.roi_align_forward(feat, rois[idx], 7, 7, sc, 2)
            assert np.abs(got[idx] - want).max() <= FAST_ATOL
        else:
            want, _ = oracle_mod.roi_pool_forward(feat, rois[idx], 7, 7, sc)
            assert np.array_equal(got[idx], want)
        assert dev_feats[5 - lvl].grad is not None and float(dev_feats[5 - lvl].grad.abs().sum()) > 0


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_fpn_level_index_from_restore_equals_the_host_expression(dtype):
    """mi_fpn_level_index_from_restore: the map index of every RoI of a pyramid blob in dataloader order, from the reference's
    restore index (utils/fpn.py:31-58) and the levels' row counts -- against np.repeat(values, counts)[restore], with empty
    levels, int32 and int64 indices."""
    import ctypes

    from detectron_pytorch_amd import _lib

    lib = _lib.lib()
    rng = np.random.RandomState(3)
    for counts in ([561, 181, 186, 72], [0, 40, 0, 9], [5, 0, 0, 0], [1, 1, 1, 1]):
        values = [3, 2, 1, 0]
        n = sum(counts)
        restore = rng.permutation(n).astype(dtype)
        want = np.repeat(np.array(values, np.int32), counts)[restore]
        out = torch.full((n,), -7, dtype=torch.int32, device=dev())
        rc = lib.mi_fpn_level_index_from_restore(to_dev(restore).data_ptr(), 1 if dtype == np.int64 else 0, n, 4,
                                                 (ctypes.c_int * 4)(*counts), (ctypes.c_int * 4)(*values), out.data_ptr(),
                                                 _lib.current_stream_handle(dev()))
        assert rc == 0, lib.mi_last_error()
        assert np.array_equal(out.cpu().numpy(), want)
    assert lib.mi_fpn_level_index_from_restore(None, 0, 4, 9, None, None, None, None) != 0   # more than 8 spans: refused


@pytest.mark.parametrize("method", ["RoIPoolF", "RoICrop", "RoIAlign"])
def test_roi_feature_transform_single_level_methods(oracle_mod, method):
    from detectron_pytorch_amd import roi_xform

    feat = syn.feature_map(2, 16, 38, 50, seed=3)
    rois = syn.rois_adversarial(40, 2, 38, 50, 1.0 / 16, seed=4)
    rois[:, 0] = np.clip(rois[:, 0], 0, 1)
    out = roi_xform.roi_feature_transform(to_dev(feat), {"rois": rois}, "rois", method, 7, 1.0 / 16, 2)
    if method == "RoIAlign":
        assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, 7, 7, 1.0 / 16, 2), method, exact=False)
    elif method == "RoIPoolF":
        assert np.array_equal(out.cpu().numpy(), oracle_mod.roi_pool_forward(feat, rois, 7, 7, 1.0 / 16)[0])
    else:
        grid_xy = roi_xform.affine_grid_gen(torch.from_numpy(rois), feat.shape[2:], 14)
        grid_yx = torch.stack([grid_xy[..., 1], grid_xy[..., 0]], 3).contiguous().numpy()
        crop = torch.from_numpy(oracle_mod.roi_crop_forward(feat, grid_yx))
        expected = torch.nn.functional.max_pool2d(crop, 2, 2).numpy()
        assert out.shape == (40, 16, 7, 7)
        assert_close(out, expected, method)   # the grid itself is computed by torch on two devices


# ---- Soft-NMS (utils.cython_nms.soft_nms) -----------------------------------------------------------------------
def test_soft_nms_golden_bit_exact():
    """Rows, row order, re-scored scores and original indices as the reference's cython build produced them."""
    from detectron_pytorch_amd import nms as mi_nms

    g = load_golden("soft_nms.npz")
    cfgs, checked = g["cfgs"], 0
    for key in g.files:
        if not key.startswith("boxes_"):
            continue
        tag = key[len("boxes_"):]
        name, m, c = tag.rsplit("_", 2)
        sigma, nt, th = (float(v) for v in cfgs[int(c[1:])])
        boxes, inds = mi_nms.soft_nms(g["dets_" + name], sigma, nt, th, int(m[1:]))
        assert boxes.dtype == np.float32 and inds.dtype == np.int64
        assert np.array_equal(inds, g["inds_" + tag]), tag
        assert np.array_equal(boxes, g[key]), tag
        checked += 1
    assert checked == 27


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("gen,n", [(syn.boxes_uniform, 1), (syn.boxes_uniform, 2), (syn.boxes_clustered, 64),
                                   (syn.boxes_uniform, 65), (syn.boxes_clustered, 257), (syn.boxes_uniform, 1000),
                                   (syn.boxes_clustered, 2000), (syn.boxes_uniform, 4096)])
def test_soft_nms_vs_oracle_bit_exact(oracle_mod, gen, n, method):
    from detectron_pytorch_amd import nms as mi_nms

    dets = gen(n, seed=n + method)
    for sigma, nt, th in ((0.5, 0.3, 0.001), (0.4, 0.5, 0.1)):
        ob, oi = oracle_mod.soft_nms(dets, sigma, nt, th, method)
        boxes, inds = mi_nms.soft_nms(dets, sigma, nt, th, method)
        assert np.array_equal(inds, oi), "indices n=%d method=%d" % (n, method)
        assert np.array_equal(boxes, ob), "rows n=%d method=%d" % (n, method)


def test_soft_nms_contract_and_wrapper():
    from detectron_pytorch_amd import nms as mi_nms

    out_dets, out_inds, num = mi_nms.soft_nms_device(torch.zeros((0, 5), device=dev()))
    assert int(num.item()) == 0
    dets = np.array([[0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.9], [20, 20, 29, 29, 0.7]], np.float32)
    boxes, inds = mi_nms.soft_nms(dets, 0.5, 0.3, 0.001, 0)
    assert inds.tolist() == [1, 2]
    boxes, keep = mi_nms.box_utils_soft_nms(dets, method="linear")            # utils/boxes.py:327-344
    assert keep.tolist() == [1, 2] and boxes.shape == (2, 5)
    assert mi_nms.box_utils_soft_nms(np.zeros((0, 5), np.float32))[1] == []
    with pytest.raises(AssertionError):
        mi_nms.box_utils_soft_nms(dets, method="nope")
    with pytest.raises(RuntimeError):
        mi_nms.soft_nms(syn.boxes_uniform(4097, seed=0))
    # tensor in -> tensors out, asynchronous entry point leaves everything on the device
    t = to_dev(syn.boxes_uniform(300, seed=3))
    b, i = mi_nms.soft_nms(t, 0.5, 0.3, 0.001, 2)
    assert b.is_cuda and i.is_cuda and i.dtype == torch.int64


# ---- RoIAlign: shapes at the edges of the fast paths ---------------------------------------------------------------
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("shape,nrois", [((1, 32, 1, 40), 40), ((1, 32, 30, 1), 40), ((2, 32, 2, 2), 30),
                                         ((1, 32, 25, 42), 9000)])
def test_roi_align_degenerate_maps_and_many_rois(oracle_mod, shape, nrois, channels_last):
    """One-pixel-high / -wide maps (no bilinear neighbour: the records cannot describe them) and more RoIs than the
    record path ranks (8192): both must fall through to the paths that keep the reference arithmetic."""
    n, c, h, w = shape
    scale = 1.0 / 16
    feat = syn.feature_map(n, c, h, w, seed=5)
    rois = syn.rois_adversarial(nrois, n, max(h, 2), max(w, 2), scale, seed=nrois)
    gtop = np.random.RandomState(9).randn(nrois, c, 7, 7).astype(np.float32)
    out, grad = _roi_align_gpu(feat, rois, 7, scale, 2, gtop, channels_last=channels_last)
    assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, 7, 7, scale, 2, threads=8), "fwd", exact=False)
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, 2, threads=8), "bwd")


# ---- per-class detection post-processing on the device (core/test.py:732-790) -------------------------------------
@pytest.mark.parametrize("name", ["c21", "c81"])
@pytest.mark.parametrize("tag,soft,method", [("hard", False, "linear"), ("linear", True, "linear"),
                                             ("gaussian", True, "gaussian")])
def test_detection_postprocess_golden_bit_exact(name, tag, soft, method):
    """Same rows, same order, same bits as the reference's own function (fixture from its source), classes batched."""
    from detectron_pytorch_amd import detection

    g = load_golden("detection.npz")
    key = "%s_%s" % (name, tag)
    s, b, cls_boxes = detection.box_results_with_nms_and_limit(g["scores_" + name], g["boxes_" + name], soft_nms=soft,
                                                               soft_nms_method=method)
    assert s.dtype == np.float32 and b.dtype == np.float32 and cls_boxes[0] == []
    assert np.array_equal(np.array([len(c) for c in cls_boxes]), g["cls_counts_" + key])
    assert np.array_equal(np.vstack(cls_boxes[1:]), g["cls_rows_" + key])
    assert np.array_equal(s, g["out_scores_" + key]) and np.array_equal(b, g["out_boxes_" + key])


@pytest.mark.parametrize("method", ["ID", "TEMP_AVG", "AVG", "IOU_AVG", "GENERALIZED_AVG", "QUASI_SUM"])
def test_box_voting_golden(method):
    """mi_box_voting against the reference's own utils.boxes.box_voting (fixture from its source text): the voter set is
    exact (IoU bit for bit), boxes and scores within 2e-6 relative (fp64 accumulation here, fp32 pairwise sums there)."""
    from detectron_pytorch_amd import detection

    g = load_golden("box_voting.npz")
    out = detection.box_voting(g["top_dets"], g["all_dets"], float(g["thresh"]), scoring_method=method, beta=float(g["beta"]))
    ref = g["voted_" + method]
    assert out.dtype == np.float32 and out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-6 * np.abs(ref).max(), np.abs(out - ref).max()
    moved = np.abs(ref[:, :4] - g["top_dets"][:, :4]).max(axis=1) > 1e-3
    assert np.array_equal(np.abs(out[:, :4] - g["top_dets"][:, :4]).max(axis=1) > 1e-3, moved)   # same rows moved
    loose = detection.box_voting(to_dev(g["top_dets"]), to_dev(g["all_dets"]), 0.5)
    assert np.abs(loose.cpu().numpy() - g["voted_loose_ID"]).max() <= 2e-6 * np.abs(ref).max()
    with pytest.raises(NotImplementedError):
        detection.box_voting(g["top_dets"], g["all_dets"], 0.8, scoring_method="MEDIAN")


@pytest.mark.parametrize("tag,soft,method", [("hard_ID", False, "ID"), ("linear_IOU_AVG", True, "IOU_AVG")])
def test_detection_postprocess_with_box_voting(oracle_mod, tag, soft, method):
    """core/test.py:732-790 with TEST.BBOX_VOTE.ENABLED: all classes voted in one call between the NMS and the
    detections_per_im cut; against the reference's function (fixture) and against the oracle on a second input."""
    from detectron_pytorch_amd import detection
    from oracle import postprocess

    g = load_golden("box_voting.npz")
    s, b, cls_boxes = detection.box_results_with_nms_and_limit(g["det_in_scores"], g["det_in_boxes"], soft_nms=soft,
                                                               bbox_vote=True, bbox_vote_method=method)
    assert np.array_equal(np.array([len(c) for c in cls_boxes]), g["det_counts_" + tag])
    assert np.abs(s - g["det_scores_" + tag]).max() <= 2e-6 and np.abs(b - g["det_boxes_" + tag]).max() <= 2e-3
    scores, boxes = syn.detection_head_outputs(500, 81, seed=9)
    rs, rb, rcls = postprocess.box_results_with_nms_and_limit(scores, boxes, soft_nms=soft, bbox_vote=True,
                                                              bbox_vote_thresh=0.6, bbox_vote_method=method)
    s, b, cls_boxes = detection.box_results_with_nms_and_limit(to_dev(scores), to_dev(boxes), soft_nms=soft, bbox_vote=True,
                                                               bbox_vote_thresh=0.6, bbox_vote_method=method)
    assert [len(c) for c in cls_boxes] == [len(c) for c in rcls]
    assert np.abs(s.cpu().numpy() - rs).max() <= 2e-6 and np.abs(b.cpu().numpy() - rb).max() <= 2e-3


@pytest.mark.parametrize("soft", [False, True])
def test_detection_postprocess_vs_oracle(oracle_mod, soft):
    from detectron_pytorch_amd import detection
    from oracle import postprocess

    for rois, classes, limit, seed in ((1000, 81, 100, 7), (60, 11, 0, 1), (5, 3, 100, 2)):
        scores, boxes = syn.detection_head_outputs(rois, classes, seed=seed)
        scores[:, 2] = 0.0                                   # an empty class
        want = postprocess.box_results_with_nms_and_limit(scores, boxes, detections_per_im=limit, soft_nms=soft)
        got = detection.box_results_with_nms_and_limit(scores, boxes, detections_per_im=limit, soft_nms=soft)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        for j in range(1, classes):
            assert got[2][j].shape == want[2][j].shape and np.array_equal(got[2][j], want[2][j]), (rois, classes, j)
    # tensors in -> tensors out, on the device
    s, b, c = detection.box_results_with_nms_and_limit(to_dev(scores), to_dev(boxes), soft_nms=soft)
    assert s.is_cuda and b.is_cuda and c[1].is_cuda
    # nothing above the score threshold at all
    s, b, c = detection.box_results_with_nms_and_limit(np.zeros((4, 3), np.float32), np.zeros((4, 12), np.float32),
                                                       soft_nms=soft)
    assert s.shape == (0,) and b.shape == (0, 4) and all(x.shape == (0, 5) for x in c[1:])


@pytest.mark.parametrize("rois,classes,seed", [(1000, 81, 3), (65, 4, 1), (1, 3, 2), (4096, 2, 5), (300, 21, 9)])
def test_nms_segmented_matches_the_per_class_loop(oracle_mod, rois, classes, seed):
    """mi_nms_segmented on the blobs in place == `np.where(scores[:, j] > thresh)` + cython NMS per class
    (core/test.py:748-771): kept flags per (class, RoI), bit-exact, including tied and NaN scores and empty classes."""
    from detectron_pytorch_amd import nms as mi_nms

    scores, boxes = syn.detection_head_outputs(rois, classes, seed=seed)
    if classes > 2:
        scores[:, 2] = 0.0                                            # an empty class
    if rois >= 64:
        scores[5:25, 1] = scores[5, 1]                                # tied scores inside a class
        scores[30, 1] = np.nan                                        # a diverged RoI takes no part
    kept, num = mi_nms.nms_segmented(to_dev(scores), to_dev(boxes), 0.05, 0.5)
    kept, num = kept.cpu().numpy(), num.cpu().numpy()
    assert kept.shape == (classes - 1, rois) and set(np.unique(kept)) <= {0, 1}
    for j in range(1, classes):
        inds = np.where(scores[:, j] > 0.05)[0]
        want = np.zeros(rois, np.int32)
        if len(inds):
            dets_j = np.hstack([boxes[inds, 4 * j:4 * j + 4], scores[inds, j:j + 1]]).astype(np.float32)
            want[inds[oracle_mod.nms_cython(dets_j, 0.5)]] = 1
        assert np.array_equal(kept[j - 1], want), "class %d" % j
        assert num[j - 1] == want.sum()


def test_nms_segmented_rejects_what_it_cannot_do():
    from detectron_pytorch_amd import _lib

    lib = _lib.lib()
    x = torch.zeros(16, device=dev())
    rc = lib.mi_nms_segmented(x.data_ptr(), 4, 8, x.data_ptr(), 1, 2, 1, 5000, 0.05, 0.5, x.data_ptr(), x.data_ptr(),
                              None, x.data_ptr(), 1 << 30, None)
    assert rc != 0 and b"4096" in lib.mi_last_error()
    rc = lib.mi_nms_segmented(x.data_ptr(), 4, 8, x.data_ptr(), 1, 2, 1, 4, 0.05, 0.5, x.data_ptr(), x.data_ptr(),
                              None, x.data_ptr(), 16, None)
    assert rc != 0 and b"workspace" in lib.mi_last_error()


def test_detection_static_result_and_tie_overflow(oracle_mod):
    """box_results_static delivers the reference's rows without a host round trip; more ties at the top-100 cut than its
    fixed-size result holds are detected (total > count) and the wrapper falls back to the compacting path."""
    from detectron_pytorch_amd import detection
    from oracle import postprocess

    scores, boxes = syn.detection_head_outputs(600, 21, seed=4)
    valid = np.ones(600, bool)
    valid[500:] = False                                              # padding rows of a static RoI blob
    res = detection.box_results_static(to_dev(scores), to_dev(boxes), 0.05, 0.5, 100, roi_valid=to_dev(valid))
    want = postprocess.box_results_with_nms_and_limit(scores[:500], boxes[:500], detections_per_im=100)
    count = int(res["count"])
    assert count == int(res["total"]) == len(want[0])
    assert np.array_equal(res["dets"][:count].cpu().numpy(), np.vstack([c for c in want[2][1:] if len(c)]))
    assert np.array_equal(res["class_counts"].cpu().numpy(), [len(c) for c in want[2][1:]])
    assert not bool(res["cls"][count:].any()) and float(res["dets"][count:].abs().sum()) == 0
    # 200 far-apart boxes with one and the same score: the reference keeps all of them (scores >= image_thresh)
    n = 200
    scores = np.zeros((n, 3), np.float32)
    scores[:, 1] = 0.5
    scores[:7, 2] = np.linspace(0.9, 0.6, 7)
    xy = (np.arange(n) * 40).astype(np.float32)
    one = np.stack([xy, xy, xy + 10, xy + 10], axis=1)
    boxes = np.tile(one, (1, 3)).astype(np.float32)
    res = detection.box_results_static(to_dev(scores), to_dev(boxes), 0.05, 0.5, 100)
    assert int(res["total"]) == 207 and int(res["count"]) == 100 + detection.TIE_SLACK
    want = postprocess.box_results_with_nms_and_limit(scores, boxes, detections_per_im=100)
    got = detection.box_results_with_nms_and_limit(scores, boxes, detections_per_im=100)
    assert len(want[0]) == 207 and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("soft,method,vote,vote_method", [(True, "linear", False, "ID"), (True, "gaussian", False, "ID"),
                                                          (False, "linear", True, "ID"), (False, "linear", True, "IOU_AVG"),
                                                          (True, "linear", True, "TEMP_AVG"), (True, "hard", True, "QUASI_SUM")])
def test_detection_static_path_with_soft_nms_and_voting_equals_the_dynamic_path(soft, method, vote, vote_method):
    """box_results_static_general (TEST.SOFT_NMS / TEST.BBOX_VOTE without a host round trip: what DetectionGraph captures)
    delivers the rows of box_results_with_nms_and_limit -- itself pinned against the reference's function by the goldens
    above -- bit for bit, padding rows of a static RoI blob included; with and without the detections_per_im cut."""
    from detectron_pytorch_amd import detection

    scores, boxes = syn.detection_head_outputs(600, 81, seed=12)
    valid = np.ones(600, bool)
    valid[520:] = False
    for dets_per_im in (100, 0, 7):
        kw = dict(score_thresh=0.05, nms_thresh=0.5, detections_per_im=dets_per_im, soft_nms=soft, soft_nms_method=method,
                  bbox_vote=vote, bbox_vote_thresh=0.7, bbox_vote_method=vote_method)
        ws, wb, wcls = detection.box_results_with_nms_and_limit(to_dev(scores[:520]), to_dev(boxes[:520]), **kw)
        res = detection.box_results_static_general(to_dev(scores), to_dev(boxes), roi_valid=to_dev(valid), **kw)
        count, total = int(res["count"]), int(res["total"])
        assert total == ws.numel(), (dets_per_im, total, ws.numel())
        cap = res["dets"].size(0)
        assert count == min(total, cap)
        want = torch.cat([ws.view(-1, 1)[:0].expand(0, 5)] + [c for c in wcls[1:] if len(c)]) if total else torch.zeros((0, 5))
        assert torch.equal(res["dets"][:count].cpu(), want[:count].cpu()), dets_per_im
        if total <= cap:
            assert torch.equal(res["class_counts"].cpu(), torch.tensor([len(c) for c in wcls[1:]]))
            assert torch.equal(res["cls"][:count].cpu().long(),
                               torch.cat([torch.full((len(c),), j + 1, dtype=torch.long) for j, c in enumerate(wcls[1:])]))
        assert not bool(res["cls"][count:].any()) and float(res["dets"][count:].abs().sum()) == 0
        out = detection._results_from_static(res, False)
        if total <= cap:
            assert torch.equal(out[0], ws) and torch.equal(out[1], wb)
        else:
            assert out is None                                   # the caller then takes the dynamic path


@pytest.mark.parametrize("soft,vote", [(True, False), (False, True), (True, True)])
def test_detection_static_general_path_degenerate_inputs(soft, vote):
    """No score above the threshold (every class segment empty), a single candidate, and a blob whose rows are all padding:
    the static path returns zero rows / the one row without touching memory behind its buffers, as the dynamic path does."""
    from detectron_pytorch_amd import detection

    scores, boxes = syn.detection_head_outputs(64, 81, seed=3)
    kw = dict(nms_thresh=0.5, detections_per_im=100, soft_nms=soft, soft_nms_method="linear", bbox_vote=vote,
              bbox_vote_thresh=0.8, bbox_vote_method="ID")
    res = detection.box_results_static_general(to_dev(scores), to_dev(boxes), score_thresh=2.0, **kw)   # nothing passes
    assert int(res["count"]) == 0 and int(res["total"]) == 0 and float(res["dets"].abs().sum()) == 0
    assert not bool(res["class_counts"].any())
    one = np.zeros_like(scores)
    one[:, 0] = 1.0
    one[17, 5] = 0.9                                                 # exactly one candidate, class 5
    one[17, 0] = 0.1
    ws, wb, wcls = detection.box_results_with_nms_and_limit(to_dev(one), to_dev(boxes), score_thresh=0.5, **kw)
    res = detection.box_results_static_general(to_dev(one), to_dev(boxes), score_thresh=0.5, **kw)
    assert int(res["count"]) == 1 == ws.numel() and int(res["cls"][0]) == 5
    assert torch.equal(res["dets"][:1].cpu(), wcls[5].cpu())
    res = detection.box_results_static_general(to_dev(scores), to_dev(boxes), score_thresh=0.05,
                                               roi_valid=to_dev(np.zeros(64, bool)), **kw)     # all rows are padding
    assert int(res["count"]) == 0 and int(res["total"]) == 0


def test_soft_nms_segmented_matches_single_calls(oracle_mod):
    from detectron_pytorch_amd import _lib

    lib = _lib.lib()
    parts = [syn.boxes_uniform(n, seed=n) for n in (300, 0, 1, 65, 700)]
    dets = to_dev(np.vstack(parts))
    offsets = to_dev(np.cumsum([0] + [len(p) for p in parts]).astype(np.int32))
    out_dets, out_inds = torch.empty_like(dets), torch.empty(dets.size(0), dtype=torch.int64, device=dev())
    num_out = torch.zeros(len(parts), dtype=torch.int32, device=dev())
    rc = lib.mi_soft_nms_segmented(dets.data_ptr(), offsets.data_ptr(), len(parts), 700, 0.5, 0.3, 0.01, 2,
                                   out_dets.data_ptr(), out_inds.data_ptr(), num_out.data_ptr(),
                                   _lib.current_stream_handle(dev()))
    assert rc == 0
    off = offsets.cpu().numpy()
    for p, part in enumerate(parts):
        boxes, inds = oracle_mod.soft_nms(part, 0.5, 0.3, 0.01, 2) if len(part) else (np.zeros((0, 5), np.float32), [])
        k = int(num_out[p].item())
        assert k == len(boxes)
        assert np.array_equal(out_dets[off[p]:off[p] + k].cpu().numpy(), boxes)
        assert np.array_equal(out_inds[off[p]:off[p] + k].cpu().numpy(), np.asarray(inds, dtype=np.int64))


def test_roi_align_fpn_fused_matches_per_level_loop_and_oracle(oracle_mod):
    """mi_roi_align_forward_fpn / _backward_fpn: all levels in one call, output in dataloader order, a level without
    RoIs gets a zero gradient; same numbers as the per-level loop (same per-RoI arithmetic, same summation order)."""
    from detectron_pytorch_amd import roi_xform
    from detectron_pytorch_amd.roi_align import roi_align_fpn, roi_align_fpn_supported

    rois, lvls, blobs, scales, feats = _fpn_inputs(channels=64, num_rois=500, batch=2, seed=11)
    gtop = np.random.RandomState(2).randn(500, 64, 7, 7).astype(np.float32)
    outs, grads = [], []
    for fused in (True, False):
        dev_feats = [to_dev(f).requires_grad_(True) for f in feats]
        out = roi_xform.roi_feature_transform(dev_feats, blobs, "rois", "RoIAlign", 7, scales, 2, fused=fused)
        out.backward(to_dev(gtop))
        outs.append(out.detach())
        grads.append([f.grad for f in dev_feats])
    assert roi_align_fpn_supported([to_dev(f) for f in feats], 500, 7, 7)
    assert torch.equal(outs[0], outs[1])
    # with the un-split blob present (as in the reference's rpn_ret) it is used directly; device-tensor blobs too
    with torch.no_grad():
        blobs_dev = {k: to_dev(v) for k, v in dict(blobs, rois=rois).items()}
        assert torch.equal(roi_xform.roi_feature_transform([to_dev(f) for f in feats], dict(blobs, rois=rois), "rois",
                                                           "RoIAlign", 7, scales, 2), outs[0])
        assert torch.equal(roi_xform.roi_feature_transform([to_dev(f) for f in feats], blobs_dev, "rois", "RoIAlign", 7,
                                                           scales, 2), outs[0])
    for g_fused, g_loop in zip(grads[0], grads[1]):
        assert_close(g_fused, g_loop.cpu().numpy(), "fused vs loop grad")
    for lvl in range(2, 6):                                  # and against the oracle, level by level
        idx = np.nonzero(lvls == lvl)[0]
        feat, sc = feats[5 - lvl], scales[5 - lvl]
        assert_fwd(outs[0][torch.from_numpy(idx).to(dev())], oracle_mod.roi_align_forward(feat, rois[idx], 7, 7, sc, 2),
                   "fused fwd lvl %d" % lvl, exact=False)
        assert_close(grads[0][5 - lvl], oracle_mod.roi_align_backward(gtop[idx], rois[idx], feat.shape, sc, 2, threads=8),
                     "fused bwd lvl %d" % lvl)
    # a level nobody maps to: its gradient is all zeros; adversarial RoIs take the in-kernel reference path
    adv = np.vstack([syn.rois_adversarial(60, 2, 50, 84, 1.0 / 16, seed=3), rois[:40]])
    adv_lvl = np.concatenate([np.full(60, 2, np.int32), (5 - lvls[:40]).astype(np.int32)])   # index 2 = P3 ... none on P5
    adv_lvl[adv_lvl == 0] = 1
    dev_feats = [to_dev(f).requires_grad_(True) for f in feats]
    out = roi_align_fpn(dev_feats, scales, to_dev(adv), to_dev(adv_lvl), 7, 7, 2)
    g2 = np.random.RandomState(4).randn(100, 64, 7, 7).astype(np.float32)
    out.backward(to_dev(g2))
    assert float(dev_feats[0].grad.abs().max()) == 0.0
    for k in (1, 2, 3):
        idx = np.nonzero(adv_lvl == k)[0]
        assert_fwd(out.detach()[torch.from_numpy(idx).to(dev())],
                   oracle_mod.roi_align_forward(feats[k], adv[idx], 7, 7, scales[k], 2), "adv fwd %d" % k, exact=False)
        assert_close(dev_feats[k].grad, oracle_mod.roi_align_backward(g2[idx], adv[idx], feats[k].shape, scales[k], 2),
                     "adv bwd %d" % k)


@pytest.mark.parametrize("channels_last", [False, True])
def test_records_written_by_the_proposal_stage_equal_the_pooling_calls_own(channels_last):
    """mi_rpn_collect_finish_records: the last launch of the proposal stage writes the RoI blob (bit-equal to
    mi_rpn_collect_finish) AND the RoIAlign records of its rows; mi_roi_align_forward_fpn_records over them returns exactly
    what mi_roi_align_forward_fpn returns for the same blob (same records, same kernel).  Rows that are no proposals (score
    -inf) carry image index -1 and pool zeros."""
    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import PreparedRecords, roi_align_fpn

    d, lib = dev(), _lib.lib()
    rng = np.random.RandomState(17)
    n, c, k_min, k_max, rows, total = 2, 64, 2, 5, 300, 420
    sizes = {5: (13, 21), 4: (25, 42), 3: (50, 84), 2: (100, 168)}                       # a 400 x 672 image
    feats = [to_dev(syn.feature_map(n, c, *sizes[l], seed=l)) for l in (5, 4, 3, 2)]     # coarsest first
    if channels_last:
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    scales = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4]
    side = np.exp(rng.uniform(np.log(12.0), np.log(380.0), total))
    x1, y1 = rng.uniform(0, 672 - 8, total), rng.uniform(0, 400 - 8, total)
    cand = np.stack([rng.randint(0, n, total), x1, y1, np.minimum(x1 + side, 671), np.minimum(y1 + side * rng.uniform(0.5, 2, total), 399)], 1)
    cand_rois = to_dev(cand.astype(np.float32))
    scores = rng.rand(total).astype(np.float32)
    scores[rng.rand(total) < 0.45] = -np.inf            # fewer proposals than rows: padding rows at the end of the blob
    best, inds = torch.topk(to_dev(scores), rows)
    inds = inds.to(torch.int64).contiguous()
    stream = _lib.current_stream_handle(d)

    def blob():
        return (torch.empty((rows, 5), device=d), torch.empty((rows,), dtype=torch.bool, device=d),
                torch.empty((rows,), dtype=torch.int32, device=d))

    rois_a, valid_a, lv_a = blob()
    _lib.check(lib.mi_rpn_collect_finish(best.data_ptr(), inds.data_ptr(), cand_rois.data_ptr(), rows, 1, k_min, k_max, 224.0,
                                         4.0, rois_a.data_ptr(), valid_a.data_ptr(), lv_a.data_ptr(), stream), "finish")
    rec = PreparedRecords(feats, scales, 7, 7, 2, rows)
    assert rec.supported
    rois_b, valid_b, lv_b = blob()
    _lib.check(lib.mi_rpn_collect_finish_records(best.data_ptr(), inds.data_ptr(), cand_rois.data_ptr(), rows, 1, k_min, k_max,
                                                 224.0, 4.0, rois_b.data_ptr(), valid_b.data_ptr(), lv_b.data_ptr(),
                                                 ctypes.byref(rec.table), rec.batch, rec.channels, 7, 7, 2, rec.layout,
                                                 rec.workspace.data_ptr(), rec.workspace.numel(), stream), "finish_records")
    rec.rois = rois_b
    assert torch.equal(rois_a, rois_b) and torch.equal(valid_a, valid_b) and torch.equal(lv_a, lv_b)
    assert 0 < int(valid_a.sum()) < rows and bool((rois_a[~valid_a, 0] == -1).all())
    with torch.no_grad():
        want = roi_align_fpn(feats, scales, rois_a, (k_max - lv_a), 7, 7, 2)
        got = roi_align_fpn(feats, scales, rois_b, (k_max - lv_b), 7, 7, 2, prepared=rec)
        assert rec.matches(feats, scales, rois_b, 7, 7, 2) and not rec.matches(feats, scales, rois_a, 7, 7, 2)
    assert torch.equal(got, want)
    assert bool((got[~valid_a] == 0).all()) and float(got[valid_a].abs().sum()) > 0


def _clustered_rois(num, batch, height, width, scale, boxes_per_image, seed):
    """RoIs jittered around a few boxes per image (what a training step samples around its ground truth)."""
    rng = np.random.RandomState(seed)
    img_w, img_h = width / scale, height / scale
    gts = []
    for n in range(batch):
        for _ in range(boxes_per_image):
            w, h = rng.uniform(0.08, 0.45) * img_w, rng.uniform(0.08, 0.45) * img_h
            x, y = rng.uniform(0, img_w - w), rng.uniform(0, img_h - h)
            gts.append((n, x, y, x + w, y + h))
    pick = rng.randint(0, len(gts), size=num)
    rois = np.zeros((num, 5), np.float32)
    for i, k in enumerate(pick):
        n, x1, y1, x2, y2 = gts[k]
        j = rng.uniform(-0.12, 0.12, size=4) * np.array([x2 - x1, y2 - y1, x2 - x1, y2 - y1])
        rois[i] = (n, max(x1 + j[0], 0), max(y1 + j[1], 0), min(x2 + j[2], img_w - 1), min(y2 + j[3], img_h - 1))
    return rois


@pytest.mark.parametrize("slice_len", [None, 8, 2])   # 2: more slices than the item table holds -> the slice length doubles
@pytest.mark.parametrize("res,channels_last", [(7, False), (14, False), (7, True)])
def test_roi_align_backward_of_clustered_rois_is_cut_into_list_slices(oracle_mod, tuning_env, res, channels_last, slice_len):
    """The planned backward (roi_align_bwd_plan + roi_align_bwd_tiles over list slices): 600 RoIs on 3 boxes per image make
    tile lists of 100+ RoIs, which several workgroups sum with atomics into tiles the plan kernel zero-filled; tiles without
    RoIs get no workgroup.  Against the oracle, overwrite contract (autograd path) and accumulate contract (raw call on a
    pre-filled buffer), and next to the unplanned backward."""
    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    n, c, h, w, scale = 2, 64, 50, 84, 1.0 / 16
    feat = syn.feature_map(n, c, h, w, seed=5)
    rois = _clustered_rois(600, n, h, w, scale, 3, seed=res)
    gtop = np.random.RandomState(3).randn(600, c, res, res).astype(np.float32)
    ref = oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, 2, threads=8)
    tuning_env(MI_ROI_ALIGN_BWD_SLICE=slice_len)
    assert _backward_workspace_bytes([(h, w)], n, 600) > _lib.lib().mi_roi_align_forward_workspace_bytes(600)
    _, grad = _roi_align_gpu(feat, rois, res, scale, 2, gtop, channels_last=channels_last)
    assert_close(grad, ref, "planned bwd")
    # accumulate contract through the C-ABI: flags = 0, the buffer keeps what it held
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    buf = torch.full((n, c, h, w), 0.5, device=dev()).contiguous(memory_format=fmt)
    ws = torch.empty(_backward_workspace_bytes([(h, w)], n, 600), dtype=torch.uint8, device=dev())
    g, r = to_dev(gtop), to_dev(rois)
    rc = _lib.lib().mi_roi_align_backward_ws(g.data_ptr(), r.data_ptr(), buf.data_ptr(), n, c, h, w, 600, res, res, scale, 2,
                                            _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NHWC if channels_last else _lib.LAYOUT_NCHW,
                                            ws.data_ptr(), ws.numel(), 0, _lib.current_stream_handle(dev()))
    _lib.check(rc, "mi_roi_align_backward_ws")
    assert_close(buf, ref + 0.5, "planned bwd, accumulate")
    # torch.use_deterministic_algorithms(True): no list slices, no atomics -- two runs agree to the last bit
    torch.use_deterministic_algorithms(True)
    try:
        _, det_a = _roi_align_gpu(feat, rois, res, scale, 2, gtop, channels_last=channels_last)
        _, det_b = _roi_align_gpu(feat, rois, res, scale, 2, gtop, channels_last=channels_last)
    finally:
        torch.use_deterministic_algorithms(False)
    assert torch.equal(det_a, det_b)
    assert_close(det_a, ref, "deterministic bwd")
    tuning_env(MI_ROI_ALIGN_BWD_SLICE=0)
    assert _backward_workspace_bytes([(h, w)], n, 600) == _lib.lib().mi_roi_align_forward_workspace_bytes(600)
    _, grad0 = _roi_align_gpu(feat, rois, res, scale, 2, gtop, channels_last=channels_last)
    assert_close(grad0, ref, "unplanned bwd")


def test_roi_align_backward_with_more_tiles_than_the_plan_holds(oracle_mod):
    """2200 small images give 8800 gradient tiles, more than the item-table kernel stages (8192): the workspace query then
    asks for the records only and the backward runs unplanned -- same gradients."""
    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    n, c, h, w, scale, res, r = 2200, 32, 17, 33, 1.0 / 16, 7, 96
    feat = syn.feature_map(n, c, h, w, seed=2)
    rois = syn.rois_adversarial(r, n, h, w, scale, seed=4)
    gtop = np.random.RandomState(6).randn(r, c, res, res).astype(np.float32)
    assert _backward_workspace_bytes([(h, w)], n, r) == _lib.lib().mi_roi_align_forward_workspace_bytes(r)
    _, grad = _roi_align_gpu(feat, rois, res, scale, 2, gtop)
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, 2, threads=8), "bwd, 8800 tiles")


@pytest.mark.parametrize("room_for_backward", [True, False])
def test_roi_align_records_ready_with_and_without_room_for_the_backward(oracle_mod, room_for_backward):
    """C-ABI contract of MI_ROI_ALIGN_RECORDS_READY: a forward given a workspace of mi_roi_align_backward_workspace_bytes
    writes the records' backward block and the backward reuses the records; with a workspace of the forward's size the
    forward skips the block (inference pays nothing for it) and a backward that passes RECORDS_READY anyway rewrites the
    records itself -- same gradients either way."""
    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    lib = _lib.lib()
    n, c, h, w, scale, res, r = 2, 64, 50, 84, 1.0 / 16, 7, 200
    feat = syn.feature_map(n, c, h, w, seed=8)
    rois = syn.rois_adversarial(r, n, h, w, scale, seed=21)
    gtop = np.random.RandomState(5).randn(r, c, res, res).astype(np.float32)
    fwd_bytes = int(lib.mi_roi_align_forward_workspace_bytes(r))
    ws_bytes = _backward_workspace_bytes([(h, w)], n, r) if room_for_backward else fwd_bytes
    assert (ws_bytes > fwd_bytes) == room_for_backward
    f, ro, g = to_dev(feat), to_dev(rois), to_dev(gtop)
    out = torch.empty((r, c, res, res), device=dev())
    grad = torch.empty((n, c, h, w), device=dev())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev())
    stream = _lib.current_stream_handle(dev())
    _lib.check(lib.mi_roi_align_forward_ws(f.data_ptr(), ro.data_ptr(), out.data_ptr(), n, c, h, w, r, res, res, scale, 2,
                                          _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW, ws.data_ptr(), ws_bytes, stream), "fwd")
    _lib.check(lib.mi_roi_align_backward_ws(g.data_ptr(), ro.data_ptr(), grad.data_ptr(), n, c, h, w, r, res, res, scale, 2,
                                           _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW, ws.data_ptr(), ws_bytes,
                                           _lib.ROI_ALIGN_RECORDS_READY | _lib.ROI_ALIGN_OVERWRITE, stream), "bwd")
    assert_fwd(out, oracle_mod.roi_align_forward(feat, rois, res, res, scale, 2, threads=8), "fwd", exact=False)
    assert_close(grad, oracle_mod.roi_align_backward(gtop, rois, feat.shape, scale, 2, threads=8), "bwd, records ready")


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("res,sr,channels", [(14, 2, 32), (7, 0, 64), (7, 3, 32), (7, 2, 256)])
def test_roi_align_fpn_fused_other_resolutions(oracle_mod, res, sr, channels, channels_last):
    """Mask-head resolution (16-channel backward tiles), adaptive and odd sampling grids through the fused entry points."""
    from detectron_pytorch_amd.roi_align import roi_align_fpn

    rois, lvls, _, scales, feats = _fpn_inputs(channels=channels, num_rois=160, batch=2, seed=res + sr)
    keep = lvls >= 3                                          # three levels only: P5, P4, P3
    rois, lvls = rois[keep], lvls[keep]
    feats, scales = feats[:3], scales[:3]
    idx_of = (5 - lvls).astype(np.int32)
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    dev_feats = [to_dev(f).contiguous(memory_format=fmt).requires_grad_(True) for f in feats]
    out = roi_align_fpn(dev_feats, scales, to_dev(rois), to_dev(idx_of), res, res, sr)
    gtop = np.random.RandomState(1).randn(len(rois), channels, res, res).astype(np.float32)
    out.backward(to_dev(gtop))
    assert all(f.grad.is_contiguous(memory_format=fmt) for f in dev_feats)
    for k in range(3):
        idx = np.nonzero(idx_of == k)[0]
        assert_fwd(out.detach()[torch.from_numpy(idx).to(dev())],
                   oracle_mod.roi_align_forward(feats[k], rois[idx], res, res, scales[k], sr, threads=8), "fwd", exact=False)
        assert_close(dev_feats[k].grad, oracle_mod.roi_align_backward(gtop[idx], rois[idx], feats[k].shape, scales[k], sr,
                                                                      threads=8), "bwd")


# ---- the nn.Module wrappers of the reference (roi_xfrom/roi_align/modules/roi_align.py:6-45, model/roi_align/modules/
# roi_align.py:6-42, model/roi_pooling/modules/roi_pool.py:5-14): same constructor arguments, forward + backward --------
@pytest.mark.parametrize("cls_name", ["RoIAlign", "RoIAlignAvg", "RoIAlignMax", "LegacyRoIAlign", "LegacyRoIAlignAvg",
                                      "LegacyRoIAlignMax", "_RoIPooling"])
def test_reference_modules_forward_backward(oracle_mod, cls_name):
    import torch.nn.functional as F

    from detectron_pytorch_amd import roi_align as ra, roi_pool as rp

    feat = syn.feature_map(2, 12, 25, 42, seed=3)
    scale, res = 1.0 / 16, 7
    rois = syn.rois_adversarial(40, 2, 25, 42, scale, seed=5)
    plus = 1 if cls_name.endswith(("Avg", "Max")) else 0
    if cls_name == "_RoIPooling":
        mod = rp._RoIPooling(res, res, scale)
        base, argmax = oracle_mod.roi_pool_forward(feat, rois, res, res, scale)
    elif cls_name.startswith("Legacy"):
        mod = getattr(ra, cls_name)(res, res, scale)
        base = oracle_mod.roi_align_legacy_forward(feat, rois, res + plus, res + plus, scale)
    else:
        mod = getattr(ra, cls_name)(res, res, scale, 2)
        base = oracle_mod.roi_align_forward(feat, rois, res + plus, res + plus, scale, 2)
    # the pooling the reference's module applies on top (avg_pool2d / max_pool2d, kernel 2, stride 1), on the CPU
    base_t = torch.from_numpy(base).requires_grad_(True)
    want = F.avg_pool2d(base_t, 2, 1) if cls_name.endswith("Avg") else F.max_pool2d(base_t, 2, 1) if plus else base_t
    f = to_dev(feat).requires_grad_(True)
    out = mod(f, to_dev(rois))
    assert out.shape == (40, 12, res, res)
    assert_fwd(out, want.detach().numpy(), cls_name + " fwd", exact=False)
    gtop = np.random.RandomState(1).randn(*out.shape).astype(np.float32)
    out.backward(to_dev(gtop))
    want.backward(torch.from_numpy(gtop))
    g_base = base_t.grad.numpy()
    if cls_name == "_RoIPooling":
        ref_grad = oracle_mod.roi_pool_backward(g_base, rois, argmax, feat.shape, scale)
    elif cls_name.startswith("Legacy"):
        ref_grad = oracle_mod.roi_align_legacy_backward(g_base, rois, feat.shape, scale)
    else:
        ref_grad = oracle_mod.roi_align_backward(g_base, rois, feat.shape, scale, 2)
    assert_close(f.grad, ref_grad, cls_name + " bwd")


# ---- sorted top-k (the selections around the NMS kernels) ------------------------------------------------------------
def _topk_reference(v, k):
    """Descending values, ties by ascending index, NaN last: np.lexsort restatement of mi_topk_batched's contract."""
    key = np.where(np.isnan(v), -np.inf, v).astype(np.float64)
    key[np.isnan(v)] = -np.inf
    nan_rank = np.isnan(v).astype(np.int64)                       # NaN below -inf
    order = np.lexsort((np.arange(len(v)), -key, nan_rank))
    return order[:k]


@pytest.mark.parametrize("n,k,kind", [(201600, 2000, "sigmoid"), (201600, 1000, "concentrated"), (50400, 4096, "normal"),
                                      (819, 819, "normal"), (1, 1, "normal"), (80000, 128, "masked"),
                                      (5000, 1000, "ties"), (3000, 700, "nan"), (70000, 1000, "few_finite")])
def test_topk_batched_matches_numpy(n, k, kind):
    from detectron_pytorch_amd import topk

    rng = np.random.RandomState(n % 97 + k)
    if kind == "sigmoid":
        v = (1 / (1 + np.exp(-rng.randn(n) * 3))).astype(np.float32)
    elif kind == "concentrated":                                   # a fresh RPN: every score within 1e-3 of 0.5
        v = (0.5 + rng.randn(n) * 1e-3).astype(np.float32)
    elif kind == "masked":                                         # the detection cut: mostly -inf
        v = np.full(n, -np.inf, np.float32)
        live = rng.choice(n, 3000, replace=False)
        v[live] = rng.rand(3000).astype(np.float32)
    elif kind == "ties":                                           # 16 distinct values: the k-th is tied many times over
        v = (rng.randint(0, 16, n) / 16).astype(np.float32)
    elif kind == "nan":
        v = rng.randn(n).astype(np.float32)
        v[rng.rand(n) < 0.3] = np.nan
        v[rng.rand(n) < 0.1] = -np.inf
        v[rng.rand(n) < 0.05] = np.inf
    elif kind == "few_finite":                                     # fewer live candidates than k: -inf rows fill up
        v = np.full(n, -np.inf, np.float32)
        v[rng.choice(n, 300, replace=False)] = rng.randn(300).astype(np.float32)
    else:
        v = rng.randn(n).astype(np.float32)
    vals, idx = topk.topk(to_dev(v), k)
    want = _topk_reference(v, k)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(vals.cpu().numpy(), v[want], equal_nan=True)


def test_topk_batched_many_problems_and_limits():
    from detectron_pytorch_amd import _lib, topk

    rng = np.random.RandomState(0)
    rows = [rng.randn(n).astype(np.float32) for n in [10, 4000, 333, 64, 65, 100000] * 4]   # 24 problems: two launches
    ks = [min(len(r), kk) for r, kk in zip(rows, [10, 1000, 1, 64, 33, 2000] * 4)]
    out = topk.topk_many([to_dev(r) for r in rows], ks)
    for r, k, (vals, idx) in zip(rows, ks, out):
        want = _topk_reference(r, k)
        assert np.array_equal(idx.cpu().numpy(), want) and np.array_equal(vals.cpu().numpy(), r[want])
    x = to_dev(rng.randn(3, 5000).astype(np.float32))
    vals, idx = topk.topk_rows(x, 50)
    tv, ti = torch.topk(x, 50, dim=1)
    assert torch.equal(vals, tv) and torch.equal(idx, ti)           # untied: identical to torch.topk
    with pytest.raises(ValueError):
        topk.topk(x[0].contiguous(), 4097)
    lib = _lib.lib()
    arr = (ctypes.c_void_p * 1)(x.data_ptr())
    one = (ctypes.c_int * 1)(5000)
    big = (ctypes.c_int * 1)(4097)
    assert lib.mi_topk_batched(1, arr, one, big, arr, arr, None, 0, None) != 0 and b"4096" in lib.mi_last_error()
    wide = (ctypes.c_int * 1)(100000)
    ten = (ctypes.c_int * 1)(10)
    assert lib.mi_topk_batched_workspace_bytes(1, wide, ten) > 16 and lib.mi_topk_batched_workspace_bytes(1, one, ten) == 16
    assert lib.mi_topk_batched(1, arr, wide, ten, arr, arr, None, 0, None) != 0 and b"workspace" in lib.mi_last_error()


def test_rpn_collect_static_fused_equals_the_tensor_expression_path():
    """generate_and_collect through the fused kernels (mi_topk_batched / mi_rpn_collect_candidates / _finish) == the same
    pipeline written with torch.topk and tensor expressions, including the static form with padding rows and levels."""
    from detectron_pytorch_amd import fpn_proposals, generate_proposals as gp

    levels = [(2, 200, 336, 4, 32), (3, 100, 168, 8, 64), (4, 50, 84, 16, 128), (5, 25, 42, 32, 256), (6, 13, 21, 64, 512)]
    im_info = to_dev(np.array([[800, 1344, 1.0], [760, 1200, 1.5]], np.float32))
    for pre, post, thresh, min_size in ((1000, 1000, 0.7, 0), (2000, 2000, 0.7, 0), (600, 300, 0.0, 16)):
        ops, heads = [], []
        for lvl, h, w, stride, size in levels:
            anchors = gp.generate_anchors(stride, (size,), (0.5, 1, 2))
            sc, dl = syn.rpn_head_outputs(2, 3, h, w, seed=lvl)
            ops.append(gp.GenerateProposalsOp(anchors, 1.0 / stride, pre, post, thresh, min_size, as_numpy=False))
            heads.append((to_dev(sc), to_dev(dl)))
        assert fpn_proposals._fused_supported(ops, heads, post)
        got = fpn_proposals.generate_and_collect(ops, heads, im_info, post, static=True, with_levels=True)
        want = fpn_proposals._generate_and_collect_torch(ops, heads, im_info, post, True, True, 2, 5)
        n_valid = int(want[1].sum())
        assert torch.equal(got[1], want[1]) and n_valid > 0
        # rows that are proposals: same RoIs in the same order (scores are untied), same levels; padding rows: image -1
        assert torch.equal(got[0][:n_valid], want[0][:n_valid]) and torch.equal(got[2][:n_valid], want[2][:n_valid])
        assert bool((got[0][n_valid:, 0] == -1).all())
        dyn = fpn_proposals.generate_and_collect(ops, heads, im_info, post)
        assert torch.equal(dyn, got[0][:n_valid])
    # an over-asked collect (post > candidates that survive): the tail is padding
    got = fpn_proposals.generate_and_collect(ops, heads, im_info, 4000, static=True)
    assert got[0].shape[0] == 4000 and int(got[1].sum()) < 4000


# ---- RPN proposal generation on the device (generate_proposals.py:12-182) -----------------------------------------
@pytest.mark.parametrize("name", ["p4", "p5", "p3min"])
def test_generate_proposals_golden(name):
    """Fixture = the reference's GenerateProposalsOp executed from its own source: same RoIs, same order, same bits."""
    from detectron_pytorch_amd import generate_proposals as gp

    g = load_golden("proposals.npz")
    stride, size, h, w, pre, post, min_size = (int(v) for v in g["cfg_" + name])
    anchors = gp.generate_anchors(stride, (size,), (0.5, 1, 2))
    scores, deltas = syn.rpn_head_outputs(2, anchors.shape[0], h, w, seed=stride)
    op = gp.GenerateProposalsOp(anchors, 1.0 / stride, pre, post, 0.7, min_size)
    rois, probs = op(to_dev(scores), to_dev(deltas), g["im_info_" + name])
    assert rois.dtype == np.float32 and probs.shape == (rois.shape[0], 1)
    assert np.array_equal(probs, g["probs_" + name]), "kept set / order differs"
    assert np.array_equal(rois, g["rois_" + name]), "max |diff| %g" % np.abs(rois - g["rois_" + name]).max()


def test_generate_proposals_vs_oracle_p2_sized_level(oracle_mod):
    """A P2-sized level (201 600 anchors), train-time top-k (2000 -> 2000), two images, tensors out."""
    from detectron_pytorch_amd import generate_proposals as gp
    from oracle import proposals

    anchors = gp.generate_anchors(4, (32,), (0.5, 1, 2))
    scores, deltas = syn.rpn_head_outputs(2, 3, 200, 336, seed=4)
    im_info = np.array([[800, 1344, 1.0], [768, 1216, 1.3]], np.float32)
    want_rois, want_probs = proposals.generate_proposals(scores, deltas, im_info, anchors, 0.25, 2000, 2000, 0.7, 0)
    op = gp.GenerateProposalsOp(anchors, 0.25, 2000, 2000, 0.7, 0, as_numpy=False)
    rois, probs = op(to_dev(scores), to_dev(deltas), to_dev(im_info))
    assert rois.is_cuda and probs.is_cuda
    assert np.array_equal(probs.cpu().numpy(), want_probs) and np.array_equal(rois.cpu().numpy(), want_rois)
    # no NMS: thresholded top-k only
    op = gp.GenerateProposalsOp(anchors, 0.25, 500, 0, 0.0, 8)
    r2, p2 = op(to_dev(scores), to_dev(deltas), im_info)
    w2 = proposals.generate_proposals(scores, deltas, im_info, anchors, 0.25, 500, 0, 0.0, 8)
    assert np.array_equal(r2, w2[0]) and np.array_equal(p2, w2[1])


def test_generate_proposals_more_than_4096_pre_nms_boxes(oracle_mod):
    """The non-FPN (C4) settings keep 6000 / 12000 boxes before NMS: beyond the batched NMS entry (4096), through mi_nms."""
    from detectron_pytorch_amd import generate_proposals as gp
    from oracle import proposals

    anchors = gp.generate_anchors(16, (32, 64, 128, 256, 512), (0.5, 1, 2))
    scores, deltas = syn.rpn_head_outputs(1, 15, 38, 50, seed=6)
    im_info = np.array([[600, 800, 1.0]], np.float32)
    want = proposals.generate_proposals(scores, deltas, im_info, anchors, 1.0 / 16, 6000, 1000, 0.7, 0)
    got = gp.GenerateProposalsOp(anchors, 1.0 / 16, 6000, 1000, 0.7, 0)(to_dev(scores), to_dev(deltas), im_info)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])


def test_collect_and_distribute_inference_path(oracle_mod):
    """collect_and_distribute_fpn_rpn_proposals.py:83-119 on the device, then the fused RoIAlign straight from its output."""
    from detectron_pytorch_amd import fpn_proposals, roi_xform
    from detectron_pytorch_amd.roi_align import roi_align_fpn

    # fixture from the reference's utils/fpn.py: level of every RoI, per-level blobs, restore permutation
    g = load_golden("fpn.npz")
    rois = g["rois"]
    blobs = fpn_proposals.distribute(to_dev(rois))
    assert np.array_equal(blobs["roi_levels"].cpu().numpy(), g["levels"].astype(np.int32))
    for key in ("rois_fpn2", "rois_fpn3", "rois_fpn4", "rois_fpn5", "rois_idx_restore_int32"):
        assert np.array_equal(blobs[key].cpu().numpy(), g[key]), key
    # collect: five levels of proposals with distinct scores -> the N best in descending order
    rng = np.random.RandomState(0)
    all_scores = rng.permutation(5 * 700).astype(np.float32) / 3500
    parts = [(syn.rois_fpn_distributed(700, batch=2, seed=l)[0], all_scores[700 * l:700 * (l + 1), None]) for l in range(5)]
    want = np.concatenate([p[0] for p in parts])[np.argsort(-all_scores)[:1000]]
    got = fpn_proposals.collect([to_dev(p[0]) for p in parts], [to_dev(p[1]) for p in parts], 1000)
    assert np.array_equal(got.cpu().numpy(), want)
    lv = fpn_proposals.map_rois_to_fpn_levels(got[:, 1:5])
    assert np.array_equal(lv.cpu().numpy(), roi_xform.map_rois_to_fpn_levels(want[:, 1:5]).astype(np.int32))
    # ... and pooled in one call from exactly these tensors
    scales = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4]
    feats = [syn.feature_map(2, 32, int(np.ceil(syn.IM_H * s)), int(np.ceil(syn.IM_W * s)), seed=9 + i)
             for i, s in enumerate(scales)]
    out = roi_align_fpn([to_dev(f) for f in feats], scales, got, 5 - lv, 7, 7, 2).cpu().numpy()
    lv_np = lv.cpu().numpy()
    for lvl in range(2, 6):
        idx = np.nonzero(lv_np == lvl)[0]
        assert_fwd(out[idx], oracle_mod.roi_align_forward(feats[5 - lvl], want[idx], 7, 7, scales[5 - lvl], 2, threads=8),
                   "lvl %d" % lvl, exact=False)


def test_generate_and_collect_matches_per_level_ops_then_collect(oracle_mod):
    """The single-pipeline form (one batched NMS for all levels and images, global top-k over masked scores) gives what
    the reference's sequence gives: GenerateProposalsOp per level, then collect (collect_and_distribute...py:83-98)."""
    from detectron_pytorch_amd import fpn_proposals, generate_proposals as gp
    from oracle import proposals

    levels = [(3, 100, 168, 8, 64), (4, 50, 84, 16, 128), (5, 25, 42, 32, 256)]
    im_info = np.array([[800, 1344, 1.0], [760, 1200, 1.5]], np.float32)
    ops, heads, want_rois, want_probs = [], [], [], []
    sizes = [2 * 3 * h * w for _, h, w, _, _ in levels]
    ranks = np.random.RandomState(3).permutation(sum(sizes))                 # scores distinct across levels AND images,
    first = 0                                                                # exactly representable: (rank + 1) / 2^22
    for (lvl, h, w, stride, size), cnt in zip(levels, sizes):
        anchors = gp.generate_anchors(stride, (size,), (0.5, 1, 2))
        _, dl = syn.rpn_head_outputs(2, 3, h, w, seed=20 + lvl)
        sc = ((ranks[first:first + cnt] + 1).astype(np.float32) / np.float32(1 << 22)).reshape(2, 3, h, w)
        first += cnt
        ops.append(gp.GenerateProposalsOp(anchors, 1.0 / stride, 1000, 300, 0.7, 0, as_numpy=False))
        heads.append((to_dev(sc), to_dev(dl)))
        r, p = proposals.generate_proposals(sc, dl, im_info, anchors, 1.0 / stride, 1000, 300, 0.7, 0)
        want_rois.append(r)
        want_probs.append(p)
    all_rois, all_probs = np.concatenate(want_rois), np.concatenate(want_probs).squeeze()
    assert len(np.unique(all_probs)) == len(all_probs)
    want = all_rois[np.argsort(-all_probs)[:500]]
    got = fpn_proposals.generate_and_collect(ops, heads, to_dev(im_info), 500)
    assert np.array_equal(got.cpu().numpy(), want)
    few = fpn_proposals.generate_and_collect(ops, heads, to_dev(im_info), 100000)   # fewer proposals than asked for
    assert np.array_equal(few.cpu().numpy(), all_rois[np.argsort(-all_probs)])


# ---- mask targets from polygon ground truth (roi_data/mask_rcnn.py:34-107, utils/segms.py) ---------------------------
def _packed(points, poly_start, inst_start, device):
    from detectron_pytorch_amd.segms import PackedPolygons

    return PackedPolygons(torch.from_numpy(np.ascontiguousarray(points)).to(device),
                          torch.from_numpy(np.ascontiguousarray(poly_start)).to(device),
                          torch.from_numpy(np.ascontiguousarray(inst_start)).to(device))


def test_polygon_mask_targets_equal_the_references_blobs():
    """Golden: the reference's own add_mask_rcnn_blobs + utils/segms.py executed from source on COCO-like polygons
    (tests/golden/generate.py gen_mask_targets; pycocotools' two calls bound to the restatement of its maskApi.c).  The
    device path -- boxes enclosing the polygons, IoU matrix, argmax, one rasterising launch for all foreground rows --
    reproduces the enclosing boxes, the mask RoIs and every mask bit for bit."""
    from detectron_pytorch_amd import nms, segms

    g = load_golden("mask_targets.npz")
    for tag in ("a", "b"):
        m = int(g[tag + "_resolution"])
        packed = _packed(g[tag + "_points"], g[tag + "_poly_start"], g[tag + "_inst_start"], dev())
        boxes_from_polys = segms.polys_to_boxes(packed)
        assert np.array_equal(boxes_from_polys.cpu().numpy(), g[tag + "_boxes_from_polys"])
        fg = g[tag + "_labels"] > 0
        rois_fg = to_dev(g[tag + "_sampled_boxes"][fg])
        inst = nms.bbox_overlaps(rois_fg, boxes_from_polys).argmax(dim=1)
        masks = segms.polys_to_masks_wrt_boxes(packed, inst, rois_fg, m)
        assert masks.dtype == torch.int32 and tuple(masks.shape) == (int(fg.sum()), m * m)
        assert np.array_equal(masks.cpu().numpy().astype(np.int8), g[tag + "_masks_int32"]), tag
        mask_rois = torch.cat([torch.full((rois_fg.size(0), 1), 1.0, device=dev()), rois_fg * 1.5], dim=1)
        assert np.array_equal(mask_rois.cpu().numpy(), g[tag + "_mask_rois"])


@pytest.mark.parametrize("m", [7, 14, 28, 56, 64])
def test_polygon_rasteriser_against_the_oracle(oracle_mod, m):
    """mi_polys_to_masks_wrt_boxes (a parallel form of pycocotools' rleFrPoly: crossings toggle column-major positions, a
    parity scan decodes) against the oracle's sequential restatement (sort, run-length merge, decode), bit for bit:
    instances of 1-3 polygons (concave, self-intersecting, repeated vertices, parts outside the image), RoIs jittered
    around them, a point, slivers, far away, larger than the image; rows of no instance stay zero."""
    import cpu_backend
    from detectron_pytorch_amd import segms
    from detectron_pytorch_amd.segms import PackedPolygons

    polys, boxes, _ = syn.polygon_instances(12, seed=100 + m)
    # an outline of 2500 vertices (the kernel takes a polygon's edges 1024 at a time) in place of instance 11
    ang = np.linspace(0, 2 * np.pi, 2500, endpoint=False)
    rad = 90 + 25 * np.sin(9 * ang) + np.random.RandomState(m).uniform(-2, 2, 2500)
    dense = np.round(np.stack([600 + rad * np.cos(ang), 400 + 0.7 * rad * np.sin(ang)], 1), 2)
    polys[11] = [[float(v) for v in dense.reshape(-1)]]
    boxes[11] = [dense[:, 0].min(), dense[:, 1].min(), dense[:, 0].max(), dense[:, 1].max()]
    rois = syn.jittered_boxes(boxes, 8, seed=m, jitter=0.4)
    inst = np.repeat(np.arange(12), 8)
    special = np.array([[boxes[0, 0], boxes[0, 1], boxes[0, 0], boxes[0, 1]], [boxes[1, 0] + 3, boxes[1, 1], boxes[1, 0] + 3.3, boxes[1, 3]],
                        [boxes[2, 0], boxes[2, 1] + 5, boxes[2, 2], boxes[2, 1] + 5.2], [5, 5, 20, 20], [-200, -200, 1600, 1000],
                        [boxes[5, 0] + 0.37, boxes[5, 1] + 0.61, boxes[5, 2] - 0.13, boxes[5, 3] - 0.29], boxes[6], boxes[7]], np.float32)
    rois = np.vstack([rois, special]).astype(np.float32)
    inst = np.concatenate([inst, [0, 1, 2, 3, 4, 5, 6, -1]]).astype(np.int64)
    packed_cpu = PackedPolygons.from_lists(polys)
    want = cpu_backend.polys_to_masks_wrt_boxes(packed_cpu, torch.from_numpy(inst), torch.from_numpy(rois), m).numpy()
    got = segms.polys_to_masks_wrt_boxes(packed_cpu.to(dev()), to_dev(inst), to_dev(rois), m).cpu().numpy()
    assert np.array_equal(got, want), "rows that differ: %s" % np.nonzero((got != want).any(1))[0]
    assert 0.05 < want[:96].mean() < 0.9 and not got[-1].any()
    # rectangles in the polygon format (SURVEY section 8d config 4's ground truth): whole pixels of an aligned rectangle
    rect = PackedPolygons.from_boxes(torch.from_numpy(boxes))
    want = cpu_backend.polys_to_masks_wrt_boxes(rect, torch.from_numpy(inst[:96]), torch.from_numpy(rois[:96]), m).numpy()
    got = segms.polys_to_masks_wrt_boxes(rect.to(dev()), to_dev(inst[:96]), to_dev(rois[:96]), m).cpu().numpy()
    assert np.array_equal(got, want)
    whole = segms.polys_to_masks_wrt_boxes(rect.to(dev()), to_dev(np.arange(12)), to_dev(boxes), m).cpu().numpy()
    assert whole.all()                                                        # a RoI that is its instance's rectangle


def test_polygon_rasteriser_contract():
    from detectron_pytorch_amd import _lib, segms
    from detectron_pytorch_amd.segms import PackedPolygons

    polys, boxes, _ = syn.polygon_instances(3, seed=5)
    packed = PackedPolygons.from_lists(polys, device=dev())
    with pytest.raises(_lib.MiOpsError, match="resolution"):
        segms.polys_to_masks_wrt_boxes(packed, to_dev(np.zeros(3, np.int64)), to_dev(boxes), 65)
    assert tuple(segms.polys_to_masks_wrt_boxes(packed, to_dev(np.zeros(0, np.int64)), to_dev(np.zeros((0, 4), np.float32)), 28).shape) == (0, 784)
    with pytest.raises(NotImplementedError):
        segms.polys_to_masks_wrt_boxes(packed, torch.zeros(3, dtype=torch.int64), torch.from_numpy(boxes), 28)


# ---- the records-free forward (roi_align_fwd_slab): one launch, no workspace contents ----------------------------------
def _forward_ws_raw(feat_t, rois_t, res, scale, sr, ws):
    from detectron_pytorch_amd import _lib

    n, c, h, w = feat_t.shape
    r = rois_t.size(0)
    out = torch.full((r, c, res, res), float("nan"), device=dev())
    lib = _lib.lib()
    rc = lib.mi_roi_align_forward_ws(feat_t.data_ptr(), rois_t.data_ptr(), out.data_ptr(), n, c, h, w, r, res, res, scale, sr,
                                     0, 0, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                     _lib.current_stream_handle(dev()))
    assert rc == 0, lib.mi_last_error()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("shape,res,sr,nrois", [((1, 256, 200, 336), 7, 2, 300),   # config-2 geometry: 1 to 4 stages per RoI
                                                ((2, 64, 100, 168), 14, 2, 96),    # mask resolution, two images
                                                ((2, 8, 25, 42), 7, 2, 64),        # one channel tile: 7 of 8 XCD columns idle
                                                ((1, 72, 50, 84), 7, 2, 80),       # 9 tiles: a second, partial phase
                                                ((3, 40, 13, 21), 7, 0, 50),       # adaptive sampling grid, tiny map
                                                ((1, 24, 30, 40), 6, 3, 40),       # generic instance, sampling ratio 3
                                                ((1, 16, 9, 70), 3, 1, 33),
                                                ((1, 32, 50, 84), 16, 2, 40),      # 32 samples per axis: the tables' limit
                                                ((1, 16, 50, 84), 21, 2, 24),      # beyond it: every RoI on the in-kernel direct path
                                                ((2, 32, 25, 42), 10, 0, 48)])
def test_roi_align_records_free_forward_equals_the_record_driven_one(oracle_mod, tuning_env, shape, res, sr, nrois):
    """A forward whose workspace has no room for a backward runs roi_align_fwd_slab: one launch that computes every RoI's
    geometry, tables and stages itself.  Same tables, same bins as the record-driven kernel -> the same bits wherever
    that kernel serves the shape (32-channel tiles), the oracle within the fast paths' bar everywhere; adversarial RoIs
    (outside the image, no image, degenerate, map-sized) take its in-kernel reference-order path."""
    from detectron_pytorch_amd import _lib

    n, c, h, w = shape
    scale = 1.0 / 16 if h < 100 else 0.25
    feat = syn.feature_map(n, c, h, w, seed=res + sr + c)
    third = nrois // 3
    rois = np.vstack([syn.rois_adversarial(third, n, h, w, scale, seed=nrois),
                      syn.rois_canonical(third, n, seed=nrois + 1, side=(2.0 / scale, min(h, w) / scale * 0.9),
                                         im_h=h / scale, im_w=w / scale),
                      syn.rois_canonical(nrois - 2 * third, n, seed=nrois + 2, side=(1.5 / scale, 22.0 / scale),
                                         im_h=h / scale, im_w=w / scale)])
    f, rt = to_dev(feat), to_dev(rois)
    lib = _lib.lib()
    fws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(nrois), dtype=torch.uint8, device=dev())
    tuning_env(MI_ROI_ALIGN_SLAB="1")
    got = _forward_ws_raw(f, rt, res, scale, sr, fws)
    got_no_ws = _forward_ws_raw(f, rt, res, scale, sr, None)
    assert torch.equal(got, got_no_ws), "the entry point without a workspace runs the same kernel"
    ref = oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr, threads=8)
    assert_fwd(got, ref, "records-free forward", exact=False)
    tuning_env(MI_ROI_ALIGN_SLAB="0")
    base = _forward_ws_raw(f, rt, res, scale, sr, fws)
    assert_fwd(base, ref, "record-driven / fallback", exact=False)
    if c % 32 == 0:
        # the two kernels cut their stages for different LDS capacities: a RoI whose single bin row overflows the smaller
        # image takes the reference-order path there and the separable one here (last bits); everything smaller: same bits
        side = np.maximum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) * scale
        small = torch.from_numpy(np.nonzero(side <= 24.0)[0]).to(dev())
        assert small.numel() >= nrois // 4
        assert torch.equal(got[small], base[small]), "records-free and record-driven forwards differ"



@pytest.mark.gpu
@pytest.mark.parametrize("res", [7, 14])
def test_roi_align_records_free_forward_over_a_pyramid(oracle_mod, tuning_env, res):
    """mi_roi_align_forward_fpn with a forward-sized workspace (an inference call): the records-free kernel with the level
    index of every RoI; equal to the record-driven pyramid call bit for bit and to the oracle level by level."""
    import ctypes

    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _fpn_table

    frois, flvls = syn.rois_fpn_distributed(240, batch=2, seed=res)
    maps = [syn.feature_map(2, 64, syn.FPN_LEVELS[l][0], syn.FPN_LEVELS[l][1], seed=l) for l in (5, 4, 3, 2)]
    scales = [syn.FPN_LEVELS[l][2] for l in (5, 4, 3, 2)]
    mt = [to_dev(m) for m in maps]
    rt, idx = to_dev(frois), to_dev((5 - flvls).astype(np.int32))
    lib = _lib.lib()
    ftab = _fpn_table(mt, scales)
    ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(240), dtype=torch.uint8, device=dev())
    outs = {}
    for slab in ("1", "0"):
        tuning_env(MI_ROI_ALIGN_SLAB=slab)
        o = torch.full((240, 64, res, res), float("nan"), device=dev())
        rc = lib.mi_roi_align_forward_fpn(ctypes.byref(ftab), rt.data_ptr(), idx.data_ptr(), o.data_ptr(), 2, 64, 240, res, res, 2, 0,
                                          ws.data_ptr(), ws.numel(), _lib.current_stream_handle(dev()))
        assert rc == 0, lib.mi_last_error()
        outs[slab] = o
    assert torch.equal(outs["1"], outs["0"])
    got = outs["1"].cpu().numpy()
    for k, lvl in enumerate((5, 4, 3, 2)):
        sel = np.nonzero(flvls == lvl)[0]
        assert_fwd(got[sel], oracle_mod.roi_align_forward(maps[k], frois[sel], res, res, scales[k], 2, threads=8),
                   "level %d" % lvl, exact=False)


@pytest.mark.gpu
def test_roi_align_records_free_forward_partial_wait(tuning_env):
    """Multi-stage items of the records-free kernel (its bins store straight to memory, so every stage waits with vmcnt(0);
    the record-driven kernel behind the same switch waits with vmcnt(#stores)): MI_ROI_ALIGN_FWD_FULL_WAIT=1 against the
    default bit for bit, finite everywhere, release build."""
    h, w, scale = 200, 336, 0.25
    rois = to_dev(syn.rois_canonical(256, 1, seed=5, side=(64.0, 700.0)))
    for c, res in ((64, 7), (32, 14)):
        f = to_dev(syn.feature_map(1, c, h, w, seed=c))
        tuning_env(MI_ROI_ALIGN_SLAB="1", MI_ROI_ALIGN_FWD_FULL_WAIT=None)
        a = _forward_ws_raw(f, rois, res, scale, 2, None)
        tuning_env(MI_ROI_ALIGN_SLAB="1", MI_ROI_ALIGN_FWD_FULL_WAIT="1")
        b = _forward_ws_raw(f, rois, res, scale, 2, None)
        assert torch.equal(a, b) and torch.isfinite(a).all()


@pytest.mark.gpu
def test_roi_align_records_free_forward_many_rois(oracle_mod, tuning_env):
    """The records keep one sweep key per RoI in LDS (8192 at most); the records-free forward has no such bound: 20 000 RoIs
    in one call (grid x = 160 000 one-wave workgroups), checked against the oracle."""
    n, c, h, w, scale, r = 2, 16, 50, 84, 1.0 / 16, 20000
    feat = syn.feature_map(n, c, h, w, seed=21)
    rois = syn.rois_canonical(r, n, seed=22, side=(2.0 / scale, 30.0 / scale), im_h=h / scale, im_w=w / scale)
    tuning_env(MI_ROI_ALIGN_SLAB="1")
    got = _forward_ws_raw(to_dev(feat), to_dev(rois), 7, scale, 2, None)
    assert_fwd(got, oracle_mod.roi_align_forward(feat, rois, 7, 7, scale, 2, threads=8), "20000 RoIs", exact=False)


@pytest.mark.gpu
def test_roi_align_records_free_forward_random_shapes(oracle_mod, tuning_env):
    """Forty seeded random cases through the one-launch forward: maps from 2x2 to 60x90, 8 to 80 channels (any multiple of 8),
    1 to 3 images, pooled sizes 1 to 12 (square and not), sampling ratio 0 to 3, RoIs from sub-pixel to beyond the image --
    every one against the oracle at the fast paths' bar."""
    tuning_env(MI_ROI_ALIGN_SLAB="1")
    rng = np.random.RandomState(1234)
    from detectron_pytorch_amd import _lib

    lib = _lib.lib()
    for case in range(40):
        n, c = int(rng.randint(1, 4)), 8 * int(rng.randint(1, 11))
        h, w = int(rng.randint(2, 61)), int(rng.randint(2, 91))
        ph, pw = int(rng.randint(1, 13)), int(rng.randint(1, 13))
        if rng.rand() < 0.5:
            pw = ph
        sr, r = int(rng.randint(0, 4)), int(rng.randint(1, 60))
        scale = float(rng.choice([1.0, 0.5, 0.25, 1.0 / 16]))
        feat = syn.feature_map(n, c, h, w, seed=case)
        x1 = rng.uniform(-4, w + 2, r) / scale
        y1 = rng.uniform(-4, h + 2, r) / scale
        bw = rng.choice([0.3, 2.0, 9.0, 40.0], r) * rng.uniform(0.5, 1.5, r) / scale
        bh = rng.choice([0.3, 2.0, 9.0, 40.0], r) * rng.uniform(0.5, 1.5, r) / scale
        rois = np.stack([rng.randint(0, n, r).astype(np.float64), x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        bad = rois[:2].copy()               # RoIs of no image pool zeros (the reference would read out of bounds)
        bad[:, 0] = (-1, n)[: len(bad)]
        f, rt = to_dev(feat), to_dev(np.vstack([rois, bad]))
        r_all = r + len(bad)
        out = torch.full((r_all, c, ph, pw), float("nan"), device=dev())
        rc = lib.mi_roi_align_forward(f.data_ptr(), rt.data_ptr(), out.data_ptr(), n, c, h, w, r_all, ph, pw, scale, sr, 0, 0,
                                      _lib.current_stream_handle(dev()))
        assert rc == 0, lib.mi_last_error()
        assert not out[r:].any(), "case %d: RoIs of no image must pool zeros" % case
        ref = oracle_mod.roi_align_forward(feat, rois, ph, pw, scale, sr, threads=8)
        assert_fwd(out[:r], ref, "case %d: n %d c %d %dx%d pooled %dx%d sr %d r %d" % (case, n, c, h, w, ph, pw, sr, r), exact=False)
