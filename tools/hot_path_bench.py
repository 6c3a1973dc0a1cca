"""Measurements of the RoIAlign / NMS hot path alone (BASELINE configs 1-2 and the per-kernel roofline), used by
bench.py as sub-objects of its JSON line: the roofline object of the RoIAlign forward (HIP events on the launch stream),
the other RoIAlign shapes of a step, NMS latencies, the post-convolution inference glue.  No oracle import here: the CPU
baseline lives in bench.py."""
import json
import time
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from detectron_pytorch_amd import _lib as _lib_mod, synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


# ------------------------------------------------------------------------------------------------
# Hot-path workload: what one image of e2e_mask_rcnn_R-50-FPN asks of the RoI/NMS operators
# (SURVEY.md section 8a/8d): box-head RoIAlign 512 RoIs x 256 x 7x7 fwd+bwd on P2, mask-head RoIAlign
# 128 RoIs x 256 x 14x14 fwd+bwd, and the 5 per-level RPN NMS calls (n = 2000 pre-NMS, thresh 0.7).
# ------------------------------------------------------------------------------------------------
class HotPath:
    def __init__(self, device, images_per_rank=2, seed=0):
        from detectron_pytorch_amd import nms as mi_nms
        from detectron_pytorch_amd.roi_align import roi_align_backward, roi_align_forward

        self.device = device
        self.images = images_per_rank
        self.fwd, self.bwd, self.nms_many = roi_align_forward, roi_align_backward, mi_nms.nms_device_many
        h, w, scale = syn.FPN_LEVELS[2]
        self.scale = scale
        n = images_per_rank
        self.feat_np = syn.feature_map(n, syn.FPN_DIM, h, w, seed=seed)
        self.feat = torch.from_numpy(self.feat_np).to(device)
        self.box_rois_np = syn.rois_canonical(512 * n, n, seed=seed)
        self.box_rois = torch.from_numpy(self.box_rois_np).to(device)
        self.mask_rois_np = syn.rois_canonical(128 * n, n, seed=seed + 1)
        self.mask_rois = torch.from_numpy(self.mask_rois_np).to(device)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.box_gtop = torch.randn(512 * n, syn.FPN_DIM, 7, 7, generator=g).to(device)
        self.mask_gtop = torch.randn(128 * n, syn.FPN_DIM, 14, 14, generator=g).to(device)
        self.dets = [torch.from_numpy(syn.sort_by_score(syn.boxes_clustered(2000, seed=seed + 10 + i))[0]).to(device)
                     for i in range(5 * n)]

    def step(self):
        fs = tuple(self.feat.shape)
        # forward returns its per-RoI records; the backward over the same RoIs reuses them (as the autograd
        # Function does through ctx)
        out, ws = self.fwd(self.feat, self.box_rois, 7, 7, self.scale, 2, return_workspace=True)
        gin = self.bwd(self.box_gtop, self.box_rois, fs, 7, 7, self.scale, 2, workspace=ws)
        out2, ws2 = self.fwd(self.feat, self.mask_rois, 14, 14, self.scale, 2, return_workspace=True)
        gin2 = self.bwd(self.mask_gtop, self.mask_rois, fs, 14, 14, self.scale, 2, workspace=ws2)
        # the per-level, per-image RPN NMS problems are independent: fanned out over side streams
        keeps = self.nms_many(self.dets, 0.7)
        return out, gin, out2, gin2, keeps


def make_stepper(work, mode, device):
    """The step of the timed loop.  graph: the ~20 launches of one step (two RoIAlign shapes fwd+bwd, batched NMS, their
    allocations) are captured once in a hipGraph on a side stream and replayed -- the step is launch-bound from Python
    otherwise.  Every replay executes all kernels on the same static inputs; the captured outputs are checked against an
    eager step before the graph is trusted.  Falls back to eager launching if capture is not possible."""
    if mode == "eager":
        return work.step, "eager"
    try:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                work.step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
            captured = work.step()
        eager = work.step()
        for t in captured[:4]:
            t.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        for got, want in zip(captured[:4], eager[:4]):
            assert torch.equal(got, want), "hipGraph replay does not reproduce the eager step"
        for (keep_g, num_g), (keep_e, num_e) in zip(captured[4], eager[4]):
            k = int(num_e.item())
            assert int(num_g.item()) == k and torch.equal(keep_g[:k], keep_e[:k]), "hipGraph replay: NMS differs"
        work.captured = captured  # keep the static outputs alive
        return graph.replay, "hipGraph"
    except Exception as exc:  # capture is an optimisation of the harness, not of the measured kernels
        sys.stderr.write("bench: hipGraph capture failed (%s: %s); launching eagerly\n" % (type(exc).__name__, exc))
        torch.cuda.synchronize()
        return work.step, "eager"


def time_kernel(fn, iters, warmup=10, trace=None):
    """Average duration (s) of one call of `fn` over `iters` back-to-back launches, HIP events recorded
    on the stream the kernels are launched on (torch's current stream).  STEADY STATE: batches of `iters` calls are
    repeated (at most 8) until two consecutive batches agree within 1 % -- after idle the chip needs ~30 ms of sustained work
    before its clocks have ramped (a 29 us kernel measures 34.6, 31.9, 30.3, 29.6, 29.4 ... 28.9 us over the first batches of
    200 calls: tools/_ramp.py, profiles/r06_slab_forward.txt); `trace` (a list) receives every batch's figure in us."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    prev = None
    for _ in range(8):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            fn()
        stop.record()
        stop.synchronize()
        t = start.elapsed_time(stop) * 1e-3 / iters
        if trace is not None:
            trace.append(round(t * 1e6, 2))
        if prev is not None and abs(t - prev) <= 0.01 * t:
            break
        prev = t
    return t


def roofline_roi_align_forward(device, iters):
    """BASELINE configs[1]: RoIAlign forward, 512 RoIs x 256 ch x 7x7, sampling_ratio 2, P2 map of one image.
    Algorithmic bytes: 4*R*C*PH*PW (write) + 4*C*U (read, U distinct pixels) + 20*R."""
    from detectron_pytorch_amd import _lib

    h, w, scale = syn.FPN_LEVELS[2]
    c, r, res, sr = syn.FPN_DIM, 512, 7, 2
    feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(device)
    rois_np = syn.rois_canonical(r, 1, seed=0)
    if os.environ.get("MI_BENCH_SORT_ROIS"):  # tuning experiment only: spatially sorted RoI order
        key = (rois_np[:, 2] + rois_np[:, 4]) // (2 * 64) * 4096 + (rois_np[:, 1] + rois_np[:, 3]) / 2
        rois_np = np.ascontiguousarray(rois_np[np.argsort(key, kind="stable")])
    rois = torch.from_numpy(rois_np).to(device)
    out = torch.empty((r, c, res, res), device=device)
    lib = _lib.lib()
    stream = _lib.current_stream_handle(device)

    layout = _lib.LAYOUT_NCHW
    if os.environ.get("MI_BENCH_NHWC"):  # tuning experiment only: channels_last storage of the same logical tensor
        feat = feat.permute(0, 2, 3, 1).contiguous()
        layout = _lib.LAYOUT_NHWC

    ws_bytes = lib.mi_roi_align_forward_workspace_bytes(r)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)

    def launch():  # one call of the C-ABI = ONE launch (roi_align_fwd_slab: a forward-sized workspace asks for no records)
        rc = lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res,
                                         scale, sr, _lib.ROI_ALIGN_CAFFE2, layout, ws.data_ptr(), ws_bytes, stream)
        assert rc == 0

    batches = []
    seconds = time_kernel(launch, iters, trace=batches)
    ready_fwd = bool(lib.mi_roi_align_forward_writes_records(c, h, w, r, res, res, _lib.ROI_ALIGN_CAFFE2, layout))
    touched = touched_pixels(rois_np, 1, h, w, res, res, scale, sr)
    alg_bytes = 4 * r * c * res * res + 4 * c * touched + 20 * r
    achieved = alg_bytes / seconds / 1e9
    traffic, traffic_src = pmc_traffic("forward")
    info = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            # "bound" names the roofline the fraction is quoted against (the contract's enum).  What LIMITS the kernel is
            # not HBM (1.2 x the algorithmic bytes): the gather alone (bins and stores ablated) takes 16.6 us + the launch floor
            # whatever the residency (14 or 18 waves per CU) -- 3.4 M cache lines = 5.2 x the algorithmic bytes cross
            # L2 -> L1 at ~77 % of the 64 B / clk / CU fill rate, a 72-byte NCHW row segment dragging in 1.5 lines -- and the
            # bins (LDS reads, 1.7 passes per tap read after the stride-4 layout) and stores (straight from the lanes) overlap it only in part
            "limited_by": "L2->L1 line fill of the per-RoI window gather (cache lines = 5.2 x algorithmic bytes; the gather alone "
                          "runs at ~77 % of the L1 fill rate) + LDS tap reads and stores that overlap it only partly; not HBM "
                          "(profiles/r06_slab_forward.txt)",
            "kernel": "roi_align_fwd_slab (one launch per mi_roi_align_forward_ws call: records-free, one wave per (RoI, 8 channels), "
                      "an XCD reads one 8-channel slab at a time)",
            "shape": "R=512 C=256 7x7 sr=2 on 200x336", "algorithmic_bytes": int(alg_bytes),
            "avg_launch_us": round(seconds * 1e6, 2), "launches": iters,
            "timing": "steady state: batches of %d back-to-back calls until two consecutive batches agree within 1 %% (the "
                      "chip's clocks ramp over the first ~30 ms after idle); us per call of every batch: %s" % (iters, batches)}
    # The record-driven pair (what a training forward runs: its workspace has room for the backward, which reads the
    # records): the full call, and its gather kernel alone over records already in the workspace -- what a caller pays whose
    # RoI producer writes the records (mi_rpn_collect_finish_records; inference_path.producer_records_pair)
    if ready_fwd and not os.environ.get("MI_BENCH_NHWC"):
        import ctypes

        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes as _bwsb

        rws_bytes = max(ws_bytes, _bwsb([(h, w)], 1, r))
        rws = torch.empty(rws_bytes, dtype=torch.uint8, device=device)
        first = out.clone()

        def launch_records():
            rc = lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res,
                                             scale, sr, _lib.ROI_ALIGN_CAFFE2, layout, rws.data_ptr(), rws_bytes, stream)
            assert rc == 0

        sec_records = time_kernel(launch_records, iters)
        assert torch.equal(out, first), "the record-driven forward must reproduce the records-free call bit for bit"
        lvt = _lib.FpnLevels()
        lvt.num_levels = 1
        lvt.features[0], lvt.height[0], lvt.width[0], lvt.spatial_scale[0] = feat.data_ptr(), h, w, scale
        level0 = torch.zeros(r, dtype=torch.int32, device=device)

        def launch_ready():
            rc = lib.mi_roi_align_forward_fpn_records(ctypes.byref(lvt), rois.data_ptr(), level0.data_ptr(), out.data_ptr(), 1, c,
                                                      r, res, res, sr, layout, rws.data_ptr(), rws_bytes, stream)
            assert rc == 0

        sec_ready = time_kernel(launch_ready, iters)
        assert torch.equal(out, first), "the records-ready forward must reproduce the full call bit for bit"
        info["records_pair"] = {"avg_launch_us": round(sec_records * 1e6, 2), "achieved": round(alg_bytes / sec_records / 1e9, 1),
                                "unit": "GB/s", "frac": round(alg_bytes / sec_records / 1e9 / HBM_PEAK_GBS, 4),
                                "what": "roi_align_prepare (with the backward's tables) + roi_align_fwd_records: the same call with a "
                                        "workspace that has room for a backward -- the training forward; rounds 1-5 and early round 6 "
                                        "quoted this pair (with forward-only records) as the headline fraction"}
        info["records_ready"] = {"avg_launch_us": round(sec_ready * 1e6, 2), "achieved": round(alg_bytes / sec_ready / 1e9, 1),
                                 "unit": "GB/s", "frac": round(alg_bytes / sec_ready / 1e9 / HBM_PEAK_GBS, 4),
                                 "what": "roi_align_fwd_records alone (mi_roi_align_forward_fpn_records over the records the "
                                         "record-driven call has just written): the call of a consumer whose RoI producer writes "
                                         "the records"}
    # backward at the same shape, reported beside it (bytes = 4*R*C*PH*PW read + 4*N*C*H*W written + 20*R)
    gtop = torch.randn(r, c, res, res, device=device)
    gin = torch.zeros(1, c, h, w, device=device)

    overwrite = bool(lib.mi_roi_align_backward_overwrites(c, h, w, r, res, res, _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW))
    # the NCHW forward (tile-centric) leaves no records: the backward call then includes its own records launch
    ready = bool(lib.mi_roi_align_forward_writes_records(c, h, w, r, res, res, _lib.ROI_ALIGN_CAFFE2, layout))
    bwd_flags = (_lib.ROI_ALIGN_RECORDS_READY if ready else 0) | (_lib.ROI_ALIGN_OVERWRITE if overwrite else 0)

    # the workspace the autograd Function allocates: records + room for the backward plan (lists per gradient tile, cut
    # into slices for the clustered RoIs of a training step); with the forward's size the backward runs unplanned
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    bws_bytes = max(ws_bytes, _backward_workspace_bytes([(h, w)], 1, r))
    bws = torch.empty(bws_bytes, dtype=torch.uint8, device=device)

    def launch_bwd(space=bws, size=bws_bytes, flags=bwd_flags & ~_lib.ROI_ALIGN_RECORDS_READY):
        if not overwrite:  # zero fill only where the path accumulates
            gin.zero_()
        rc = lib.mi_roi_align_backward_ws(gtop.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res,
                                          scale, sr, _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW, space.data_ptr(), size,
                                          flags, stream)
        assert rc == 0

    launch_bwd()  # leaves the records (with their backward block) in `bws`
    sec_bwd = time_kernel(lambda: launch_bwd(flags=bwd_flags), max(iters // 4, 10))
    # the unplanned kernel (one workgroup per tile walks the whole list): MI_ROI_ALIGN_BWD_SLICE=0, records ready
    prev = os.environ.get("MI_ROI_ALIGN_BWD_SLICE")
    os.environ["MI_ROI_ALIGN_BWD_SLICE"] = "0"
    lib.mi_dbg_reload_tuning()
    launch_bwd()
    sec_unplanned = time_kernel(lambda: launch_bwd(flags=bwd_flags), max(iters // 4, 10))
    if prev is None:
        del os.environ["MI_ROI_ALIGN_BWD_SLICE"]
    else:
        os.environ["MI_ROI_ALIGN_BWD_SLICE"] = prev
    lib.mi_dbg_reload_tuning()
    bwd_bytes = 4 * r * c * res * res + 4 * c * h * w + 20 * r
    info["backward"] = {"zero_fill_needed": not overwrite, "avg_us_incl_zero_fill": round(sec_bwd * 1e6, 2),
                        "achieved": round(bwd_bytes / sec_bwd / 1e9, 1), "unit": "GB/s",
                        "algorithmic_bytes": int(bwd_bytes), "planned": bws_bytes > ws_bytes,
                        "unplanned_us": round(sec_unplanned * 1e6, 2),
                        "what": "planned = roi_align_bwd_plan + _tiles (two launches) with the workspace the autograd Function "
                                "allocates (the plan files every tile's list in a cost class -- longest first -- and cuts the "
                                "long lists of a training step's clustered RoIs into slices: roi_align_step_rois); unplanned = "
                                "the same call under MI_ROI_ALIGN_BWD_SLICE=0 (one workgroup per tile scans the RoIs itself and "
                                "walks the whole list: no atomics, bit-reproducible; + roi_align_bwd_untabled behind it)"}
    info["l2_line_frac"] = pmc_l2_line_frac(info["avg_launch_us"])  # one launch per call: the call time is the kernel plus its launch floor
    info["other_shapes"] = other_shapes(device, lib, stream, max(iters // 4, 10))
    info["roi_pool_roi_crop"] = pool_and_crop(device, max(iters // 2, 20))
    if layout == _lib.LAYOUT_NCHW:
        info["channels_last"] = channels_last_variant(device, lib, stream, feat, rois, out, ws, alg_bytes, gtop, iters)
    if layout == _lib.LAYOUT_NCHW:
        info["cold_cache"] = cold_cache_variant(device, lib, stream, feat, rois, ws, ws_bytes, alg_bytes, r, c, h, w, res,
                                                scale, sr, max(iters // 2, 20))
        # `frac` above is measured on ONE map visited back to back (it stays in the 256 MB Infinity Cache); a step that
        # pools a map a convolution has just written sees something between the two
        info["cold_cache_frac"] = info["cold_cache"]["frac"]
        info["frac_is"] = "warm: the same feature map every call; cold_cache_frac: six maps round-robin"
    copy_gbs, torch_gbs = copy_ceiling(device)
    info["copy_ceiling"] = {"measured": round(copy_gbs, 1), "unit": "GB/s", "frac_of_copy": round(achieved / copy_gbs, 4),
                            "ceiling_frac_of_peak": round(copy_gbs / HBM_PEAK_GBS, 3),
                            "torch_d2d_copy": round(torch_gbs, 1),
                            "what": "mi_dbg_copy_float4 (16 B / lane, non-temporal, 4 workgroups / CU) over 256 MiB, read + write bytes / "
                                    "time; torch_d2d_copy: the same buffers through torch's copy_ (rounds 1-3 quoted this one)"}
    return info


def channels_last_variant(device, lib, stream, feat_nchw, rois, out, ws, alg_bytes, gtop, iters):
    """Same logical input with the features stored channels-last (what MIOpen's NHWC convolutions hand over on gfx950):
    forward = roi_align_prepare + roi_align_fwd_nhwc, output still dense [R,C,PH,PW]; backward = the tile kernel writing
    the gradient channels-last (through roi_align.roi_align_backward, allocation included).  Reported beside the NCHW
    headline, not as it."""
    from detectron_pytorch_amd import _lib, roi_align as ra

    n, c, h, w = feat_nchw.shape
    r, _, res, _ = out.shape
    feat = feat_nchw.permute(0, 2, 3, 1).contiguous()
    scale, sr = syn.FPN_LEVELS[2][2], 2

    def launch():
        rc = lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), n, c, h, w, r, res, res,
                                         scale, sr, _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NHWC, ws.data_ptr(), ws.numel(),
                                         stream)
        assert rc == 0

    sec = time_kernel(launch, iters)

    def bwd():  # records of the forward above are reused; the tile kernel writes the channels-last gradient itself
        ra.roi_align_backward(gtop, rois, (n, c, h, w), res, res, scale, sr, channels_last=True, workspace=ws)

    sec_bwd = time_kernel(bwd, max(iters // 8, 5))
    gbs = alg_bytes / sec / 1e9
    return {"kernel": "roi_align_prepare + roi_align_fwd_nhwc", "avg_launch_us": round(sec * 1e6, 2),
            "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "bwd_us": round(sec_bwd * 1e6, 2)}


def cold_cache_variant(device, lib, stream, feat, rois, ws, ws_bytes, alg_bytes, r, c, h, w, res, scale, sr, iters):
    """The headline call with the 256 MB Infinity Cache taken out of the picture: six distinct copies of the feature map
    (6 x 68.8 MB = 413 MB) and six output buffers (6 x 25.7 MB) are visited round-robin, so that by the time a map comes
    round again 567 MB of other traffic have passed through the MALL.  Same RoIs, same kernels, same timing method."""
    from detectron_pytorch_amd import _lib

    copies = 6
    feats = [feat.clone() for _ in range(copies)]
    outs = [torch.empty((r, c, res, res), device=device) for _ in range(copies)]
    state = {"i": 0}

    def launch():
        i = state["i"] = (state["i"] + 1) % copies
        rc = lib.mi_roi_align_forward_ws(feats[i].data_ptr(), rois.data_ptr(), outs[i].data_ptr(), 1, c, h, w, r, res, res,
                                         scale, sr, _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW, ws.data_ptr(), ws_bytes, stream)
        assert rc == 0

    sec = time_kernel(launch, iters)
    gbs = alg_bytes / sec / 1e9
    return {"avg_launch_us": round(sec * 1e6, 2), "achieved": round(gbs, 1), "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "distinct_feature_maps": copies,
            "rotated_bytes": int(copies * (feat.numel() + outs[0].numel()) * 4),
            "what": "same call over %d feature maps / outputs visited round-robin (> 256 MB between two visits of a map)"
                    % copies}


def copy_ceiling(device):
    """The box's own streaming ceiling (SURVEY.md section 8d asks for both denominators): the library's float4 copy kernel
    (mi_dbg_copy_float4: 16 bytes per lane, non-temporal; read + write bytes of a 256 MiB buffer) and, beside it,
    what torch's device-to-device copy reaches (the denominator of rounds 1-3, ~14 % lower)."""
    from detectron_pytorch_amd import _lib

    a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    lib, stream = _lib.lib(), _lib.current_stream_handle(device)
    nbytes = a.numel() * 4

    def launch():
        assert lib.mi_dbg_copy_float4(a.data_ptr(), b.data_ptr(), nbytes, stream) == 0

    sec = time_kernel(launch, 50)
    sec_torch = time_kernel(lambda: b.copy_(a), 50)
    return 2 * nbytes / sec / 1e9, 2 * nbytes / sec_torch / 1e9


def other_shapes(device, lib, stream, iters):
    """Per-call times of the other RoIAlign shapes of the step (not roofline-gated): the mask head (128 x 256 x 14x14) and
    the two-image box head (1024 RoIs, N = 2), forward and backward through the workspace entry points."""
    from detectron_pytorch_amd import _lib

    out = {}
    h, w, scale = syn.FPN_LEVELS[2]
    c, sr = syn.FPN_DIM, 2
    for name, n, r, res in [("mask_128x256x14x14", 1, 128, 14), ("box_1024x256x7x7_2img", 2, 1024, 7)]:
        feat = torch.from_numpy(syn.feature_map(n, c, h, w, seed=0)).to(device)
        rois = torch.from_numpy(syn.rois_canonical(r, n, seed=1)).to(device)
        o = torch.empty((r, c, res, res), device=device)
        gtop = torch.randn(r, c, res, res, device=device)
        gin = torch.empty(n, c, h, w, device=device)
        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

        ws_bytes = max(lib.mi_roi_align_forward_workspace_bytes(r), _backward_workspace_bytes([(h, w)], n, r))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        over = bool(lib.mi_roi_align_backward_overwrites(c, h, w, r, res, res, _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW))
        ready = bool(lib.mi_roi_align_forward_writes_records(c, h, w, r, res, res, _lib.ROI_ALIGN_CAFFE2, _lib.LAYOUT_NCHW))
        flags = (_lib.ROI_ALIGN_RECORDS_READY if ready else 0) | (_lib.ROI_ALIGN_OVERWRITE if over else 0)

        def fwd():
            assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), o.data_ptr(), n, c, h, w, r, res, res, scale,
                                               sr, 0, 0, ws.data_ptr(), ws_bytes, stream) == 0

        def bwd():
            if not over:
                gin.zero_()
            assert lib.mi_roi_align_backward_ws(gtop.data_ptr(), rois.data_ptr(), gin.data_ptr(), n, c, h, w, r, res, res,
                                                scale, sr, 0, 0, ws.data_ptr(), ws_bytes, flags, stream) == 0

        fws_bytes = lib.mi_roi_align_forward_workspace_bytes(r)  # an inference call: no room for a backward -> records-free

        def fwd_only():
            assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), o.data_ptr(), n, c, h, w, r, res, res, scale,
                                               sr, 0, 0, ws.data_ptr(), fws_bytes, stream) == 0

        fwd_only_s = time_kernel(fwd_only, iters)
        fwd_s, bwd_s = time_kernel(fwd, iters), time_kernel(bwd, iters)
        # the same algorithmic-bytes formulas as the headline shape: forward 4 R C PH PW + 4 C U + 20 R
        # with U = distinct tapped pixels of THIS RoI set, backward 4 R C PH PW + 4 N C H W + 20 R
        u = touched_pixels(rois.cpu().numpy(), n, h, w, res, res, scale, sr)
        fwd_bytes = 4 * r * c * res * res + 4 * c * u + 20 * r
        bwd_bytes = 4 * r * c * res * res + 4 * n * c * h * w + 20 * r
        out[name] = {"fwd_only_us": round(fwd_only_s * 1e6, 1), "fwd_only_frac": round(fwd_bytes / fwd_only_s / 1e9 / HBM_PEAK_GBS, 4),
                     "fwd_us": round(fwd_s * 1e6, 1), "bwd_us": round(bwd_s * 1e6, 1),
                     "what": "fwd_only: forward-sized workspace (inference; roi_align_fwd_slab, one launch); fwd / bwd: the training pair "
                             "over a workspace with room for the backward (records + gather, then plan + tiles)",
                     "touched_pixels": int(u), "fwd_algorithmic_bytes": int(fwd_bytes),
                     "fwd_achieved": round(fwd_bytes / fwd_s / 1e9, 1), "fwd_frac": round(fwd_bytes / fwd_s / 1e9 / HBM_PEAK_GBS, 4),
                     "bwd_algorithmic_bytes": int(bwd_bytes), "bwd_achieved": round(bwd_bytes / bwd_s / 1e9, 1),
                     "bwd_frac": round(bwd_bytes / bwd_s / 1e9 / HBM_PEAK_GBS, 4), "unit": "GB/s"}
    out["fpn_1000rois_P2-P5_7x7"] = fpn_variant(device, iters)
    return out


def fpn_variant(device, iters):
    """Config-2 variant (ii): 1000 RoIs distributed over P2..P5 by the FPN heuristic, pooled by
    roi_xform.roi_feature_transform (one RoIAlign call per level + concat + restore permutation), RoIs on the device."""
    from detectron_pytorch_amd import roi_xform

    rois, lvls = syn.rois_fpn_distributed(1000, batch=1, seed=2)
    blobs = roi_xform.add_multilevel_roi_blobs({"rois": rois}, "rois", rois, lvls, 2, 5)
    blobs = {k: torch.from_numpy(v).to(device) for k, v in blobs.items()}
    feats, scales = [], []
    for lvl in (5, 4, 3, 2):  # coarsest first, as the reference orders blobs_in
        h, w, scale = syn.FPN_LEVELS[lvl]
        feats.append(torch.from_numpy(syn.feature_map(1, syn.FPN_DIM, h, w, seed=lvl)).to(device))
        scales.append(scale)

    def fwd(fused):
        with torch.no_grad():
            return roi_xform.roi_feature_transform(feats, blobs, "rois", "RoIAlign", 7, scales, 2, fused=fused)

    # fused call with the RoIs already in dataloader order (what a caller that keeps them on the device passes)
    from detectron_pytorch_amd.roi_align import roi_align_fpn

    rois_d = torch.from_numpy(rois).to(device)
    lvl_d = torch.from_numpy((5 - lvls).astype(np.int32)).to(device)

    def fused_direct():
        with torch.no_grad():
            roi_align_fpn(feats, scales, rois_d, lvl_d, 7, 7, 2)

    # algorithmic bytes of the pyramid call: every level's RoIs tap their own map
    u_total = 0
    for lvl in (2, 3, 4, 5):
        hh, ww_, sc = syn.FPN_LEVELS[lvl]
        sel = rois[lvls == lvl]
        if len(sel):
            u_total += touched_pixels(sel, 1, hh, ww_, 7, 7, sc, 2)
    alg = 4 * len(rois) * syn.FPN_DIM * 49 + 4 * syn.FPN_DIM * u_total + 20 * len(rois)
    fused_s = time_kernel(fused_direct, iters)
    return {"fwd_us": round(time_kernel(lambda: fwd(True), iters) * 1e6, 1),
            "fwd_us_per_level_loop": round(time_kernel(lambda: fwd(False), iters) * 1e6, 1),
            "fwd_us_fused_call_only": round(fused_s * 1e6, 1),
            "touched_pixels": int(u_total), "fwd_algorithmic_bytes": int(alg),
            "fwd_achieved": round(alg / fused_s / 1e9, 1), "fwd_frac": round(alg / fused_s / 1e9 / HBM_PEAK_GBS, 4),
            "unit": "GB/s (of the fused call)",
            "rois_per_level": {int(l): int((lvls == l).sum()) for l in (2, 3, 4, 5)}}


def pool_and_crop(device, iters):
    """RoIPool and RoICrop (roi_pool.hip, roi_crop.hip) at the config-2 shape -- 512 RoIs x 256 channels x 7x7 on the
    200 x 336 P2 map of one image -- per C-ABI call, HIP events.  Algorithmic bytes, by the rule of section 8d (compulsory
    traffic): RoIPool forward 8 R C PH PW (values + int32 argmax written) + 4 C U (U = distinct pixels inside the RoIs' bins)
    + 20 R; RoICrop forward 4 R C GH GW written + 8 R GH GW (grid) + 4 C U (U = distinct in-image taps); the backwards
    read the output-sized gradient (+ the argmax) and write the zero-filled map once: + 4 N C H W (the caller's fill, timed
    with the call as the reference's functions do it)."""
    from detectron_pytorch_amd import _lib

    lib, stream = _lib.lib(), _lib.current_stream_handle(device)
    h, w, scale = syn.FPN_LEVELS[2]
    c, r, res = syn.FPN_DIM, 512, 7
    feat_np = syn.feature_map(1, c, h, w, seed=0)
    rois_np = syn.rois_canonical(r, 1, seed=0)
    if os.environ.get("MI_BENCH_SORT_ROIS"):  # tuning experiment only: RoIs in a spatial sweep order
        key = (rois_np[:, 2] + rois_np[:, 4]) // (2 * 64) * 4096 + (rois_np[:, 1] + rois_np[:, 3]) / 2
        rois_np = np.ascontiguousarray(rois_np[np.argsort(key, kind="stable")])
    feat, rois = torch.from_numpy(feat_np).to(device), torch.from_numpy(rois_np).to(device)
    out = torch.empty((r, c, res, res), device=device)
    argmax = torch.empty((r, c, res, res), dtype=torch.int32, device=device)
    gtop = torch.randn(r, c, res, res, device=device)
    gin = torch.empty(1, c, h, w, device=device)
    result = {"shape": "R=512 C=256 7x7 on 200x336 (config 2's inputs)", "unit": "GB/s"}

    def entry(sec, nbytes):
        return {"us": round(sec * 1e6, 2), "algorithmic_bytes": int(nbytes), "achieved": round(nbytes / sec / 1e9, 1),
                "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- RoIPool ----
    def pool_fwd():
        assert lib.mi_roi_pool_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), argmax.data_ptr(), 1, c, h, w, r, res,
                                       res, scale, stream) == 0

    def pool_bwd():   # overwrites every element (roi_pooling_kernel.cu:202): no fill
        assert lib.mi_roi_pool_backward(gtop.data_ptr(), rois.data_ptr(), argmax.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res,
                                        res, scale, stream) == 0

    x1 = np.clip(np.floor(rois_np[:, 1] * scale + 0.5), 0, w).astype(int)   # round half up == roundf for these positive values
    y1 = np.clip(np.floor(rois_np[:, 2] * scale + 0.5), 0, h).astype(int)
    x2 = np.clip(np.floor(rois_np[:, 3] * scale + 0.5) + 1, 0, w).astype(int)
    y2 = np.clip(np.floor(rois_np[:, 4] * scale + 0.5) + 1, 0, h).astype(int)
    seen = np.zeros((h, w), bool)
    for a, b, cc, d in zip(y1, y2, x1, x2):
        seen[a:b, cc:d] = True
    u_pool = int(seen.sum())
    out_bytes = 4 * r * c * res * res
    result["roi_pool_fwd"] = dict(entry(time_kernel(pool_fwd, iters), 2 * out_bytes + 4 * c * u_pool + 20 * r),
                                  kernel="roi_pool_fwd", distinct_pixels=u_pool)
    result["roi_pool_bwd"] = dict(entry(time_kernel(pool_bwd, max(iters // 4, 10)), 2 * out_bytes + 4 * c * h * w + 20 * r),
                                  kernel="roi_pool_bwd_tiles (LDS accumulators per 8x32 tile, the reference's addition order, no fill, no global atomics)")
    # ---- RoICrop: the affine grids of the same 512 boxes (what model_builder.py:279-287 builds from the RoIs) ----
    cx = (rois_np[:, 1] + rois_np[:, 3]) * 0.5 * scale / (w - 1) * 2 - 1
    cy = (rois_np[:, 2] + rois_np[:, 4]) * 0.5 * scale / (h - 1) * 2 - 1
    sx = (rois_np[:, 3] - rois_np[:, 1]) * 0.5 * scale / (w - 1) * 2
    sy = (rois_np[:, 4] - rois_np[:, 2]) * 0.5 * scale / (h - 1) * 2
    lin = np.linspace(-1, 1, res, dtype=np.float64)
    gy = cy[:, None, None] + sy[:, None, None] * lin[None, :, None] + 0 * lin[None, None, :]
    gx = cx[:, None, None] + sx[:, None, None] * lin[None, None, :] + 0 * lin[None, :, None]
    grid_np = np.stack([gy, gx], axis=3).astype(np.float32)
    grid = torch.from_numpy(grid_np).to(device)
    ty = np.floor((grid_np[..., 0].astype(np.float64) + 1) * (h - 1) / 2).astype(int)
    tx = np.floor((grid_np[..., 1].astype(np.float64) + 1) * (w - 1) / 2).astype(int)
    seen = np.zeros((h + 2, w + 2), bool)
    for dy in (0, 1):
        for dx in (0, 1):
            yy, xx = np.clip(ty + dy, -1, h) + 1, np.clip(tx + dx, -1, w) + 1
            seen[yy, xx] = True
    u_crop = int(seen[1:h + 1, 1:w + 1].sum())

    def crop_fwd():
        out.zero_()   # functions/roi_crop.py:11: the zero fill is part of the reference's forward
        assert lib.mi_roi_crop_forward(feat.data_ptr(), grid.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, stream) == 0

    def crop_bwd_atomics():
        gin.zero_()
        assert lib.mi_roi_crop_backward(feat.data_ptr(), grid.data_ptr(), gtop.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res,
                                        stream) == 0

    crop_ws = torch.empty(lib.mi_roi_crop_backward_workspace_bytes(r), dtype=torch.uint8, device=device) if hasattr(lib, "mi_roi_crop_backward_ws") else None

    def crop_bwd():   # the tile form: overwrites, no fill
        assert lib.mi_roi_crop_backward_ws(feat.data_ptr(), grid.data_ptr(), gtop.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res,
                                           crop_ws.data_ptr(), crop_ws.numel(), stream) == 0

    result["roi_crop_fwd"] = dict(entry(time_kernel(crop_fwd, iters), out_bytes + 8 * r * res * res + 4 * c * u_crop),
                                  kernel="zero fill + roi_crop_fwd", distinct_pixels=u_crop)
    if crop_ws is not None:
        result["roi_crop_bwd"] = dict(entry(time_kernel(crop_bwd, max(iters // 4, 10)), out_bytes + 8 * r * res * res + 4 * c * h * w),
                                      kernel="roi_crop_boxes + roi_crop_bwd_tiles (LDS accumulators per 8x32 tile, no fill, no global atomics)")
    result["roi_crop_bwd_atomics"] = dict(entry(time_kernel(crop_bwd_atomics, max(iters // 4, 10)), out_bytes + 8 * r * res * res + 4 * c * h * w),
                                          kernel="zero fill + roi_crop_bwd (four global atomics per output element, as the reference; the entry point without a workspace)")
    # ---- legacy RoIAlign (row a3: one bilinear point per bin at the corners of an (aligned - 1) grid, model/roi_align), same
    # inputs; algorithmic bytes by the same rule: output + 4 C U (U = distinct in-image taps) + 20 R ----
    def legacy(fn, src, dst):
        def run():
            assert fn(src.data_ptr(), rois.data_ptr(), dst.data_ptr(), 1, c, h, w, r, res, res, scale, 0, _lib.ROI_ALIGN_LEGACY,
                      _lib.LAYOUT_NCHW, stream) == 0, lib.mi_last_error()
        return run

    sw, sh = rois_np[:, 1] * np.float32(scale), rois_np[:, 2] * np.float32(scale)
    bw = np.maximum(rois_np[:, 3] * np.float32(scale) - sw + 1, 0) / (res - 1)
    bh = np.maximum(rois_np[:, 4] * np.float32(scale) - sh + 1, 0) / (res - 1)
    py = sh[:, None] + np.arange(res)[None, :] * bh[:, None]
    px = sw[:, None] + np.arange(res)[None, :] * bw[:, None]
    seen = np.zeros((h, w), bool)
    for yy, xx in zip(py, px):
        ys = np.minimum(np.floor(yy[(yy >= 0) & (yy < h)]), h - 2).astype(int)
        xs = np.minimum(np.floor(xx[(xx >= 0) & (xx < w)]), w - 2).astype(int)
        for dy in (0, 1):
            for dx in (0, 1):
                seen[np.ix_(ys + dy, xs + dx)] = True
    u_legacy = int(seen.sum())
    legacy_bwd = legacy(lib.mi_roi_align_backward, gtop, gin)

    def legacy_bwd_filled():
        gin.zero_()   # the reference's functions zero the gradient in front of the kernel (functions/roi_align.py:39)
        legacy_bwd()

    result["roi_align_legacy_fwd"] = dict(entry(time_kernel(legacy(lib.mi_roi_align_forward, feat, out), iters),
                                                out_bytes + 4 * c * u_legacy + 20 * r),
                                          kernel="roi_align_legacy<fwd> (per-workgroup point table, lanes over (channel, point))",
                                          distinct_pixels=u_legacy)
    result["roi_align_legacy_bwd"] = dict(entry(time_kernel(legacy_bwd_filled, max(iters // 4, 10)), out_bytes + 4 * c * h * w + 20 * r),
                                          kernel="zero fill + roi_align_legacy<bwd> (four global atomics per output element, as the reference)")
    return result


def pmc_l2_line_frac(kernel_us):
    """L2 -> L1 cache-line traffic of the forward kernel as a fraction of the L2's rate: TCP_TCC_READ_REQ x 128 B of the newest
    committed PMC summary / the kernel's duration / 34.5 TB/s (MI355X_MICROARCH.md: aggregate L2 read bandwidth)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_roi_align.json")))
    if not files or not kernel_us:
        return None
    try:
        with open(files[-1]) as f:
            kernels = json.load(f)["forward"]["kernels"]
        req = max(v.get("TCP_TCC_READ_REQ_sum", 0.0) for v in kernels.values())
        line_bytes = req * 128.0
        return {"l2_to_l1_line_bytes": int(line_bytes), "frac_of_l2_rate": round(line_bytes / (kernel_us * 1e-6) / 34.5e12, 3),
                "l2_peak": "34.5 TB/s", "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def pmc_traffic(direction):
    """HBM bytes per call from the newest committed PMC summary (profiles/rNN_pmc_roi_align.json), or None."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_roi_align.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return int(d[direction]["hbm_bytes_per_call"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def touched_pixels(rois_np, batch, h, w, ph, pw, scale, sr):
    """U of the algorithmic-bytes formula (a workload descriptor computed on the host, not timed)."""
    return syn.roi_align_touched_pixels(rois_np, batch, h, w, ph, pw, scale, sr)


def nms_latency(device, iters):
    from detectron_pytorch_amd import _lib

    out = {}
    lib = _lib.lib()
    stream = _lib.current_stream_handle(device)
    for name, dets_np, thresh in [("cfg1_uniform_n1000_t0.5", syn.boxes_uniform(1000, seed=0), 0.5),
                                  ("rpn_clustered_n2000_t0.7", syn.boxes_clustered(2000, seed=0), 0.7)]:
        dets = torch.from_numpy(dets_np).to(device)
        n = dets.shape[0]
        keep = torch.empty(n, dtype=torch.int64, device=device)
        num = torch.empty(1, dtype=torch.int32, device=device)
        ws_bytes = lib.mi_nms_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)

        def launch():
            rc = lib.mi_nms(dets.data_ptr(), n, thresh, _lib.NMS_GE_ORIG_ASC, keep.data_ptr(), num.data_ptr(),
                            ws.data_ptr(), ws_bytes, stream)
            assert rc == 0

        sec = time_kernel(launch, max(iters // 4, 10))
        out[name] = {"us_per_call": round(sec * 1e6, 1), "kept": int(num.item()),
                     "pair_tests_per_s": round(n * (n - 1) / 2 / sec, 0)}
    # Soft-NMS (off by default in the reference, core/config.py:362): one sequential pick per kept box, all in LDS
    dets = torch.from_numpy(syn.boxes_uniform(1000, seed=0)).to(device)
    od, oi = torch.empty((1000, 5), device=device), torch.empty(1000, dtype=torch.int64, device=device)
    num = torch.empty(1, dtype=torch.int32, device=device)

    def launch_soft():
        assert lib.mi_soft_nms(dets.data_ptr(), 1000, 0.5, 0.3, 0.001, 1, od.data_ptr(), oi.data_ptr(), num.data_ptr(),
                               stream) == 0

    sec = time_kernel(launch_soft, 5, warmup=2)
    out["soft_nms_linear_uniform_n1000"] = {"us_per_call": round(sec * 1e6, 1), "kept": int(num.item())}
    # IoU matrix (utils/cython_bbox.pyx:32-73): the labelling shape of a step (2000 proposals x 8 gt boxes) and a square one
    for name, nb, nq in (("bbox_overlaps_2000x8", 2000, 8), ("bbox_overlaps_1000x1000", 1000, 1000)):
        b = torch.from_numpy(syn.boxes_uniform(nb, seed=1)[:, :4].copy()).to(device)
        q = torch.from_numpy(syn.boxes_uniform(nq, seed=2)[:, :4].copy()).to(device)
        o = torch.empty((nb, nq), device=device)
        sec = time_kernel(lambda: lib.mi_bbox_overlaps(b.data_ptr(), nb, q.data_ptr(), nq, o.data_ptr(), stream), 50)
        out[name] = {"us_per_call": round(sec * 1e6, 2), "pairs_per_s": round(nb * nq / sec, 0)}
    # the test-time caller of NMS (core/test.py:732-790): 1000 RoIs x 81 classes, classes batched, two host syncs
    from detectron_pytorch_amd import detection

    sc_np, bx_np = syn.detection_head_outputs(1000, 81, seed=7)
    sc, bx = torch.from_numpy(sc_np).to(device), torch.from_numpy(bx_np).to(device)
    post = {}
    for name, soft in (("hard", False), ("soft_linear", True)):
        sec = time_kernel(lambda: detection.box_results_with_nms_and_limit(sc, bx, soft_nms=soft), 10, warmup=3)
        post[name + "_ms"] = round(sec * 1e3, 3)
    out["detection_postprocess_R1000_C81"] = post
    # the RPN-side caller of NMS (generate_proposals.py:12-182) for one P2-sized level, 2 images, train-time top-k
    from detectron_pytorch_amd import generate_proposals as gp

    anchors = gp.generate_anchors(4, (32,), (0.5, 1, 2))
    sc_np, dl_np = syn.rpn_head_outputs(2, 3, 200, 336, seed=4)
    sc, dl = torch.from_numpy(sc_np).to(device), torch.from_numpy(dl_np).to(device)
    info = torch.tensor([[800, 1344, 1.0], [800, 1344, 1.0]], dtype=torch.float32, device=device)
    op = gp.GenerateProposalsOp(anchors, 0.25, 2000, 2000, 0.7, 0, as_numpy=False)
    sec = time_kernel(lambda: op(sc, dl, info), 10, warmup=3)
    out["generate_proposals_P2_2img_top2000"] = {"ms": round(sec * 1e3, 3)}
    # the result formats of one image (core/test.py:793-866): 100 masks pasted + run-length encoded, 20 persons' keypoints
    from detectron_pytorch_amd.rcnn import results

    masks_np, boxes_np, maps_np, person_np = syn.result_format_inputs()
    masks, maps = torch.from_numpy(masks_np).to(device), torch.from_numpy(maps_np).to(device)
    boxes_int = results.expand_boxes(torch.from_numpy(boxes_np).to(device), 30.0 / 28).to(torch.int32)
    person = torch.from_numpy(person_np).to(device)

    def segm():
        return results.mask_rle(masks, boxes_int, 800, 1333)[2]

    t0 = time.perf_counter()
    for _ in range(5):
        segm()
    host_sec = (time.perf_counter() - t0) / 5
    counts = torch.empty((100, 1024), dtype=torch.int32, device=device)
    strs = torch.empty((100, 4096), dtype=torch.uint8, device=device)
    num = torch.empty((2, 100), dtype=torch.int32, device=device)
    lib = _lib_mod.lib()

    def paste_kernel():
        assert lib.mi_mask_paste_rle(masks.data_ptr(), boxes_int.data_ptr(), 100, 28, 800, 1333, 0.5, 1024, counts.data_ptr(),
                                     num[0].data_ptr(), 4096, strs.data_ptr(), num[1].data_ptr(),
                                     _lib_mod.current_stream_handle(device)) == 0

    # mask targets of a step's foreground RoIs from polygon ground truth (roi_data/mask_rcnn.py:66-76 in one launch)
    from detectron_pytorch_amd import segms as segms_mod
    polys, gt_boxes, _ = syn.polygon_instances(16, seed=9)
    packed = segms_mod.PackedPolygons.from_lists(polys, device=device)
    fg_rois = torch.from_numpy(syn.jittered_boxes(gt_boxes, 16, seed=10)).to(device)
    fg_inst = torch.arange(16, device=device).repeat_interleave(16)
    out["mask_targets"] = {
        "polys_to_masks_256rois_28x28_us": round(time_kernel(lambda: segms_mod.polys_to_masks_wrt_boxes(packed, fg_inst, fg_rois, 28), 20, warmup=3) * 1e6, 1),
        "what": "mi_polys_to_masks_wrt_boxes: 256 foreground RoIs (16 around each of 16 instances of 1-3 polygons, %d vertices "
                "in all) rasterised to 28x28 by pycocotools' rule in one launch" % int(packed.points.size(0))}
    out["result_formats"] = {
        "segm_100_masks_800x1333_ms": round(host_sec * 1e3, 3),
        "mask_paste_rle_kernel_us": round(time_kernel(paste_kernel, 20, warmup=3) * 1e6, 1),
        "keypoint_decode_20x17_us": round(time_kernel(lambda: results.heatmaps_to_keypoints(maps, person), 20, warmup=3) * 1e6, 1),
        "what": "segm = mi_mask_paste_rle for 100 detections (run lengths and COCO strings) + one D2H copy; keypoints = mi_keypoint_decode for 20 person boxes x 17 heat maps (bicubic resize to the box, "
                "arg-max, softmax probability)"}
    return out


def inference_path(device, iters=10):
    """One test-time image through everything between the RPN / box-head convolutions that this repository provides,
    without a host round trip in between: GenerateProposals on P2..P6 (TEST: 1000 pre-NMS / 1000 post-NMS per level,
    configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml:41-43) -> collect the 1000 best -> RoIAlign 7x7 over P2..P5 in one
    fused call -> per-class NMS + top-100 on (synthetic) box-head outputs for those RoIs."""
    from detectron_pytorch_amd import detection, fpn_proposals, generate_proposals as gp
    from detectron_pytorch_amd.roi_align import roi_align_fpn

    levels = [(2, 200, 336, 4, 32), (3, 100, 168, 8, 64), (4, 50, 84, 16, 128), (5, 25, 42, 32, 256), (6, 13, 21, 64, 512)]
    ops, heads = [], []
    for lvl, h, w, stride, size in levels:
        anchors = gp.generate_anchors(stride, (size,), (0.5, 1, 2))
        sc, dl = syn.rpn_head_outputs(1, 3, h, w, seed=lvl)
        ops.append(gp.GenerateProposalsOp(anchors, 1.0 / stride, 1000, 1000, 0.7, 0, as_numpy=False))
        heads.append((torch.from_numpy(sc).to(device), torch.from_numpy(dl).to(device)))
    info = torch.tensor([[800, 1344, 1.0]], dtype=torch.float32, device=device)
    feats = [torch.from_numpy(syn.feature_map(1, syn.FPN_DIM, h, w, seed=l)).to(device) for l, h, w, _, _ in levels[3::-1]]
    scales = [1.0 / s for _, _, _, s, _ in levels[3::-1]]
    cls_np, box_np = syn.detection_head_outputs(1000, 81, seed=7)
    cls, box = torch.from_numpy(cls_np).to(device), torch.from_numpy(box_np).to(device)
    stats = {}

    def run():
        rois = fpn_proposals.generate_and_collect(ops, heads, info, 1000)
        lv = fpn_proposals.map_rois_to_fpn_levels(rois[:, 1:5])
        with torch.no_grad():
            pooled = roi_align_fpn(feats, scales, rois, 5 - lv, 7, 7, 2)
        dets = detection.box_results_with_nms_and_limit(cls[:rois.size(0)], box[:rois.size(0)])
        stats["rois"], stats["detections"] = int(rois.size(0)), int(dets[0].numel())
        return pooled

    def run_static():
        """The same work with fixed shapes and no host synchronisation (static RoI blob + validity mask, image index -1
        for the padding rows, mi_nms_segmented on the blobs in place): the form a hipGraph takes."""
        rois, valid, lv = fpn_proposals.generate_and_collect(ops, heads, info, 1000, static=True, with_levels=True)
        with torch.no_grad():
            pooled = roi_align_fpn(feats, scales, rois, 5 - lv, 7, 7, 2)
        return pooled, detection.box_results_static(cls, box, roi_valid=valid)

    sec = time_kernel(run, iters, warmup=3)
    out = {"eager_dynamic_ms_per_image": round(sec * 1e3, 3), **stats}
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        for _ in range(3):
            run_static()
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        pooled, res = run_static()
        sizes = res["sizes"]

    def replay():
        graph.replay()
        return sizes.cpu()                      # the one device-to-host copy per image: the result sizes

    got = replay().tolist()
    sec_graph = time_kernel(replay, 50, warmup=5)
    out.update(ms_per_image=round(sec_graph * 1e3, 3), launch="hipgraph", host_syncs_per_image=1,
               detections_static=int(got[0]), static_equals_dynamic=bool(got[0] == got[1] == stats["detections"]),
               what="GenerateProposals P2-P6 + collect + fused RoIAlign P2-P5 + per-class NMS (mi_nms_segmented) + top-100, "
                    "one image, static shapes, captured once and replayed; eager_dynamic_ms_per_image = the same work "
                    "launched from Python with the reference's variable-length results")
    out["records_by_producer"] = producer_records_pair(device, feats, scales)
    return out


def producer_records_pair(device, feats, scales, rows=1000, iters=200):
    """The last launch of the proposal stage + the box head's pooling call, as two C-ABI calls on resident buffers:
    mi_rpn_collect_finish + mi_roi_align_forward_fpn (three launches: blob, records, gather) against
    mi_rpn_collect_finish_records + mi_roi_align_forward_fpn_records (two: blob + records, gather)."""
    import ctypes

    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import PreparedRecords

    lib, stream = _lib.lib(), _lib.current_stream_handle(device)
    rois_np, _ = syn.rois_fpn_distributed(rows + 200, batch=1, seed=4)
    cand = torch.from_numpy(rois_np).to(device)
    best, inds = torch.topk(torch.rand(rows + 200, generator=torch.Generator().manual_seed(1)).to(device), rows)
    inds = inds.to(torch.int64).contiguous()
    rois = torch.empty((rows, 5), device=device)
    valid = torch.empty((rows,), dtype=torch.bool, device=device)
    lv = torch.empty((rows,), dtype=torch.int32, device=device)
    idx = torch.empty((rows,), dtype=torch.int32, device=device)
    out = torch.empty((rows, syn.FPN_DIM, 7, 7), device=device)
    rec = PreparedRecords(feats, scales, 7, 7, 2, rows)
    ws = rec.workspace

    def plain():
        assert lib.mi_rpn_collect_finish(best.data_ptr(), inds.data_ptr(), cand.data_ptr(), rows, 1, 2, 5, 224.0, 4.0,
                                         rois.data_ptr(), valid.data_ptr(), lv.data_ptr(), stream) == 0
        torch.sub(5, lv, out=idx)
        assert lib.mi_roi_align_forward_fpn(ctypes.byref(rec.table), rois.data_ptr(), idx.data_ptr(), out.data_ptr(), 1,
                                            syn.FPN_DIM, rows, 7, 7, 2, rec.layout, ws.data_ptr(), ws.numel(), stream) == 0

    def fused():
        assert lib.mi_rpn_collect_finish_records(best.data_ptr(), inds.data_ptr(), cand.data_ptr(), rows, 1, 2, 5, 224.0, 4.0,
                                                 rois.data_ptr(), valid.data_ptr(), lv.data_ptr(), ctypes.byref(rec.table), 1,
                                                 syn.FPN_DIM, 7, 7, 2, rec.layout, ws.data_ptr(), ws.numel(), stream) == 0
        torch.sub(5, lv, out=idx)
        assert lib.mi_roi_align_forward_fpn_records(ctypes.byref(rec.table), rois.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                                    1, syn.FPN_DIM, rows, 7, 7, 2, rec.layout, ws.data_ptr(), ws.numel(),
                                                    stream) == 0

    a = [time_kernel(plain, iters) * 1e6, 0.0]
    b = [time_kernel(fused, iters) * 1e6, 0.0]
    a[1], b[1] = time_kernel(plain, iters) * 1e6, time_kernel(fused, iters) * 1e6
    return {"collect_finish_plus_forward_fpn_us": [round(x, 1) for x in a],
            "collect_finish_records_plus_forward_fpn_records_us": [round(x, 1) for x in b],
            "what": "1000-row blob over P2-P5, 7x7, alternating runs of %d call pairs" % iters}
