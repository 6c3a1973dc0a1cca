"""Re-entrancy of the C-ABI (-m gpu).  The reference calls its operators from one Python thread per GPU
(lib/nn/parallel/parallel_apply.py:41-59: a threading.Thread per replica, each entering the extension concurrently);
include/mi_detectron_ops.h promises the same of this library: no process-wide mutable state on the launch path, caller-owned
workspaces, a per-thread error text.  Here two (and four) host threads, each with its own HIP stream, its own workspace and
its own inputs, hammer mi_roi_align_forward_ws / mi_roi_align_backward_ws / mi_nms (round 6: and mi_roi_pool_forward /
_backward / mi_roi_crop_backward_ws) at the same time (ctypes drops the GIL
around every call); every result must equal the one the same call produced alone, on every repetition.

The backward is called repeatedly over ONE workspace with RECORDS_READY: its plan counters alternate between two sets
(roi_align_record_layout.h) instead of being reset by a launch -- every repetition must give the first one's bits.
"""
import ctypes
import threading

import numpy as np
import pytest
import torch

from detectron_pytorch_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu

REPS = 12


def _dev():
    return torch.device("cuda", 0)


class Job:
    """One thread's private problem: a RoIAlign forward + backward and an NMS, all buffers its own."""

    def __init__(self, seed, rois_n, channels, res, channels_last, adversarial):
        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

        d = _dev()
        self.n, self.c, self.h, self.w, self.scale, self.res, self.r = 2, channels, 50, 84, 1.0 / 16, res, rois_n
        self.layout = _lib.LAYOUT_NHWC if channels_last else _lib.LAYOUT_NCHW
        feat = torch.from_numpy(syn.feature_map(self.n, self.c, self.h, self.w, seed=seed)).to(d)
        self.feat = feat.permute(0, 2, 3, 1).contiguous() if channels_last else feat
        rois = (syn.rois_adversarial(rois_n, self.n, self.h, self.w, self.scale, seed=seed + 1) if adversarial
                else syn.rois_canonical(rois_n, self.n, seed=seed + 1, side=(24.0, 400.0), im_h=self.h * 16, im_w=self.w * 16))
        self.rois = torch.from_numpy(rois).to(d)
        self.gtop = torch.from_numpy(np.random.RandomState(seed + 2).randn(rois_n, self.c, res, res).astype(np.float32)).to(d)
        self.out = torch.empty((rois_n, self.c, res, res), device=d)
        self.gin = torch.empty_like(self.feat)
        self.ws = torch.empty(_backward_workspace_bytes([(self.h, self.w)], self.n, rois_n), dtype=torch.uint8, device=d)
        nb = 600 + 97 * seed
        self.dets = torch.from_numpy(syn.boxes_clustered(nb, seed=seed)).to(d)
        self.keep = torch.empty(nb, dtype=torch.int64, device=d)
        self.num_keep = torch.zeros(1, dtype=torch.int32, device=d)
        self.nms_ws = torch.empty(_lib.lib().mi_nms_workspace_bytes(nb), dtype=torch.uint8, device=d)
        self.nb = nb
        # round 6: the RoIPool / RoICrop tile kernels (NCHW only): forward + ordered backward, and the sampler's backward
        # over a workspace of its own
        self.pool = not channels_last
        if self.pool:
            self.pool_out = torch.empty((rois_n, self.c, res, res), device=d)
            self.argmax = torch.empty((rois_n, self.c, res, res), dtype=torch.int32, device=d)
            self.pool_gin = torch.empty_like(self.feat)
            rr = (rois_n // self.n) * self.n
            grid = syn.crop_grid(rr, res, res, seed=seed + 3, span=1.1)
            self.crop_r = rr
            self.grid = torch.from_numpy(grid).to(d)
            self.crop_gin = torch.empty_like(self.feat)
            self.crop_ws = torch.empty(_lib.lib().mi_roi_crop_backward_workspace_bytes(rr), dtype=torch.uint8, device=d)

    def run(self, lib, stream_handle):
        """forward (writes the records), backward over those records (OVERWRITE), NMS -- on `stream_handle`."""
        rc = lib.mi_roi_align_forward_ws(self.feat.data_ptr(), self.rois.data_ptr(), self.out.data_ptr(), self.n, self.c, self.h,
                                         self.w, self.r, self.res, self.res, self.scale, 2, _lib.ROI_ALIGN_CAFFE2, self.layout,
                                         self.ws.data_ptr(), self.ws.numel(), stream_handle)
        assert rc == 0, lib.mi_last_error()
        self.backward(lib, stream_handle)
        rc = lib.mi_nms(self.dets.data_ptr(), self.nb, 0.5, _lib.NMS_GE_ORIG_ASC, self.keep.data_ptr(), self.num_keep.data_ptr(),
                        self.nms_ws.data_ptr(), self.nms_ws.numel(), stream_handle)
        assert rc == 0, lib.mi_last_error()
        if self.pool:
            rc = lib.mi_roi_pool_forward(self.feat.data_ptr(), self.rois.data_ptr(), self.pool_out.data_ptr(), self.argmax.data_ptr(),
                                         self.n, self.c, self.h, self.w, self.r, self.res, self.res, self.scale, stream_handle)
            assert rc == 0, lib.mi_last_error()
            rc = lib.mi_roi_pool_backward(self.gtop.data_ptr(), self.rois.data_ptr(), self.argmax.data_ptr(), self.pool_gin.data_ptr(),
                                          self.n, self.c, self.h, self.w, self.r, self.res, self.res, self.scale, stream_handle)
            assert rc == 0, lib.mi_last_error()
            rc = lib.mi_roi_crop_backward_ws(self.feat.data_ptr(), self.grid.data_ptr(), self.gtop.data_ptr(), self.crop_gin.data_ptr(),
                                             self.n, self.c, self.h, self.w, self.crop_r, self.res, self.res, self.crop_ws.data_ptr(),
                                             self.crop_ws.numel(), stream_handle)
            assert rc == 0, lib.mi_last_error()

    def backward(self, lib, stream_handle):
        rc = lib.mi_roi_align_backward_ws(self.gtop.data_ptr(), self.rois.data_ptr(), self.gin.data_ptr(), self.n, self.c, self.h,
                                          self.w, self.r, self.res, self.res, self.scale, 2, _lib.ROI_ALIGN_CAFFE2, self.layout,
                                          self.ws.data_ptr(), self.ws.numel(),
                                          _lib.ROI_ALIGN_RECORDS_READY | _lib.ROI_ALIGN_OVERWRITE, stream_handle)
        assert rc == 0, lib.mi_last_error()

    def snapshot(self):
        k = int(self.num_keep.item())
        extra = (self.pool_out.clone(), self.argmax.clone(), self.pool_gin.clone(), self.crop_gin.clone()) if self.pool else ()
        return (self.out.clone(), self.gin.clone(), self.keep[:k].clone()) + extra


def _same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


def _jobs(count):
    # lists of at most 32 RoIs per tile (no atomically added slices): the backward is deterministic, bit equality is the bar;
    # job 1 has RoIs without backward tables (outside the map, wider than 63 columns), job 2 is channels-last
    spec = [(0, 96, 64, 7, False, False), (1, 64, 32, 7, False, True), (2, 80, 64, 14, True, False), (3, 48, 96, 7, False, False)]
    return [Job(*spec[i]) for i in range(count)]


@pytest.mark.parametrize("nthreads", [2, 4])
def test_concurrent_host_threads_equal_the_serial_calls(nthreads):
    lib = _lib.lib()
    d = _dev()
    jobs = _jobs(nthreads)
    serial = []
    for j in jobs:  # alone, on the default stream
        j.run(lib, _lib.current_stream_handle(d))
        torch.cuda.synchronize()
        serial.append(j.snapshot())
    streams = [torch.cuda.Stream(device=d) for _ in jobs]
    torch.cuda.synchronize()
    errors, start = [], threading.Barrier(nthreads)

    def worker(i):
        try:
            torch.cuda.set_device(d)
            handle = ctypes.c_void_p(streams[i].cuda_stream)
            start.wait()
            for rep in range(REPS):
                jobs[i].run(lib, handle)
                streams[i].synchronize()
                if not _same(jobs[i].snapshot(), serial[i]):
                    errors.append("thread %d repetition %d differs from its serial result" % (i, rep))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_repeated_backward_over_one_workspace_alternates_its_counter_sets():
    """RECORDS_READY backward calls back to back (what bench.py's timed loop and a gradient-accumulation loop do): nothing
    but the backward's own launches touches the plan counters between them."""
    lib = _lib.lib()
    d = _dev()
    for job in _jobs(3):
        h = _lib.current_stream_handle(d)
        job.run(lib, h)
        torch.cuda.synchronize()
        first = job.gin.clone()
        for rep in range(7):
            job.gin.fill_(float("nan"))
            job.backward(lib, h)
            torch.cuda.synchronize()
            assert torch.equal(job.gin, first), "backward repetition %d over the same workspace differs" % rep


def test_last_error_is_per_host_thread():
    """One thread makes calls the library refuses while another one makes good calls: each reads its own error text."""
    lib = _lib.lib()
    d = _dev()
    job = _jobs(1)[0]
    stream = torch.cuda.Stream(device=d)
    seen, start = {}, threading.Barrier(2)

    def bad():
        torch.cuda.set_device(d)
        start.wait()
        texts = set()
        for _ in range(200):
            rc = lib.mi_nms(job.dets.data_ptr(), -1, 0.5, _lib.NMS_GE_ORIG_ASC, job.keep.data_ptr(), job.num_keep.data_ptr(),
                            job.nms_ws.data_ptr(), job.nms_ws.numel(), ctypes.c_void_p(stream.cuda_stream))
            texts.add((rc, lib.mi_last_error()))
        seen["bad"] = texts

    def good():
        torch.cuda.set_device(d)
        start.wait()
        texts = set()
        for _ in range(50):
            job.run(lib, _lib.current_stream_handle(d))
            texts.add(lib.mi_last_error())
        torch.cuda.synchronize()
        seen["good"] = texts

    ts = [threading.Thread(target=bad), threading.Thread(target=good)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(seen["bad"]) == 1
    rc, text = next(iter(seen["bad"]))
    assert rc != 0 and text, "a refused call must leave its reason in the calling thread's error text"
    assert seen["good"] == {b""}, "the refusing thread's text leaked into the other thread: %r" % (seen["good"],)
