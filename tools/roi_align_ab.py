#!/usr/bin/env python3
"""A/B timing of the RoIAlign forward launch forms inside ONE process on ONE box (boxes differ by several percent, so
variants are only ever compared inside one run): each arm is a set of MI_ROI_ALIGN_* variables (and optionally another
build of the library), arms are visited round-robin ROUNDS times, HIP events around ITERS back-to-back calls of the C-ABI
entry point; the first arm's output is the reference the others must equal bit for bit.

usage: python tools/roi_align_ab.py "name:VAR=v,VAR=v[,lib=path]" ...      (ROUNDS=3 ITERS=200 SHAPES="config2 nhwc mask box2 fpn")
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402

TUNING_VARS = ("MI_ROI_ALIGN_BWD_SLICE", "MI_ROI_ALIGN_BWD_TH", "MI_ROI_ALIGN_CAP", "MI_ROI_ALIGN_ABLATE", "MI_ROI_ALIGN_IMPL",
               "MI_ROI_ALIGN_NO_WS", "MI_ROI_ALIGN_SLAB", "MI_ROI_ALIGN_FWD_FULL_WAIT", "MI_ROI_ALIGN_NHWC_V", "MI_ROI_ALIGN_NHWC_PB", "MI_ROI_ALIGN_NHWC_ZIGZAG", "MI_ROI_ALIGN_FWD_SPLIT")
_LIBS = {}


def load(path):
    path = os.path.abspath(path)
    if path not in _LIBS:
        h = ctypes.CDLL(path)
        for name, (restype, argtypes) in _lib.SIGNATURES.items():
            if hasattr(h, name):
                fn = getattr(h, name)
                fn.restype, fn.argtypes = restype, argtypes
        _LIBS[path] = h
    return _LIBS[path]


class Arm:
    def __init__(self, spec):
        self.name, _, rest = spec.partition(":")
        self.env, self.lib_path = {}, _lib.LIB_PATH
        for kv in filter(None, rest.split(",")):
            k, _, v = kv.partition("=")
            if k == "lib":
                self.lib_path = v
            else:
                self.env[k] = v

    def activate(self):
        for k in TUNING_VARS:
            os.environ.pop(k, None)
        os.environ.update(self.env)
        h = load(self.lib_path)
        h.mi_dbg_reload_tuning()
        return h


def shapes(dev, which):
    """name -> (call(lib) -> rc, output tensor)"""
    out = {}
    h, w, scale = syn.FPN_LEVELS[2]
    c = syn.FPN_DIM

    def single(name, n, r, res, layout, seed=0):
        feat = torch.from_numpy(syn.feature_map(n, c, h, w, seed=seed)).to(dev)
        if layout:
            feat = feat.permute(0, 2, 3, 1).contiguous()
        rois = torch.from_numpy(syn.rois_canonical(r, n, seed=seed)).to(dev)
        o = torch.empty((r, c, res, res), device=dev)
        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes
        # the forward-sized workspace of bench.py's roofline call (WS=bwd: with room for a backward, whose tables the
        # records launch then writes too, +2 us)
        ws_bytes = (_backward_workspace_bytes([(h, w)], n, r) if os.environ.get("WS") == "bwd"
                    else _lib.lib().mi_roi_align_forward_workspace_bytes(r))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        stream = _lib.current_stream_handle(dev)

        def call(lib, n=n, r=r, res=res, layout=layout, feat=feat, rois=rois, o=o, ws=ws):
            return lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), o.data_ptr(), n, c, h, w, r, res, res,
                                               scale, 2, 0, layout, ws.data_ptr(), ws.numel(), stream)
        out[name] = (call, o, (feat, rois, ws))

    if "config2" in which:
        single("config2", 1, 512, 7, 0)
    if "nhwc" in which:
        single("nhwc", 1, 512, 7, 1)
    if "mask" in which:
        single("mask128", 1, 128, 14, 0, seed=1)
    if "box2" in which:
        single("box1024x2img", 2, 1024, 7, 0)
    if "nhwc2" in which:
        single("nhwc1024x2img", 2, 1024, 7, 1)
    for key, res in (("fpn", 7), ("fpnmask", 14)):  # the fused pyramid call on the RoIs a training step samples
        if key not in which:
            continue
        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes, _fpn_table
        g = np.load(os.path.join(ROOT, "tests", "golden", "step_rois.npz"))
        rois = torch.from_numpy(g["rois" if res == 7 else "mask_rois"]).to(dev).contiguous()
        idx = (5 - torch.from_numpy(g["levels" if res == 7 else "mask_levels"]).to(dev)).clamp(0, 3).to(torch.int32).contiguous()
        maps = [torch.from_numpy(syn.feature_map(2, 256, syn.FPN_LEVELS[l][0], syn.FPN_LEVELS[l][1], seed=l)).to(dev)
                for l in (5, 4, 3, 2)]
        scales = [syn.FPN_LEVELS[l][2] for l in (5, 4, 3, 2)]
        r = int(rois.size(0))
        o = torch.empty((r, 256, res, res), device=dev)
        # WS=fwd: the forward-sized workspace of an inference call (no backward tables; the records-free kernel serves it)
        ws = torch.empty(_lib.lib().mi_roi_align_forward_workspace_bytes(r) if os.environ.get("WS") == "fwd" else
                         _backward_workspace_bytes([(m.size(2), m.size(3)) for m in maps], 2, r), dtype=torch.uint8, device=dev)
        ftab = _fpn_table(maps, scales)
        stream = _lib.current_stream_handle(dev)

        def call(lib, rois=rois, idx=idx, o=o, r=r, res=res, ws=ws, ftab=ftab):
            return lib.mi_roi_align_forward_fpn(ctypes.byref(ftab), rois.data_ptr(), idx.data_ptr(), o.data_ptr(), 2, 256, r,
                                                res, res, 2, 0, ws.data_ptr(), ws.numel(), stream)
        out["fpn_step_%s" % ("box" if res == 7 else "mask")] = (call, o, (maps, rois, idx, ws, ftab))
    # ---- backward cases: (call, output, keep, setup) -- setup(lib) runs that library's forward so that ITS records are in
    # the workspace (the record layout differs between builds); the timed call has RECORDS_READY | OVERWRITE set
    for key, layout in (("bwd_config2", 0), ("bwd_nhwc", 1)):
        if key not in which:
            continue
        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes
        n, r, res = 1, 512, 7
        feat = torch.from_numpy(syn.feature_map(n, c, h, w, seed=0)).to(dev)
        if layout:
            feat = feat.permute(0, 2, 3, 1).contiguous()
        rois = torch.from_numpy(syn.rois_canonical(r, n, seed=0)).to(dev)
        o = torch.empty((r, c, res, res), device=dev)
        gtop = torch.randn(r, c, res, res, device=dev)
        gin = torch.empty_like(feat)
        ws = torch.empty(_backward_workspace_bytes([(h, w)], n, r) + 65536, dtype=torch.uint8, device=dev)
        stream = _lib.current_stream_handle(dev)

        def setup(lib, n=n, r=r, res=res, feat=feat, rois=rois, o=o, ws=ws, layout=layout):
            assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), o.data_ptr(), n, c, h, w, r, res, res,
                                               scale, 2, 0, layout, ws.data_ptr(), ws.numel(), stream) == 0

        def call(lib, n=n, r=r, res=res, gtop=gtop, rois=rois, gin=gin, ws=ws, layout=layout):
            return lib.mi_roi_align_backward_ws(gtop.data_ptr(), rois.data_ptr(), gin.data_ptr(), n, c, h, w, r, res, res,
                                                scale, 2, 0, layout, ws.data_ptr(), ws.numel(), 3, stream)
        out[key] = (call, gin, (feat, rois, ws, gtop, o), setup)
    for key, res in (("bwd_fpn", 7), ("bwd_fpnmask", 14)):
        if key not in which:
            continue
        from detectron_pytorch_amd.roi_align import _backward_workspace_bytes, _fpn_table
        g = np.load(os.path.join(ROOT, "tests", "golden", "step_rois.npz"))
        rois = torch.from_numpy(g["rois" if res == 7 else "mask_rois"]).to(dev).contiguous()
        idx = (5 - torch.from_numpy(g["levels" if res == 7 else "mask_levels"]).to(dev)).clamp(0, 3).to(torch.int32).contiguous()
        maps = [torch.from_numpy(syn.feature_map(2, 256, syn.FPN_LEVELS[l][0], syn.FPN_LEVELS[l][1], seed=l)).to(dev)
                for l in (5, 4, 3, 2)]
        flat = torch.empty(sum(m.numel() for m in maps), device=dev)
        grads, at = [], 0
        for m in maps:
            grads.append(flat[at:at + m.numel()].view_as(m))
            at += m.numel()
        scales = [syn.FPN_LEVELS[l][2] for l in (5, 4, 3, 2)]
        r = int(rois.size(0))
        o = torch.empty((r, 256, res, res), device=dev)
        gtop = torch.randn(r, 256, res, res, device=dev)
        ws = torch.empty(_backward_workspace_bytes([(m.size(2), m.size(3)) for m in maps], 2, r) + 65536, dtype=torch.uint8, device=dev)
        ftab, gtab = _fpn_table(maps, scales), _fpn_table(grads, scales, grads=True)
        stream = _lib.current_stream_handle(dev)

        def setup(lib, rois=rois, idx=idx, o=o, r=r, res=res, ws=ws, ftab=ftab):
            assert lib.mi_roi_align_forward_fpn(ctypes.byref(ftab), rois.data_ptr(), idx.data_ptr(), o.data_ptr(), 2, 256, r,
                                                res, res, 2, 0, ws.data_ptr(), ws.numel(), stream) == 0

        def call(lib, rois=rois, idx=idx, gtop=gtop, r=r, res=res, ws=ws, gtab=gtab):
            return lib.mi_roi_align_backward_fpn(ctypes.byref(gtab), gtop.data_ptr(), rois.data_ptr(), idx.data_ptr(), 2, 256, r,
                                                 res, res, 2, 0, ws.data_ptr(), ws.numel(), 3, stream)
        out["bwd_fpn_step_%s" % ("box" if res == 7 else "mask")] = (call, flat, (maps, grads, rois, idx, ws, ftab, gtab, gtop, o), setup)
    return out


def time_calls(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    arms = [Arm(s) for s in sys.argv[1:]] or [Arm("default:")]
    rounds, iters = int(os.environ.get("ROUNDS", "3")), int(os.environ.get("ITERS", "200"))
    dev = torch.device("cuda", 0)
    cases = shapes(dev, os.environ.get("SHAPES", "config2 nhwc mask box2").split())
    res = {n: {a.name: [] for a in arms} for n in cases}
    ref = {}
    for rnd in range(rounds):
        for arm in arms:
            lib = arm.activate()
            for name, case in cases.items():
                call, o = case[0], case[1]
                if len(case) > 3:
                    case[3](lib)

                def fn():
                    rc = call(lib)
                    assert rc == 0, (arm.name, name, lib.mi_last_error())
                if rnd == 0:
                    o.fill_(float("nan"))
                    fn()
                    torch.cuda.synchronize()
                    got = o.clone()
                    ablated = "MI_ROI_ALIGN_ABLATE" in arm.env
                    if name not in ref:
                        ref[name] = got
                    elif not ablated:
                        same = torch.equal(got, ref[name])
                        diff = float((got - ref[name]).abs().max()) if not same else 0.0
                        # the backward adds the slices of a long list with atomics: last bits depend on their order
                        verdict = "bit-equal" if same else ("within 1e-4" if name.startswith("bwd") and diff <= 1e-4 else "DIFFERS")
                        print("check %-14s %-12s %s (max |d| %.3g, nan %d)" % (name, arm.name, verdict, diff,
                                                                                   int(torch.isnan(got).sum())), flush=True)
                res[name][arm.name].append(round(time_calls(fn, iters), 2))
    for name in cases:
        for arm in arms:
            v = res[name][arm.name]
            print(json.dumps({"shape": name, "arm": arm.name, "env": arm.env, "us_per_call": v, "min": min(v)}), flush=True)


if __name__ == "__main__":
    main()
