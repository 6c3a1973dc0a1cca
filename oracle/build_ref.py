#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: build the reference's own hot-path sources for the host into oracle/_ref/.

Nothing here is product code and no reference source is copied into the repository: every
input is read from where it lies under /root/reference, intermediate files live in a temporary
directory that is deleted afterwards, and only the built shared objects land in oracle/_ref/
(git-ignored; they travel to the GPU box with the gpurun snapshot because /root/reference does
not exist there).

Two families of artefacts:

1. The four CUDA translation units of the path
       lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu   -> libref_roi_align.so
       lib/model/roi_align/src/roi_align_kernel.cu                -> libref_roi_align_legacy.so
       lib/model/roi_pooling/src/roi_pooling_kernel.cu            -> libref_roi_pool.so
       lib/model/roi_crop/src/roi_crop_cuda_kernel.cu             -> libref_roi_crop.so
       lib/model/nms/src/nms_cuda_kernel.cu                       -> libref_nms.so
   compiled with g++ against oracle/cuda_on_cpu.h.  The only edit is mechanical: the CUDA launch
   syntax `k<<<cfg...>>>(args)` is not C++, so it is rewritten to
   `cuda_on_cpu::launch(k, cuda_on_cpu::cfg(cfg...), args)`.  Kernel bodies are untouched.
   The exported symbols are the reference's own `extern "C"` launchers
   (ROIAlignForwardLaucher, ROIPoolForwardLaucher, BilinearSamplerBHWD_*_cuda_kernel,
   nms_cuda_compute ...), called with host pointers.

2. The Cython modules lib/utils/cython_nms.pyx and cython_bbox.pyx -> cython_nms*.so, cython_bbox*.so.
   cython_nms.pyx does not cythonize under Cython 3 / numpy 2 as shipped; the 2-token
   compatibility patch of SURVEY.md section 8c (np.int_t -> np.intp_t, dtype=np.int -> np.intp; no
   arithmetic touched) is applied to the temporary copy.

Usage: python oracle/build_ref.py [--reference /root/reference] [--force]
Exit status 0 with a message when the reference tree is absent (GPU box): the prebuilt files
are used as they are.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

CUDA_UNITS = [
    # (relative .cu path, output name, extra defines)
    ("lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu", "libref_roi_align.so", []),
    ("lib/model/roi_align/src/roi_align_kernel.cu", "libref_roi_align_legacy.so", []),
    ("lib/model/roi_pooling/src/roi_pooling_kernel.cu", "libref_roi_pool.so", []),
    ("lib/model/roi_crop/src/roi_crop_cuda_kernel.cu", "libref_roi_crop.so", []),
    # nms_kernel stages a tile in __shared__ behind __syncthreads(): run each block twice.
    ("lib/model/nms/src/nms_cuda_kernel.cu", "libref_nms.so", ["-DCUDA_ON_CPU_PASSES=2"]),
]
CYTHON_UNITS = [
    ("lib/utils/cython_nms.pyx", "cython_nms", [("np.int_t", "np.intp_t"), ("dtype=np.int)", "dtype=np.intp)")]),
    ("lib/utils/cython_bbox.pyx", "cython_bbox", []),
]

_LAUNCH = re.compile(r"(\b[A-Za-z_]\w*)\s*<<<(.*?)>>>\s*\(", re.S)


def rewrite_launches(src: str) -> str:
    return _LAUNCH.sub(lambda m: "cuda_on_cpu::launch(%s, cuda_on_cpu::cfg(%s), " % (m.group(1), m.group(2)), src)


def newer(target, *sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_cuda_units(ref_root, tmp, force):
    shim = os.path.join(HERE, "cuda_on_cpu.h")
    for rel, out_name, defines in CUDA_UNITS:
        src = os.path.join(ref_root, rel)
        out = os.path.join(OUT, out_name)
        if not force and newer(out, src, shim, __file__):
            continue
        with open(src) as f:
            text = rewrite_launches(f.read())
        tmp_src = os.path.join(tmp, out_name.replace(".so", ".cpp"))
        with open(tmp_src, "w") as f:
            f.write(text)
        cmd = ["g++", "-x", "c++", "-std=c++14", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w",
               "-include", shim, "-I", os.path.dirname(src)] + defines + [tmp_src, "-o", out]
        subprocess.check_call(cmd)
        print("built", os.path.relpath(out, HERE))


def build_cython_units(ref_root, tmp, force):
    import numpy as np

    ext_suffix = sysconfig.get_config_var("EXT_SUFFIX")
    py_inc = sysconfig.get_paths()["include"]
    for rel, mod, patches in CYTHON_UNITS:
        src = os.path.join(ref_root, rel)
        out = os.path.join(OUT, mod + ext_suffix)
        if not force and newer(out, src, __file__):
            continue
        with open(src) as f:
            text = f.read()
        for old, new in patches:
            assert old in text, "patch token %r not found in %s" % (old, rel)
            text = text.replace(old, new)
        pyx = os.path.join(tmp, mod + ".pyx")
        with open(pyx, "w") as f:
            f.write(text)
        c_file = os.path.join(tmp, mod + ".c")
        subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_file],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        # -O2, no -march: the reference's setup.py builds with the distutils defaults plus
        # -Wno-cpp (lib/setup.py:37-40); x86-64 baseline has no FMA, so IoU stays unfused.
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-w", "-fno-strict-aliasing",
               "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION", "-I", py_inc, "-I", np.get_include(),
               c_file, "-o", out]
        subprocess.check_call(cmd)
        print("built", os.path.relpath(out, HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("REFERENCE_ROOT", "/root/reference"))
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.reference, "lib")):
        print("reference tree not found at %s: keeping prebuilt oracle/_ref as is" % args.reference)
        return 0
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="oracle_ref_")
    try:
        build_cuda_units(args.reference, tmp, args.force)
        build_cython_units(args.reference, tmp, args.force)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
